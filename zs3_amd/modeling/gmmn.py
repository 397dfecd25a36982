"""GMMN generator -- drop-in for zs3.modeling.gmmn.GMMNnetwork (gmmn.py:6-49).  cat(embd, noise) and the two
Linear layers run as row-GEMMs on the conv kernel (bias + LeakyReLU fused in the epilogue)."""
import torch
import torch.nn as nn

from .. import functional as Fz
from .layers import Dropout


class Linear(nn.Linear):
    def forward(self, x, act=Fz.ACT_NONE, leak=0.2):
        n, c = x.shape
        y = Fz.conv_bn_act(x.reshape(1, 1, n, c), self.weight, bias=self.bias, act=act, leak=leak)
        return y.reshape(n, self.out_features)


class GMMNnetwork(nn.Module):
    def __init__(self, noise_dim, embed_dim, hidden_size, feature_dim, semantic_reconstruction=False):
        super().__init__()
        d_in = noise_dim + embed_dim
        if hidden_size:
            self.model = nn.Sequential(Linear(d_in, hidden_size), nn.LeakyReLU(0.2, inplace=True), Dropout(p=0.5),
                                       Linear(hidden_size, feature_dim))
            linears = [self.model[0], self.model[3]]
        else:
            self.model = Linear(d_in, feature_dim)
            linears = [self.model]
        for lin in linears:
            torch.nn.init.xavier_uniform_(lin.weight)
            lin.bias.data.fill_(0.01)
        self.semantic_reconstruction = semantic_reconstruction
        if semantic_reconstruction:
            self.semantic_reconstruction_layer = Linear(feature_dim, d_in)

    def _mlp(self, x):
        if isinstance(self.model, nn.Sequential):
            h = self.model[0](x, act=Fz.ACT_LEAKY, leak=self.model[1].negative_slope)
            h = self.model[2](h)
            return self.model[3](h)
        return self.model(x)

    def forward(self, embd, noise):
        features = self._mlp(torch.cat((embd, noise), 1))
        if self.semantic_reconstruction:
            return features, self.semantic_reconstruction_layer(features)
        return features


class GraphConvolution(nn.Module):
    """pygcn.layers.GraphConvolution (tkipf/pygcn, the un-vendored dependency of zs3/modeling/gmmn.py:2): weight stored
    [in, out], uniform(-1/sqrt(out), 1/sqrt(out)) init, forward = adj @ (x @ W) + b.  Both products run as row-GEMMs on the
    MFMA conv kernel; adj may be a dense [N, N] tensor or a torch sparse tensor (densified: cluster graphs are tiny)."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(in_features, out_features))
        self.bias = nn.Parameter(torch.empty(out_features)) if bias else None
        stdv = 1.0 / out_features ** 0.5
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.uniform_(-stdv, stdv)

    def forward(self, x, adj, act=Fz.ACT_NONE, leak=0.2):
        if adj.is_sparse:
            adj = adj.to_dense()
        n = x.shape[0]
        support = Fz.conv_bn_act(x.reshape(1, 1, n, x.shape[1]), self.weight.t()).reshape(n, self.out_features)
        # adj @ support as a row-GEMM: rows = adj rows, "weight" = support^T [out, N]; the reduction axis N (number of
        # clusters, arbitrary) is zero-padded to a multiple of 8 for the kernel's 16-byte row loads
        n8 = (n + 7) // 8 * 8
        adj_p = torch.nn.functional.pad(adj.float(), (0, n8 - n)).contiguous()
        sup_t = torch.nn.functional.pad(support.t(), (0, n8 - n))
        out = Fz.conv_bn_act(adj_p.reshape(1, 1, n, n8), sup_t, bias=self.bias, act=act, leak=leak)
        return out.reshape(n, self.out_features)


class GMMNnetwork_GCN(nn.Module):
    """drop-in for zs3.modeling.gmmn.GMMNnetwork_GCN (gmmn.py:52-67): keys gcn{1,2}.{weight,bias}, weight [in, out]"""

    def __init__(self, noise_dim=300, embed_dim=300, hidden_size=256, feature_dim=256):
        super().__init__()
        self.gcn1 = GraphConvolution(noise_dim + embed_dim, hidden_size)
        self.relu = nn.LeakyReLU(0.2)
        self.dropout = Dropout(p=0.5)
        self.gcn2 = GraphConvolution(hidden_size, feature_dim)
        for m in (self.gcn1, self.gcn2):
            torch.nn.init.xavier_uniform_(m.weight)
            m.bias.data.fill_(0.01)

    def forward(self, embd, noise, adj_mat):
        x = self.gcn1(torch.cat((embd, noise), 1), adj_mat, act=Fz.ACT_LEAKY, leak=self.relu.negative_slope)
        return self.gcn2(self.dropout(x), adj_mat)
