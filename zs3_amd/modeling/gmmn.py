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
