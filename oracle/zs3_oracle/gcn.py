"""CPU restatement of the GCN-context cluster graph and generator -- test infrastructure only.

cluster_graph follows construct_adj_mat (zs3/train_context_GMMN_GCNcontext.py:33-102) and is pinned against it by
tests/golden/gcn_graph.npz (generated from the reference by tools/make_goldens.py).  The GraphConvolution arithmetic
lives in the un-vendored dependency tkipf/pygcn (unpinned version; absent from /root/reference): gcn_forward restates its
published form `adj @ (x @ W) + b` (pygcn/layers.py) behind the reference's call site zs3/modeling/gmmn.py:52-67, so the
GCN layer's parity is anchored on that call site only ("parity unpinned" for the layer itself)."""
import numpy as np
import torch
import torch.nn as nn


def cluster_graph(segmap, embeddingmap, featmap=None, avg_feat=False):
    """-> (adj [Nc, Nc] float32 dense 0/1 or None when Nc <= 1, cmap [H, W] cluster id per pixel, labels [Nc],
    emb [Nc, E], feat [Nc, D] or None).

    Clusters are the 8-connected components of equal label, numbered in raster order of their first pixel
    (train_context_GMMN_GCNcontext.py:53-57); two clusters are adjacent when any of their pixels touch in the
    8-neighbourhood (:75-86, each unordered pair once, both directions, weight 1); a cluster's embedding is the embedding
    of its first pixel (:58), and so is its feature -- with avg_feat the reference re-averages that same seed feature once
    per visited pixel in float32, `(f * (cnt - 1) + seed) / cnt` (:70-73), which is reproduced literally."""
    seg = np.asarray(segmap)
    h, w = seg.shape
    cmap = -np.ones((h, w), dtype=np.int64)
    labels, seeds, sizes = [], [], []
    for i in range(h):
        for j in range(w):
            if cmap[i, j] >= 0:
                continue
            cid = len(labels)
            labels.append(seg[i, j])
            seeds.append((i, j))
            # the reference marks a pixel when it is POPPED (:69), so a pixel can sit on the stack several times and is
            # then visited (and counted, :66) several times; `pops` is that count -- it drives the avg_feat recurrence
            todo = [(i, j)]
            pops = 0
            while todo:
                a, b = todo.pop()
                pops += 1
                cmap[a, b] = cid
                for da in (-1, 0, 1):
                    for db in (-1, 0, 1):
                        p, q = a + da, b + db
                        if 0 <= p < h and 0 <= q < w and cmap[p, q] < 0 and seg[p, q] == seg[i, j]:
                            todo.append((p, q))
            sizes.append(pops)
    nc = len(labels)
    adj = np.zeros((nc, nc), dtype=np.float32)
    for da in (-1, 0, 1):
        for db in (-1, 0, 1):
            if da == 0 and db == 0:
                continue
            a0, a1 = max(0, -da), h - max(0, da)
            b0, b1 = max(0, -db), w - max(0, db)
            c1 = cmap[a0:a1, b0:b1]
            c2 = cmap[a0 + da:a1 + da, b0 + db:b1 + db]
            diff = c1 != c2
            adj[c1[diff], c2[diff]] = 1.0
    emb = np.stack([np.asarray(embeddingmap)[:, i, j] for (i, j) in seeds])
    feat = None
    if featmap is not None:
        rows = []
        for (i, j), n in zip(seeds, sizes):
            seed = np.asarray(featmap)[:, i, j]
            f = seed
            if avg_feat:
                for cnt in range(2, n + 1):
                    f = (f * (cnt - 1) + seed) / cnt
            rows.append(f)
        feat = np.stack(rows)
    return (adj if nc > 1 else None), cmap, np.asarray(labels), emb, feat


def gcn_forward(x, adj, weight, bias):
    """pygcn GraphConvolution: adj @ (x @ W) + b, weight stored [in, out]."""
    return adj @ (x @ weight) + bias


class GraphConvolution(nn.Module):
    """pygcn.layers.GraphConvolution: weight [in, out], bias [out]; forward = adj @ (x @ W) + b (adj dense or sparse)."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(in_features, out_features))
        self.bias = nn.Parameter(torch.empty(out_features))
        stdv = 1.0 / out_features ** 0.5        # pygcn reset_parameters (overwritten by the caller's init below)
        self.weight.data.uniform_(-stdv, stdv)
        self.bias.data.uniform_(-stdv, stdv)

    def forward(self, x, adj):
        support = x @ self.weight
        return (torch.sparse.mm(adj, support) if adj.is_sparse else adj @ support) + self.bias


class GMMNnetwork_GCN(nn.Module):
    """zs3/modeling/gmmn.py:52-67: gcn1 -> LeakyReLU(0.2) -> Dropout(0.5) -> gcn2; xavier_uniform weights, bias 0.01."""

    def __init__(self, noise_dim=300, embed_dim=300, hidden_size=256, feature_dim=256):
        super().__init__()
        self.gcn1 = GraphConvolution(noise_dim + embed_dim, hidden_size)
        self.relu = nn.LeakyReLU(0.2)
        self.dropout = nn.Dropout(p=0.5)
        self.gcn2 = GraphConvolution(hidden_size, feature_dim)
        for m in (self.gcn1, self.gcn2):
            nn.init.xavier_uniform_(m.weight)
            m.bias.data.fill_(0.01)

    def forward(self, embd, noise, adj_mat):
        x = self.gcn1(torch.cat((embd, noise), 1), adj_mat)
        return self.gcn2(self.dropout(self.relu(x)), adj_mat)
