"""GCN-context branch of ZS3 (train_context_GMMN_GCNcontext.py): the cluster graph of a label
map on the device, and what the training loop needs from it.

`construct_adj_mat` mirrors the reference function of the same name (:33-102) but keeps everything on the GPU: one kernel
(zs3_cluster_graph) replaces the pure-Python depth-first search (0.3-0.5 s per 129x129 image), the seed embeddings /
features are row gathers.  Returned objects differ from the reference's only in container type: tensors on the device
instead of numpy arrays / a scipy-built sparse tensor, and the cluster -> pixels dictionary is built lazily
(`ClusterGraph.pixel_index()`), because the training loop never reads it (:307-330, :399-425)."""
import ctypes

import torch

from . import ops
from ._lib import I, P, check, lib, require_gpu, stream


class ClusterGraph:
    """adj: dense [Nc, Nc] float 0/1 (None when there is a single cluster, like the reference's adj_mat);
    cluster_map: [H, W] int32; labels: [Nc] int64 (clsidx_2_lbl); seeds: [Nc] int64 flat pixel index of each cluster's
    first pixel in raster order; embedding / feature: [Nc, E] / [Nc, D] rows of the seed pixels."""

    def __init__(self, adj, cluster_map, labels, seeds, embedding, feature):
        self.adj, self.cluster_map, self.labels, self.seeds = adj, cluster_map, labels, seeds
        self.embedding, self.feature = embedding, feature

    @property
    def num_clusters(self):
        return int(self.labels.shape[0])

    def adj_sparse(self):
        """torch sparse COO tensor like sparse_mx_to_torch_sparse_tensor (:24-30) produces"""
        return None if self.adj is None else self.adj.to_sparse()

    def pixel_index(self):
        """clsidx_2_pixidx: {cluster id: [(i, j), ...]} (pixels in raster order, each once)"""
        cm = self.cluster_map.cpu()
        w = cm.shape[1]
        out = {c: [] for c in range(self.num_clusters)}
        for p, c in enumerate(cm.reshape(-1).tolist()):
            out[c].append((p // w, p % w))
        return out

    def as_reference_tuple(self):
        """(adj_mat, clsidx_2_pixidx, clsidx_2_lbl, embedding_GCN, feat_GCN) with the reference's container types"""
        feat = self.feature.cpu().numpy() if self.feature is not None else []
        return (self.adj_sparse(), self.pixel_index(), self.labels.cpu().tolist(), self.embedding.cpu().numpy(), feat)


def cluster_graph_rows(seg, emb_rows, feat_rows=None, max_clusters=2048):
    """seg: [H, W] label map (any dtype); emb_rows: [H*W, E], feat_rows: [H*W, D] or None -- pixel-major rows, the layout
    the GMMN step already holds.  -> ClusterGraph.  One kernel + one 4-byte read-back (the graph size decides the shapes
    of everything downstream) + two row gathers."""
    require_gpu(seg, emb_rows, feat_rows)
    h, w = seg.shape
    if h * w > lib().zs3_cluster_graph_max_pixels():
        raise ValueError(f"label map of {h}x{w} pixels exceeds the single-workgroup limit of zs3_cluster_graph")
    dev = seg.device
    seg = seg.to(torch.int32).contiguous()
    cmap = torch.empty((h, w), dtype=torch.int32, device=dev)
    cap = int(max_clusters)
    seed = torch.zeros(cap, dtype=torch.int32, device=dev)
    labels = torch.zeros(cap, dtype=torch.int32, device=dev)
    ncl = torch.zeros(1, dtype=torch.int32, device=dev)
    adj = torch.zeros((cap, cap), dtype=torch.float32, device=dev)
    check(lib().zs3_cluster_graph(P(seg), I(h), I(w), P(cmap), P(seed), P(labels), P(ncl), P(adj), I(cap), stream()),
          "zs3_cluster_graph")
    n = int(ncl.item())
    if n > cap:
        raise ValueError(f"{n} clusters in the label map, max_clusters={cap}")
    seeds = seed[:n].long()
    emb = ops.gather_rows(emb_rows.float().contiguous(), seeds)
    feat = ops.gather_rows(feat_rows.float().contiguous(), seeds) if feat_rows is not None else None
    return ClusterGraph(adj[:n, :n].contiguous() if n > 1 else None, cmap, labels[:n].long(), seeds, emb, feat)


def construct_adj_mat(segmap, embeddingmap, featmap=None, avg_feat=False, max_clusters=2048):
    """segmap: [H, W] class map; embeddingmap: [E, H, W]; featmap: [D, H, W] or None -- CUDA tensors.

    avg_feat: the reference "averages" a cluster's feature by re-adding its *seed* pixel's feature once per visited pixel
    (:70-73), i.e. the result is the seed feature up to float32 rounding noise whose exact value depends on how often the
    depth-first search revisits pixels; the device version returns the seed feature itself (within ~1e-5 relative of the
    reference's value; bit-identical for avg_feat=False)."""
    require_gpu(segmap, embeddingmap, featmap)
    h, w = segmap.shape
    emb_rows = embeddingmap.reshape(embeddingmap.shape[0], h * w).t().contiguous()
    feat_rows = featmap.reshape(featmap.shape[0], h * w).t().contiguous() if featmap is not None else None
    return cluster_graph_rows(segmap, emb_rows, feat_rows, max_clusters)
