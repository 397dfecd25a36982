"""GCN-context branch of ZS3 (train_context_GMMN_GCNcontext.py): the cluster graph of a label
map on the device, and what the training loop needs from it.

`construct_adj_mat` mirrors the reference function of the same name (:33-102) but keeps everything on the GPU: one kernel
(zs3_cluster_graph) replaces the pure-Python depth-first search (0.3-0.5 s per 129x129 image), the seed embeddings /
features are row gathers.  Returned objects differ from the reference's only in container type: tensors on the device
instead of numpy arrays / a scipy-built sparse tensor, and the cluster -> pixels dictionary is built lazily
(`ClusterGraph.pixel_index()`), because the training loop never reads it (:307-330, :399-425)."""

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


class ClusterGraphBatch:
    """Cluster graphs of a batch of label maps from ONE launch (zs3_cluster_graph_batch) and one read-back of the B cluster
    counts; `graph(i, emb_rows, feat_rows)` then assembles image i's ClusterGraph from device slices and two row gathers,
    without touching the host again."""

    def __init__(self, seg, max_clusters=1024):
        require_gpu(seg)
        b, h, w = seg.shape
        if h * w > lib().zs3_cluster_graph_max_pixels():
            raise ValueError(f"label map of {h}x{w} pixels exceeds the single-workgroup limit of zs3_cluster_graph")
        dev = seg.device
        cap = int(max_clusters)
        seg = seg.to(torch.int32).contiguous()
        self.cap = cap
        self.cmap = torch.empty((b, h, w), dtype=torch.int32, device=dev)
        self.seed = torch.zeros((b, cap), dtype=torch.int32, device=dev)
        self.labels = torch.zeros((b, cap), dtype=torch.int32, device=dev)
        self.ncl = torch.zeros(b, dtype=torch.int32, device=dev)
        self.adj = torch.zeros((b, cap, cap), dtype=torch.float32, device=dev)
        check(lib().zs3_cluster_graph_batch(P(seg), I(b), I(h), I(w), P(self.cmap), P(self.seed), P(self.labels), P(self.ncl),
                                            P(self.adj), I(cap), stream()), "zs3_cluster_graph_batch")
        self._counts = None

    @property
    def counts(self):
        if self._counts is None:
            self._counts = [int(v) for v in self.ncl.tolist()]     # the one host read
            if max(self._counts) > self.cap:
                raise ValueError(f"{max(self._counts)} clusters in a label map, max_clusters={self.cap}")
        return self._counts

    def graph(self, i, emb_rows, feat_rows=None):
        n = self.counts[i]
        seeds = self.seed[i, :n].long()
        emb = ops.gather_rows(emb_rows, seeds)
        feat = ops.gather_rows(feat_rows, seeds) if feat_rows is not None else None
        adj = self.adj[i, :n, :n].contiguous() if n > 1 else None
        return ClusterGraph(adj, self.cmap[i], self.labels[i, :n].long(), seeds, emb, feat)


def cluster_graph_rows(seg, emb_rows, feat_rows=None, max_clusters=2048):
    """seg: [H, W] label map (any dtype); emb_rows: [H*W, E], feat_rows: [H*W, D] or None -- pixel-major rows, the layout
    the GMMN step already holds.  -> ClusterGraph.  One kernel + one 4-byte read-back (the graph size decides the shapes
    of everything downstream) + two row gathers."""
    require_gpu(seg, emb_rows, feat_rows)
    batch = ClusterGraphBatch(seg.unsqueeze(0), max_clusters)
    return batch.graph(0, emb_rows.float(), feat_rows.float() if feat_rows is not None else None)


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
