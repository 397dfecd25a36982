"""GCN-context training step of ZS3 on MI355X -- the loop body of zs3/train_context_GMMN_GCNcontext.py:239-457
(BASELINE configs[4]).

It is the GMMN step (gmmn_trainer.GMMNStep: frozen-backbone features, per-(image, class) generator updates as hipGraph
replays, stitched features, CE/SGD step of `pred_conv`) plus

* the cluster graph of every label map at feature resolution (8-connected components, adjacency; :307-322): one launch for
  the whole batch (zs3_cluster_graph_batch, a workgroup per image) in place of the reference's pure-Python depth-first
  search (0.3-0.5 s per 129x129 image);
* per image, one update of the graph generator `GMMNnetwork_GCN` on the clusters (:399-418): embeddings / real features of the
  clusters' seed pixels, noise, two graph convolutions as MFMA row-GEMMs, the MMD between generated and real cluster features
  (N = number of clusters), Adam;
* per batch, a second CE term through `pred_conv` on the clusters' features, weighted by GCN_weight (:431-454), accumulated
  into the same gradients before the one SGD step.

Reference quirks kept: clusters of label 255 are nodes of the graph and their features go through the MMD (their CE targets
are ignored by the loss); a cluster's feature is its *seed* pixel's feature (with GCN_avg_feat the reference re-averages that
same seed feature, see gcn_context.construct_adj_mat); images with an unseen class train neither generator and contribute
generated cluster features; an image whose label map is a single cluster contributes nothing (adj_mat is None, :323,:399).
Host synchronisation: the B cluster counts (they decide tensor shapes) are read together with the step's class histogram; the
losses are read back once per step.
"""

import torch

from . import functional as Fz
from . import ops
from .gcn_context import ClusterGraphBatch
from .gmmn_trainer import GMMNStep, GMMNTrainer


# (Running the graph generator's per-image update on a second stream next to the GMMN update chain, and not cutting the chain
# after every image, were measured in round 2 -- +1.3 ms and noise, tools/probe/r2bb.sh, r2cc.sh -- and removed in round 3.)


class GCNContextStep(GMMNStep):
    _hook_reads_generator = True   # the graph update reads the GMMN generator: the captured update chain is cut after every image

    def __init__(self, model, generator, generator_GCN, optimizer, optimizer_generator, optimizer_generator_GCN, criterion,
                 criterion_generator=None, *, GCN_weight=0.1, GCN_avg_feat=False, max_clusters=1024, **kw):
        """criterion_generator: the MMD callable for the cluster update (default: zs3_amd GMMNLoss with GMMNStep's sigma).
        Remaining keywords: GMMNStep's (seen, unseen, noise_dim, ..., noise, group, context_aware)."""
        super().__init__(model, generator, optimizer, optimizer_generator, criterion, **kw)
        self.generator_GCN, self.optimizer_generator_GCN = generator_GCN, optimizer_generator_GCN
        if criterion_generator is None:
            from .utils.loss import GMMNLoss
            criterion_generator = GMMNLoss(sigma=list(self.sigma), cuda=True).build_loss()
        self.criterion_generator = criterion_generator
        self.GCN_weight, self.GCN_avg_feat, self.max_clusters = float(GCN_weight), bool(GCN_avg_feat), int(max_clusters)
        self._cluster_feats, self._cluster_labels, self._gcn_losses = [], [], []
        self.last_generator_GCN_loss = 0.0
        self.last_num_clusters = 0

    def _feature_shaping(self):
        # This step's loop is cut after every image (the graph update reads the generator) and carries the cluster work: the prefetched
        # feature pass is its critical path, not the loop -- every cap on the pass costs (same box, tools/probe/r6c.sh: lanes on and no
        # caps 36.90 / 36.75 ms per step; ASPP in sequence 37.27 / 37.15; strip launches on 224 workgroups 37.59 / 37.51, on 128: 40.95).
        return True, 0, 0

    def _replica_parameters(self):
        return list(self.generator.parameters()) + list(self.generator_GCN.parameters())

    def _replica_modules(self):
        return [self.generator, self.generator_GCN]

    # ---- per batch: every image's cluster graph in one launch (:307-322), counts read back with the class histogram
    def _before_images(self, label_maps):
        self._graphs = ClusterGraphBatch(label_maps, self.max_clusters)

    # ---- per image: one update of the graph generator on its clusters (:399-427)
    def _after_image(self, i, label_map, real_rows_i, has_unseen):
        return self._graph_update(i, real_rows_i, has_unseen)

    def _graph_update(self, i, real_rows_i, has_unseen):
        graph = self._graphs.graph(i, self._st["emb"], real_rows_i)
        if graph.adj is None:
            return
        k = graph.num_clusters
        self._cluster_labels.append(graph.labels)
        if self.noise == "cpu":
            z = torch.rand((k, self.noise_dim)).to(real_rows_i.device)        # :402, CPU generator like the reference
        else:
            z = ops.uniform((k, self.noise_dim), Fz.next_seed(), real_rows_i.device)
        train_it = not has_unseen
        self.optimizer_generator_GCN.zero_grad()
        with torch.set_grad_enabled(train_it):
            fake = self.generator_GCN(graph.embedding, z, graph.adj)
        if train_it:
            loss = self.criterion_generator(fake, graph.feature)
            loss.backward()
            self.optimizer_generator_GCN.step()
            self._gcn_losses.append(loss.detach().reshape(1))
        use_real = self.real_seen_features and not has_unseen
        self._cluster_feats.append(graph.feature if use_real else fake.detach())

    # ---- per batch: CE of the clusters' features through pred_conv, weighted (:436-454)
    def _extra_classifier_terms(self):
        if not self._cluster_feats:
            # the globally normalised CE all-reduces [sum(w), sum(w * nll)] on every rank that calls the criterion in its training
            # step: a rank without clusters still has to take part, with zeros.  (Not through the criterion on a dummy logit under
            # no_grad: "auto" is local there by design -- validation must not touch torch.distributed -- and this rank would skip
            # the collective the others issue.  ADVICE r5; tests/test_gpu_world2.py runs a rank without clusters.)
            from .utils.loss import ce_exchange_nothing
            owner = getattr(self.criterion, "__self__", None)
            ce_exchange_nothing(getattr(owner, "group", None), self._st["emb"].device)
            return
        feats = torch.cat(self._cluster_feats, 0)                               # [K, D] = NHWC [1, K, 1, D]
        k, d = feats.shape
        out = self.model.decoder.forward_class_prediction(ops.nchw(feats.view(1, k, 1, d)))     # [1, C, K, 1]
        target = torch.cat(self._cluster_labels).float().view(1, k, 1)
        (self.GCN_weight * self.criterion(out, target)).backward()

    def __call__(self, image, target, embedding=None, table=None, next_image=None):
        """-> (generator_loss_batch, generator_GCN_loss_batch, classifier_loss, logits)"""
        self._cluster_feats, self._cluster_labels, self._gcn_losses = [], [], []
        g_loss, c_loss, out = super().__call__(image, target, embedding=embedding, table=table, next_image=next_image)
        gcn_loss = float(torch.cat(self._gcn_losses).sum().item()) if self._gcn_losses else 0.0
        self.last_generator_GCN_loss = gcn_loss
        self.last_num_clusters = int(sum(t.shape[0] for t in self._cluster_labels))
        self._cluster_feats, self._cluster_labels, self._gcn_losses = [], [], []
        return g_loss, gcn_loss, c_loss, out


class GCNContextTrainer(GMMNTrainer):
    """Trainer with the attribute/driver surface of zs3/train_context_GMMN_GCNcontext.py:105-513 (`training(epoch, args)`),
    built from injected pieces like GMMNTrainer."""

    def __init__(self, args, model, generator, generator_GCN, optimizer, optimizer_generator, optimizer_generator_GCN,
                 criterion, train_loader, scheduler, writer=None, noise="device"):
        self.args, self.model, self.generator, self.generator_GCN = args, model, generator, generator_GCN
        self.optimizer, self.optimizer_generator = optimizer, optimizer_generator
        self.optimizer_generator_GCN = optimizer_generator_GCN
        self.criterion, self.train_loader, self.scheduler, self.writer = criterion, train_loader, scheduler, writer
        self.best_pred = 0.0
        self.step_fn = GCNContextStep(model, generator, generator_GCN, optimizer, optimizer_generator,
                                      optimizer_generator_GCN, criterion,
                                      GCN_weight=args.GCN_weight, GCN_avg_feat=args.GCN_avg_feat,
                                      seen=args.seen_classes_idx_metric, unseen=args.unseen_classes_idx_metric,
                                      noise_dim=args.noise_dim, embed_dim=args.embed_dim, feature_dim=args.feature_dim,
                                      batch_size_generator=args.batch_size_generator,
                                      real_seen_features=args.real_seen_features,
                                      context_aware=getattr(args, "context_aware", False), noise=noise)

    def training(self, epoch, args=None):
        train_loss = 0.0
        self.model.train()
        num_img_tr = len(self.train_loader)
        for i, sample in enumerate(self.train_loader):
            if len(sample["image"]) <= 1:
                continue
            image, target, embedding = sample["image"].cuda(), sample["label"].cuda(), sample["label_emb"].cuda()
            self.scheduler(self.optimizer, i, epoch, self.best_pred)
            g_loss, gcn_loss, c_loss, _ = self.step_fn(image, target, embedding)
            train_loss += c_loss
            if self.writer is not None:
                step = i + num_img_tr * epoch
                self.writer.add_scalar("train/total_loss_iter", c_loss, step)
                self.writer.add_scalar("train/generator_loss", g_loss, step)
                self.writer.add_scalar("train/generator_GCN_loss", gcn_loss, step)
        if self.writer is not None:
            self.writer.add_scalar("train/total_loss_epoch", train_loss, epoch)
        return train_loss
