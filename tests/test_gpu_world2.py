"""World-size-2 run of the PRODUCT's N>1 path on ONE MI355X: two processes share cuda:0 and talk through the gloo backend on
device tensors (RCCL refuses two ranks on one device; the collectives' call sites are the same for either backend).

  supervised: DeepLab(sync_bn=True) -- which arms GradSync by itself at its first training forward (zero-copy buckets, hooks,
              end-of-backward join) -- + the CE, which normalises over both shards by itself: the worker's body names no
              GradSync and no group.  Two DIFFERENT shards must equal the single-process step on the concatenated batch -- logits (SyncBN
              statistics are global), loss, every gradient, the BN running statistics.  SyncBN is on by construction: the
              script makes no enable call (zs3/modeling/sync_batchnorm/batchnorm.py:46-89, train_pascal.py:279).
  GMMN step : GMMNStep (group="auto", the default) on two shards ends with identical generator / pred_conv weights on both ranks, the
              generator equal to the average of the two single-process runs on the shards (replicas with
              parameter averaging), the classifier loss equal to the CE normalised over both shards.
  GCN-context: GCNContextStep (likewise): both generators (GMMN and graph) identical on both ranks and equal to the mean of the
              two single-shard runs.
"""
import os
import socket
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _tamed_model(sync_bn):
    from zs3_amd.modeling.deeplab import DeepLab
    torch.manual_seed(1)
    m = DeepLab(num_classes=21, pretrained=False, sync_bn=sync_bn)
    for name, mod in m.named_modules():
        if name.endswith("bn3"):
            mod.weight.data.fill_(0.1)
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    return m


def _supervised(dev, image, label, validate=None):
    """The body a reference training script has -- model, criterion, forward, loss, backward -- and nothing else: no gradient-sync
    object, no process-group argument, no SyncBN switch.  Under torch.distributed with two ranks the model arms its
    gradient all-reduce at this first training forward, SyncBN exchanges its sums, and the criterion normalises over both
    shards, all by construction; in a single process the same lines are the plain step."""
    from zs3_amd.utils.loss import SegmentationLosses
    m = _tamed_model(sync_bn=True).to(dev).train()
    w = torch.ones(21, device=dev)
    w[[10, 14]] = 100.0
    crit = SegmentationLosses(weight=w, cuda=True).build_loss("ce")
    out = m(image)
    loss = crit(out, label)
    loss.backward()
    torch.cuda.synchronize()
    keys = ("decoder.pred_conv.weight", "decoder.last_conv.0.weight", "aspp.aspp2.atrous_conv.weight", "aspp.bn1.weight",
            "backbone.layer3.5.conv2.weight", "backbone.layer3.5.bn2.bias", "backbone.layer1.0.conv1.weight",
            "backbone.conv1.weight", "backbone.bn1.weight")
    params = dict(m.named_parameters())
    res = {"logits": out.detach().cpu(), "loss": float(loss.item()),
           "grads": {k: params[k].grad.detach().float().cpu().clone() for k in keys},
           "running": {k: v.detach().cpu().clone() for k, v in m.state_dict().items()
                       if k in ("backbone.bn1.running_mean", "backbone.layer3.5.bn2.running_var", "aspp.bn1.running_var",
                                "decoder.last_conv.1.running_mean")}}
    sync = getattr(m, "_zs3_grad_sync", None)     # what the model armed by itself (None in a single process)
    if validate is not None:
        # a validation pass on ONE rank only (train_pascal.py:115-172 runs model + criterion under no_grad; a script may well do it
        # on rank 0 alone): eval-mode BatchNorm and the criterion without a gradient are local by construction -- no collective, no
        # deadlock, this rank's own loss value -- and an armed model deep-copies / pickles (its GradSync stays behind)
        import copy
        import io
        clone = copy.deepcopy(m)
        assert getattr(clone, "_zs3_grad_sync", None) is None
        torch.save(m, io.BytesIO())
        m.eval()
        with torch.no_grad():
            res["val_loss"] = float(crit(m(validate[0]), validate[1]).item())
            res["val_loss_clone"] = float(crit(clone.eval()(validate[0]), validate[1]).item())
        m.train()
    if sync is not None:
        res["bytes"] = sync.bytes_reduced
        from zs3_amd.parallel import disarm_data_parallel
        disarm_data_parallel(m)
    return res


def _gmmn(dev, image, label, table, steps=1):
    from zs3_amd import functional as Fz
    from zs3_amd.gmmn_trainer import GMMNStep
    from zs3_amd.modeling.gmmn import GMMNnetwork
    from zs3_amd.optim import SGD, Adam
    from zs3_amd.utils.loss import SegmentationLosses
    seen = [c for c in range(21) if c not in (10, 14)]
    m = _tamed_model(sync_bn=False).to(dev).train()
    torch.manual_seed(4)
    gen = GMMNnetwork(300, 300, 256, 256).to(dev).train()
    for mod in gen.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    Fz.manual_seed(77)
    w = torch.ones(21, device=dev)
    w[[10, 14]] = 100.0
    groups = [{"params": m.get_1x_lr_params(), "lr": 0.007}, {"params": m.get_10x_lr_params(), "lr": 0.07}]
    opt, opt_g = SGD(groups, momentum=0.9, weight_decay=5e-4), Adam(gen.parameters(), lr=2e-4)
    crit = SegmentationLosses(weight=w, cuda=True).build_loss("ce")
    step = GMMNStep(m, gen, opt, opt_g, crit, seen=seen, unseen=[10, 14], noise="cpu")
    losses = []
    for it in range(steps):
        torch.manual_seed(21 + it)
        gl, cl, _ = step(image, label, table=table)
        losses.append((gl, cl))
    torch.cuda.synchronize()
    fp = step.__dict__.get("_feature_plan")
    return {"losses": losses, "gen": [p.detach().cpu().clone() for p in gen.parameters()],
            "pred_w": m.decoder.pred_conv.weight.detach().cpu().clone(), "pred_b": m.decoder.pred_conv.bias.detach().cpu().clone(),
            "feature_plan": (fp.eager_calls, fp.recordings, fp.replays) if fp is not None else None}


def _gcn(dev, image, label, table):
    """GCNContextStep (train_context_GMMN_GCNcontext.py:239-457): the graph generator is a second replica"""
    from zs3_amd import functional as Fz
    from zs3_amd.gcn_trainer import GCNContextStep
    from zs3_amd.modeling.gmmn import GMMNnetwork, GMMNnetwork_GCN
    from zs3_amd.optim import SGD, Adam
    from zs3_amd.utils.loss import SegmentationLosses
    seen = [c for c in range(21) if c not in (10, 14)]
    m = _tamed_model(sync_bn=False).to(dev).train()
    torch.manual_seed(4)
    gen, gcn = GMMNnetwork(300, 300, 256, 256).to(dev).train(), GMMNnetwork_GCN(300, 300, 256, 256).to(dev).train()
    gen.model[2].p = 0.0
    gcn.dropout.p = 0.0
    Fz.manual_seed(78)
    w = torch.ones(21, device=dev)
    w[[10, 14]] = 100.0
    groups = [{"params": m.get_1x_lr_params(), "lr": 0.007}, {"params": m.get_10x_lr_params(), "lr": 0.07}]
    opt, opt_g, opt_c = SGD(groups, momentum=0.9, weight_decay=5e-4), Adam(gen.parameters(), lr=2e-4), Adam(gcn.parameters(), lr=2e-4)
    crit = SegmentationLosses(weight=w, cuda=True).build_loss("ce")
    step = GCNContextStep(m, gen, gcn, opt, opt_g, opt_c, crit, seen=seen, unseen=[10, 14], noise="cpu", GCN_weight=0.1)
    torch.manual_seed(22)
    gl, gcl, cl, _ = step(image, label, table=table)
    torch.cuda.synchronize()
    return {"losses": (gl, gcl, cl), "gen": [p.detach().cpu().clone() for p in gen.parameters()],
            "gcn": [p.detach().cpu().clone() for p in gcn.parameters()], "pred_w": m.decoder.pred_conv.weight.detach().cpu().clone()}


def _batch():
    import zs3_oracle as zo
    b = zo.make_synthetic_batch(4, 65, seed=31, with_label_emb=False)
    label = b["label"].clone()
    seen_only = label.clone()          # GMMN part: no unseen pixels, so no generated features enter the classifier loss
    seen_only[(seen_only == 10) | (seen_only == 14)] = 3
    table = torch.nn.functional.normalize(torch.randn(21, 300, generator=torch.Generator().manual_seed(5)), dim=1)
    return b["image"], label, seen_only, table


def _worker(rank, world, port, outdir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    import datetime
    # (a timeout: collectives that do not pair up -- the failure the lonely-rank case below guards against -- raise instead of hanging)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
    try:
        image, label, seen_only, table = _batch()
        sl = slice(2 * rank, 2 * rank + 2)
        val = (image[:2].to(dev), label[:2].to(dev)) if rank == 0 else None     # rank-0-only validation: must not hang
        res = {"sup": _supervised(dev, image[sl].to(dev), label[sl].to(dev), validate=val),
               "gmmn": _gmmn(dev, image[sl].to(dev), seen_only[sl].to(dev), table.to(dev)),
               "gcn": _gcn(dev, image[sl].to(dev), seen_only[sl].to(dev), table.to(dev))}
        # a rank WITHOUT clusters (every label map one connected component: adj_mat is None, :323) still takes part in the
        # criterion's all-reduce that the other rank issues from its cluster term -- with zeros, not through a no_grad dummy call
        lonely = torch.full_like(seen_only[sl], 3) if rank == 1 else seen_only[sl]
        res["gcn_lonely"] = _gcn(dev, image[sl].to(dev), lonely.to(dev), table.to(dev))
        # four GMMN steps: in a single process the feature pass is recorded on its third call and replayed on the fourth; with two
        # ranks over gloo its SyncBN-free forward has no collective, but the step's do go through torch.distributed -- the plan
        # machinery must leave such a process alone (zs3_amd.plan.collectives_recordable)
        res["gmmn4"] = _gmmn(dev, image[sl].to(dev), seen_only[sl].to(dev), table.to(dev), steps=4)
        torch.save(res, os.path.join(outdir, f"rank{rank}.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def test_two_ranks_on_one_device_equal_the_single_process_step():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    dev = torch.device("cuda:0")
    image, label, seen_only, table = _batch()
    with tempfile.TemporaryDirectory() as td:
        mp.spawn(_worker, args=(2, _free_port(), td), nprocs=2, join=True)
        r = [torch.load(os.path.join(td, f"rank{k}.pt")) for k in range(2)]
    # ---------------- supervised: two shards + SyncBN + GradSync + global CE == one process on the whole batch
    one = _supervised(dev, image.to(dev), label.to(dev), validate=(image[:2].to(dev), label[:2].to(dev)))
    # rank 0 validated alone (no collective, no hang) and got its own shard's loss: the value a single process computes on that shard
    # (same parameters; running statistics after one synchronised training forward are the global ones)
    assert abs(r[0]["sup"]["val_loss"] - one["val_loss"]) < 2e-4 * abs(one["val_loss"]), (r[0]["sup"]["val_loss"], one["val_loss"])
    assert abs(r[0]["sup"]["val_loss_clone"] - r[0]["sup"]["val_loss"]) < 1e-6 * abs(one["val_loss"]) and "val_loss" not in r[1]["sup"]
    both = torch.cat([r[0]["sup"]["logits"], r[1]["sup"]["logits"]], 0)
    assert _rel(both, one["logits"]) < 2e-4                 # global BN statistics in every one of the 113 layers
    for k in range(2):
        assert abs(r[k]["sup"]["loss"] - one["loss"]) < 1e-5 * abs(one["loss"])
        assert r[k]["sup"]["bytes"] == 4 * 59344309         # every parameter went through the all-reduce once
        for name, gref in one["grads"].items():
            assert _rel(r[k]["sup"]["grads"][name], gref) < 5e-3, (k, name)
        for name, v in one["running"].items():
            assert _rel(r[k]["sup"]["running"][name], v) < 1e-4, (k, name)
    for name in one["grads"]:                               # the reduced gradients are bit-identical on both ranks
        assert torch.equal(r[0]["sup"]["grads"][name], r[1]["sup"]["grads"][name]), name
    # ---------------- GMMN step: replicas with parameter averaging
    for a, b in zip(r[0]["gmmn"]["gen"], r[1]["gmmn"]["gen"]):
        assert torch.equal(a, b)
    assert torch.equal(r[0]["gmmn"]["pred_w"], r[1]["gmmn"]["pred_w"])
    solo = [_gmmn(dev, image[2 * k:2 * k + 2].to(dev), seen_only[2 * k:2 * k + 2].to(dev), table.to(dev))
            for k in range(2)]
    for p2, pa, pb in zip(r[0]["gmmn"]["gen"], solo[0]["gen"], solo[1]["gen"]):
        assert _rel(p2, (pa + pb) / 2) < 1e-5               # generator = mean of the two ranks' locally trained replicas
    # no unseen pixels -> the classifier loss does not depend on the generator, and (plain BN: per-rank statistics) each
    # rank sees the features its shard sees alone: the globally normalised CE must be sum(w*nll) / sum(w) / B over BOTH
    # shards (loss.py:33-46 on the gathered batch), rebuilt here from the two single-shard losses and their valid-pixel counts
    n = [float((seen_only[2 * k:2 * k + 2] != 255).sum()) for k in range(2)]
    expect = (solo[0]["losses"][0][1] * 2 * n[0] + solo[1]["losses"][0][1] * 2 * n[1]) / (n[0] + n[1]) / 4
    for k in range(2):
        assert abs(r[k]["gmmn"]["losses"][0][1] - expect) < 1e-5 * abs(expect), (r[k]["gmmn"]["losses"], expect)
    assert not torch.equal(r[0]["gmmn"]["pred_w"], solo[0]["pred_w"])     # the classifier step used both shards' gradients
    assert r[0]["gmmn4"]["feature_plan"] == r[1]["gmmn4"]["feature_plan"] == (4, 0, 0)        # two ranks over gloo: always eager
    solo4 = _gmmn(dev, image[:2].to(dev), seen_only[:2].to(dev), table.to(dev), steps=4)
    assert solo4["feature_plan"] == (2, 1, 1)                                                   # one process: recorded, replayed
    # ---------------- GCN-context step (configs[4] flow): two replicated generators, same exchange
    for key in ("gen", "gcn"):
        for a, b in zip(r[0]["gcn"][key], r[1]["gcn"][key]):
            assert torch.equal(a, b), key
    assert torch.equal(r[0]["gcn"]["pred_w"], r[1]["gcn"]["pred_w"])
    solo_g = [_gcn(dev, image[2 * k:2 * k + 2].to(dev), seen_only[2 * k:2 * k + 2].to(dev), table.to(dev))
              for k in range(2)]
    for key in ("gen", "gcn"):
        for p2, pa, pb in zip(r[0]["gcn"][key], solo_g[0][key], solo_g[1][key]):
            assert _rel(p2, (pa + pb) / 2) < 1e-5, key
    assert not torch.equal(r[0]["gcn"]["gcn"][0], solo_g[0]["gcn"][0])
    # ---------------- one rank without clusters: no hang, no mis-paired collective -- both ranks end on the same weights
    for key in ("gen", "gcn"):
        for a, b in zip(r[0]["gcn_lonely"][key], r[1]["gcn_lonely"][key]):
            assert torch.equal(a, b), key
    assert torch.equal(r[0]["gcn_lonely"]["pred_w"], r[1]["gcn_lonely"]["pred_w"])
    assert all(torch.isfinite(torch.tensor(r[k]["gcn_lonely"]["losses"])).all() for k in range(2))
