"""CPU-side checks: the C-ABI library loads and exports exactly what include/zs3hip.h declares, the product
refuses to run without a GPU (no CPU fallback), host logic (LR schedule, synthetic batches, module surface)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libpath():
    from zs3_amd import build
    return build.build(verbose=False)


def test_library_exports_every_declared_symbol(libpath):
    header = open(os.path.join(ROOT, "include", "zs3hip.h")).read()
    declared = set(re.findall(r"^(?:int|long)\s+(zs3_\w+)\s*\(", header, flags=re.M))
    assert len(declared) >= 30
    lib = ctypes.CDLL(libpath)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in zs3hip.h but not exported"
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", libpath], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (zs3_\w+)", out))
    assert exported == declared, (exported ^ declared)


def test_host_side_planners_need_no_gpu(libpath):
    lib = ctypes.CDLL(libpath)
    assert lib.zs3_conv_igemm_mtiles(266256, 256, 1) == (266256 + 127) // 128
    assert lib.zs3_conv_igemm_mtiles(100, 256, 4) == 2
    sk, ws = ctypes.c_int(0), ctypes.c_long(0)
    assert lib.zs3_conv_wgrad_plan(266256, 129, 256, 256, 9, ctypes.byref(sk), ctypes.byref(ws)) == 0
    assert sk.value >= 1 and (ws.value == 0 or ws.value == sk.value * 256 * 9 * 256)
    ch, rpb = ctypes.c_int(0), ctypes.c_int(0)
    lib.zs3_colstats_plan(1000, 64, ctypes.byref(ch), ctypes.byref(rpb))
    assert ch.value * rpb.value >= 1000
    assert lib.zs3_ce_ws_doubles() == 2048
    # argument validation happens before any launch
    assert lib.zs3_conv_igemm(None, None, None, None, None, None, None, 1, 8, 8, 8, 8, 30, 8, 8, 1, 1, 1, 0, 0, 1, 8, 8, 0, 0,
                              ctypes.c_float(0.2), 0, 0, 3, 0, None, None) == -1


def test_product_has_no_cpu_fallback():
    from zs3_amd._lib import Zs3HipError
    from zs3_amd.modeling.deeplab import DeepLab
    from zs3_amd.utils.loss import GMMNLoss, SegmentationLosses
    torch.manual_seed(0)
    m = DeepLab(num_classes=21, pretrained=False, sync_bn=False).eval()
    with pytest.raises((Zs3HipError, RuntimeError)):
        m(torch.randn(1, 3, 33, 33))
    with pytest.raises(Zs3HipError):
        SegmentationLosses().build_loss("ce")(torch.randn(1, 21, 9, 9), torch.zeros(1, 9, 9))
    with pytest.raises(Zs3HipError):
        GMMNLoss().build_loss()(torch.randn(8, 16), torch.randn(8, 16))
    import zs3_amd
    src = "".join(open(os.path.join(ROOT, "zs3_amd", f)).read() for f in os.listdir(os.path.join(ROOT, "zs3_amd")) if f.endswith(".py"))
    assert "zs3_oracle" not in src and "import oracle" not in src  # the checker is never reachable from the product


def test_module_surface_matches_reference_interface(golden):
    from zs3_amd.modeling.deeplab import DeepLab
    from zs3_amd.modeling.gmmn import GMMNnetwork
    from zs3_amd.modeling.sync_batchnorm.batchnorm import SynchronizedBatchNorm2d
    from zs3_amd.modeling.sync_batchnorm.replicate import patch_replication_callback
    g = golden("init.npz")
    torch.manual_seed(1)
    m = DeepLab(num_classes=21, pretrained=False, sync_bn=False)
    sd = m.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["names21"]]
    for k, r in zip(g["names21"], g["stats21"]):   # same seed => same weights as the reference constructor
        t = sd[str(k)].double().reshape(-1)
        assert abs(t.sum().item() - r[0]) <= 1e-9 * max(1.0, r[1]) and abs(t.abs().sum().item() - r[1]) <= 1e-9 * max(1.0, r[1]), k
    assert sum(p.numel() for p in m.get_1x_lr_params()) == 42500160
    assert sum(p.numel() for p in m.get_10x_lr_params()) == 16844149
    for attr in ("backbone", "aspp", "decoder", "forward_before_class_prediction", "forward_class_prediction",
                 "forward_before_last_conv_finetune", "forward_class_last_conv_finetune", "freeze_bn"):
        assert hasattr(m, attr)
    assert hasattr(m.decoder, "forward_class_prediction") and hasattr(m.decoder, "pred_conv")
    torch.manual_seed(1)
    m60 = DeepLab(num_classes=60, pretrained=False, sync_bn=True, global_avg_pool_bn=False)
    assert list(m60.state_dict().keys()) == [str(k) for k in g["names60"]]
    assert isinstance(m60.backbone.bn1, SynchronizedBatchNorm2d) and isinstance(m60.backbone.bn1, torch.nn.BatchNorm2d)
    m60.train()
    m60.freeze_bn()
    assert not m60.backbone.bn1.training and m60.backbone.conv1.training
    torch.manual_seed(1)
    gen = GMMNnetwork(300, 300, 256, 256)
    assert list(gen.state_dict().keys()) == [str(k) for k in g["names_g"]]
    for k, r in zip(g["names_g"], g["stats_g"]):
        assert abs(gen.state_dict()[str(k)].double().abs().sum().item() - r[1]) <= 1e-9 * max(1.0, r[1])
    torch.manual_seed(3)
    gen2 = GMMNnetwork(300, 300, 0, 256, semantic_reconstruction=True)
    assert list(gen2.state_dict().keys()) == [str(k) for k in g["names_g2"]]
    patch_replication_callback(torch.nn.DataParallel(torch.nn.Linear(2, 2), device_ids=None) if torch.cuda.is_available()
                               else _FakeDP())
    with pytest.raises(AssertionError):
        patch_replication_callback(torch.nn.Linear(2, 2))
    # the reference's own `--gpu-ids 0,1` (train_pascal.py:88-93): single-process replication over several devices is refused with
    # the one-process-per-GPU recipe in the message, instead of sending ctypes-backed modules through DataParallel.replicate
    two = _FakeDP()
    two.device_ids = [0, 1]
    with pytest.raises(RuntimeError, match="torchrun"):
        patch_replication_callback(two)
    one = _FakeDP()
    one.device_ids = [0]
    patch_replication_callback(one)
    # a "pretrained" ImageNet checkpoint with the 7-character `module.` prefix loads (resnet.py:211-226)
    import tempfile
    ck = {"state_dict": {"module." + k: torch.full_like(v, 0.5) for k, v in m.backbone.state_dict().items() if "layer1.0.conv1" in k}}
    with tempfile.NamedTemporaryFile(suffix=".pth") as f:
        torch.save(ck, f.name)
        m2 = DeepLab(num_classes=21, pretrained=True, sync_bn=False, imagenet_pretrained_path=f.name)
    assert torch.all(m2.backbone.layer1[0].conv1.weight == 0.5)
    assert m2.backbone.layer1[0].conv2.weight.is_contiguous(memory_format=torch.channels_last)


class _FakeDP(torch.nn.DataParallel):
    def __init__(self):
        torch.nn.Module.__init__(self)


def test_lr_scheduler_and_synthetic_batch(golden):
    from zs3_amd.utils.lr_scheduler import LR_Scheduler
    from zs3_amd.utils.synthetic import make_batch
    import zs3_oracle as zo
    g = golden("misc.npz")
    sch = LR_Scheduler("poly", 0.007, 3, 11, verbose=False)
    opt = torch.optim.SGD([{"params": [torch.nn.Parameter(torch.zeros(1))]}, {"params": [torch.nn.Parameter(torch.zeros(1))]}], lr=0.1)
    lrs = []
    for ep in range(3):
        for it in range(11):
            sch(opt, it, ep, 0.0)
            lrs.append([pg["lr"] for pg in opt.param_groups])
    assert np.allclose(lrs, g["poly_lrs"], rtol=1e-12)
    a = make_batch(4, 65, seed=5, with_label_emb=True, device="cpu")
    b = zo.make_synthetic_batch(4, 65, seed=5, with_label_emb=True)
    for k in ("image", "label", "table", "label_emb"):
        assert torch.equal(a[k], b[k]), k   # the product's generator and the oracle's produce identical batches


def test_evaluator_matches_the_reference_evaluator():
    """zs3_amd.utils.metrics.Evaluator (host path) against zs3/utils/metrics.py run in the build container
    (tests/golden/misc.npz): same confusion matrix, same return tuples with and without the seen/unseen split"""
    import os
    import numpy as np
    from zs3_amd.utils.metrics import Evaluator
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "misc.npz"))
    gt, pred = g["cm_gt"].astype(np.int64), g["cm_pred"].astype(np.int64)
    seen = [c for c in range(21) if c not in (10, 14)]
    ev = Evaluator(21, seen, [10, 14])
    ev.add_batch(gt[:2], pred[:2])
    ev.add_batch(torch.from_numpy(gt[2:]), torch.from_numpy(pred[2:]))     # CPU tensors take the host path too
    assert np.array_equal(ev.confusion_matrix, g["cm"])
    miou, by_class, miou_seen, miou_unseen = ev.Mean_Intersection_over_Union()
    assert miou == float(g["miou"]) and miou_seen == float(g["miou_seen"]) and miou_unseen == float(g["miou_unseen"])
    assert np.array_equal(by_class, g["miou_by_class"], equal_nan=True)
    assert np.array_equal(np.array(ev.Pixel_Accuracy()), g["pix_acc"])
    acc_cls, acc_by_class, acc_seen, acc_unseen = ev.Pixel_Accuracy_Class()
    assert np.array_equal(np.array([acc_cls, acc_seen, acc_unseen]), g["pix_acc_class"])
    assert np.array_equal(acc_by_class, g["pix_acc_by_class"], equal_nan=True)
    assert np.array_equal(np.array(ev.Frequency_Weighted_Intersection_over_Union()), g["fwiou"])
    plain = Evaluator(21)
    plain.add_batch(gt, pred)
    got = [plain.Pixel_Accuracy(), plain.Pixel_Accuracy_Class()[0], plain.Mean_Intersection_over_Union()[0],
           plain.Frequency_Weighted_Intersection_over_Union()]
    assert np.array_equal(np.array(got), g["plain"])
    plain.reset()
    assert plain.confusion_matrix.sum() == 0


def test_gcn_context_surface(libpath):
    """GCN-context pieces: state-dict keys / [in, out] weight layout of the pygcn-based generator
    (zs3/modeling/gmmn.py:52-67), the step object's constructor contract, host-side limits of the cluster-graph kernel"""
    from zs3_amd.gcn_trainer import GCNContextStep
    from zs3_amd.gmmn_trainer import GMMNStep
    from zs3_amd.modeling.deeplab import DeepLab
    from zs3_amd.modeling.gmmn import GMMNnetwork, GMMNnetwork_GCN
    from zs3_amd.optim import SGD, Adam
    from zs3_amd.utils.loss import SegmentationLosses
    lib = ctypes.CDLL(libpath)
    assert lib.zs3_cluster_graph_max_pixels() >= 129 * 129
    assert lib.zs3_cluster_graph_batch(None, 0, 4, 4, None, None, None, None, None, 8, None) == 0      # empty batch: no launch
    assert lib.zs3_cluster_graph_batch(None, 2, 1000, 1000, None, None, None, None, None, 8, None) == -1   # too many pixels
    torch.manual_seed(0)
    gcn = GMMNnetwork_GCN(300, 300, 256, 256)
    sd = gcn.state_dict()
    assert list(sd) == ["gcn1.weight", "gcn1.bias", "gcn2.weight", "gcn2.bias"]
    assert tuple(sd["gcn1.weight"].shape) == (600, 256) and tuple(sd["gcn2.weight"].shape) == (256, 256)
    assert torch.all(sd["gcn1.bias"] == 0.01) and abs(sd["gcn1.weight"].abs().max().item() - (6 / 856) ** 0.5) < 2e-3
    m = DeepLab(num_classes=21, pretrained=False, sync_bn=False)
    gen = GMMNnetwork(300, 300, 256, 256)
    opt = SGD([{"params": m.get_1x_lr_params(), "lr": 0.007}, {"params": m.get_10x_lr_params(), "lr": 0.07}], momentum=0.9)
    crit = SegmentationLosses(cuda=False, group=True).build_loss("ce")
    step = GCNContextStep(m, gen, gcn, opt, Adam(gen.parameters(), lr=2e-4), Adam(gcn.parameters(), lr=2e-4), crit,
                          seen=range(19), unseen=[19, 20], GCN_weight=0.25, GCN_avg_feat=True, context_aware=True)
    assert isinstance(step, GMMNStep) and step.GCN_weight == 0.25 and step.context_aware
    assert step.grad_reduce == "sum"                       # the criterion normalises globally (group=...)
    assert len(step._replica_parameters()) == 8            # both generators are averaged over the ranks
    with pytest.raises(ValueError):                        # context_aware feeds the mean embedding in place of the noise
        GMMNStep(m, GMMNnetwork(100, 300, 256, 256), opt, None, crit, seen=[0], unseen=[1], noise_dim=100, context_aware=True)
    with pytest.raises(ValueError):
        step(torch.zeros(2, 3, 65, 65), torch.zeros(2, 65, 65))      # neither `embedding` nor `table`


def test_evaluator_seen_unseen_matches_the_reference(golden):
    """zs3/utils/metrics.py:88-196 (imported by eval_pascal.py:14): overall / seen / unseen / per-class score tuples on the
    fixture's four label maps equal the reference's, NaN pattern included (classes absent from the ground truth)."""
    import warnings
    from zs3_amd.utils.metrics import Evaluator_seen_unseen
    g = golden("seen_unseen.npz")
    gt, pred = list(g["gt"].astype(np.int64)), list(g["pred"].astype(np.int64))

    def flat(x, out):
        if isinstance(x, (tuple, list)):
            for y in x:
                flat(y, out)
        else:
            out.append(float(x))
        return np.array(out)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        split = flat(Evaluator_seen_unseen(21, [10, 14]).label_accuracy_score(gt, pred, by_class=True), [])
        plain = flat(Evaluator_seen_unseen(21, None).label_accuracy_score(gt, pred), [])
    assert split.shape == g["split"].shape and np.array_equal(np.isnan(split), np.isnan(g["split"]))
    assert np.allclose(split, g["split"], rtol=1e-12, atol=0, equal_nan=True)
    assert np.allclose(plain, g["plain"], rtol=1e-12, atol=0)
    ev = Evaluator_seen_unseen(21, [10, 14])
    h = ev._fast_hist(gt[0].ravel(), pred[0].ravel(), 21, target="unseen", unseen=[10, 14])
    assert h.sum() == np.isin(gt[0], [10, 14]).sum() and h[[c for c in range(21) if c not in (10, 14)]].sum() == 0


def test_binding_declares_every_signature_from_the_header(libpath):
    """zs3_amd/_lib.py derives ctypes argtypes / restype for every exported function from include/zs3hip.h: a call with the
    wrong number or kind of arguments fails in Python instead of corrupting the callee's stack."""
    from zs3_amd import _lib
    handle = ctypes.CDLL(libpath)
    _lib._declare(handle)
    header = open(os.path.join(ROOT, "include", "zs3hip.h")).read()
    declared = set(re.findall(r"^(?:int|long)\s+(zs3_\w+)\s*\(", header, flags=re.M))
    for name in declared:
        fn = getattr(handle, name)
        assert fn.argtypes is not None, name
    assert handle.zs3_ce_ws_doubles.argtypes == []
    assert len(handle.zs3_affine_act.argtypes) == 20          # x .. mask_out, drop_p, drop_seed, io, stream
    with pytest.raises((ctypes.ArgumentError, TypeError)):
        handle.zs3_colstats_plan(1000, 64)                      # two of four arguments
    with pytest.raises((ctypes.ArgumentError, TypeError)):
        handle.zs3_conv_igemm_mtiles("not an int", 256, 1)


def test_bench_refuses_to_run_without_the_devices_it_was_asked_for():
    """`python bench.py --gpus N` spawns its own ranks only when the node has N devices, and the product never falls back to
    the CPU: both refusals are loud (this container has no GPU)."""
    import subprocess
    import sys
    if torch.cuda.is_available():
        pytest.skip("needs a machine without GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "only 0 GPU(s) visible" in (r.stderr + r.stdout)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)


def test_io_argument_sits_before_the_stream_on_every_activation_entry_point():
    """Round 4, the 2-byte mode at the C ABI: every entry point whose tensors are activations carries `int io` as its last
    argument before `void* stream` (bit 0: inputs are bf16, bit 1: outputs are); entry points that only see fp32 state
    (optimizers, finalizes, weight preparation, losses on fp32 class scores) do not."""
    header = re.sub(r"/\*.*?\*/", " ", open(os.path.join(ROOT, "include", "zs3hip.h")).read(), flags=re.S)
    sigs = {name: [a.strip() for a in args.split(",")] for _, name, args in
            re.findall(r"\b(int|long)\s+(zs3_\w+)\s*\(([^)]*)\)\s*;", header)}
    with_io = {n for n, a in sigs.items() if len(a) >= 2 and a[-1] == "void* stream" and a[-2] == "int io"}
    for name in ("zs3_conv_igemm", "zs3_conv_igemm_in", "zs3_conv_igemm_bnstats", "zs3_conv_wgrad", "zs3_conv_wgrad_strip",
                 "zs3_conv_wgrad_pw", "zs3_affine_act", "zs3_bn_act_bwd", "zs3_bn_bwd_stats", "zs3_maxpool_fwd", "zs3_maxpool_bwd",
                 "zs3_bilinear_fwd", "zs3_bilinear_bwd", "zs3_sum_n", "zs3_group_colsum", "zs3_colstats", "zs3_dropout",
                 "zs3_pad_rows"):
        assert name in with_io, name
    assert len(with_io) == 18
    for name in ("zs3_sgd_multi", "zs3_adam_step", "zs3_bn_fwd_finalize", "zs3_bn_bwd_finalize", "zs3_prep_weight",
                 "zs3_prep_weight_f16fwd", "zs3_prep_weight_f32", "zs3_mmd_fwd"):
        assert name in sigs and name not in with_io, name
    assert "#define ZS3_IO_IN16 1" in open(os.path.join(ROOT, "include", "zs3hip.h")).read()


def test_forward_arithmetic_rule_and_loss_log():
    """Host logic of round 4 that needs no GPU: (1) which fused layers multiply f16x3 in their forward pass -- batch-statistics
    BatchNorm under the default arithmetic and fp32 storage, nothing else (functional.forward_is_f16x3); (2) LossLog hands every
    loss value back exactly once, in order, one push late."""
    from zs3_amd import functional as Fz, ops
    from zs3_amd.base_trainer import LossLog
    train, frozen = {"training": True}, {"training": False}
    assert ops.PREC_DEFAULT == 3 and ops.ACT_DTYPE == torch.float32
    assert Fz.forward_is_f16x3(None, train) and Fz.forward_is_f16x3(3, train)
    assert not Fz.forward_is_f16x3(None, frozen) and not Fz.forward_is_f16x3(None, None) and not Fz.forward_is_f16x3(1, train)
    old = ops.FWD_F16
    try:
        ops.FWD_F16 = False
        assert not Fz.forward_is_f16x3(None, train)
        ops.FWD_F16 = True
        ops.PREC_DEFAULT = 1
        assert not Fz.forward_is_f16x3(None, train)          # plain-bf16 products: nothing to split
        ops.PREC_DEFAULT = 0
        assert not Fz.forward_is_f16x3(None, train)          # exact-fp32 test mode
    finally:
        ops.FWD_F16, ops.PREC_DEFAULT = old, 3
    # ops.set_storage is a round trip: the arithmetic and the weight-gradient kernel selection in force before the
    # 2-byte mode come back with set_storage(float32), whatever they were; likewise the exact-fp32 test mode
    from zs3_amd._lib import I, lib
    for prec0, wk0 in ((3, 0), (1, 2), (3, 1)):
        ops.PREC_DEFAULT = prec0
        lib().zs3_conv_wgrad_set_kernel(I(wk0))
        ops.set_storage(torch.bfloat16)
        assert ops.ACT_DTYPE == torch.bfloat16 and ops.PREC_DEFAULT == 1 and not ops.fwd_f16()
        ops.set_storage(torch.bfloat16)                        # entering twice must not forget the fp32 state
        ops.set_storage(torch.float32)
        assert ops.ACT_DTYPE == torch.float32 and ops.PREC_DEFAULT == prec0
        assert lib().zs3_conv_wgrad_set_kernel(I(wk0)) == wk0  # (returns the selection that was in force)
        ops.set_storage(torch.float32)                         # idempotent
        assert ops.PREC_DEFAULT == prec0
    ops.PREC_DEFAULT = 3
    lib().zs3_conv_wgrad_set_kernel(I(2))
    ops.set_exact_fp32(True)
    assert ops.PREC_DEFAULT == 0 and not ops.HALO
    ops.set_exact_fp32(False)
    assert ops.PREC_DEFAULT == 3 and lib().zs3_conv_wgrad_set_kernel(I(0)) == 2 and ops.fwd_f16()
    # nested: the exact-fp32 mode toggled INSIDE the 2-byte mode hands back plain-bf16 products, not PREC_DEFAULT = 3
    # next to bf16 storage (conv_igemm would refuse: "a bf16-stored input needs prec = 1")
    ops.set_storage(torch.bfloat16)
    ops.set_exact_fp32(True)
    ops.set_exact_fp32(False)
    assert ops.PREC_DEFAULT == 1 and ops.ACT_DTYPE == torch.bfloat16 and ops.HALO
    ops.set_exact_fp32(False)                                  # idempotent
    assert ops.PREC_DEFAULT == 1
    ops.set_storage(torch.float32)
    assert ops.PREC_DEFAULT == 3 and ops.fwd_f16()
    log, seen = LossLog(), []
    for i in range(7):
        seen.extend(log.push(torch.tensor(float(i))))
        assert [k for k, _ in seen] == list(range(max(0, i))) or [k for k, _ in seen] == list(range(i + 1))
    seen.extend(log.flush())
    assert seen == [(i, float(i)) for i in range(7)] and log.flush() == []


def test_modules_copy_and_pickle_without_their_process_state():
    """the data-parallel arming of a model (GradSync: bucket tensors, hooks, a process group) and SyncBN's cached
    process-group answer live on the module objects; copy.deepcopy / torch.save of the module must leave them behind."""
    import copy
    import io
    from zs3_amd.modeling.sync_batchnorm.batchnorm import SynchronizedBatchNorm2d

    bn = SynchronizedBatchNorm2d(8)
    bn._sync_cache = ((True, False, True), _Unpicklable())
    bn.sync_group = _Unpicklable()
    twin = copy.deepcopy(bn)
    assert twin._sync_cache is None and twin.sync_group is None and torch.equal(twin.weight, bn.weight)
    torch.save(bn, io.BytesIO())
    from zs3_amd.modeling.deeplab import DeepLab
    m = DeepLab.__new__(DeepLab)
    torch.nn.Module.__init__(m)
    m.lin = torch.nn.Linear(2, 2)
    object.__setattr__(m, "_zs3_grad_sync", _Unpicklable())
    object.__setattr__(m, "_zs3_broadcast_done", True)
    twin = copy.deepcopy(m)
    assert getattr(twin, "_zs3_grad_sync", None) is None and not getattr(twin, "_zs3_broadcast_done", False)
    assert torch.equal(twin.lin.weight, m.lin.weight) and m._zs3_grad_sync is not None
    torch.save(m, io.BytesIO())


class _Unpicklable:
    def __reduce_ex__(self, protocol):
        raise TypeError("process-local state must not be copied")


def test_forward_inside_a_running_backward_keeps_the_backward_bookkeeping():
    """conv_bn_act clears what a DEAD backward pass left behind when a new forward starts -- but a forward that runs
    inside a live backward (activation checkpointing, double backward) must leave the bucket hand-out and the side-stream records
    alone."""
    from zs3_amd import functional as Fz
    Fz._handed.add(12345)
    Fz._handed_armed[0] = True
    seen = {}

    class Probe(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 2

        @staticmethod
        def backward(ctx, g):
            Fz._reset_backward_state()      # what a fused layer's forward would call while this backward is running
            seen["inside"] = (12345 in Fz._handed, Fz._handed_armed[0])
            return g * 2

    x = torch.ones(2, requires_grad=True)
    Probe.apply(x).sum().backward()
    assert seen["inside"] == (True, True)
    Fz._reset_backward_state()              # outside any autograd pass: the leftovers of a dead pass go
    assert 12345 not in Fz._handed and not Fz._handed_armed[0]


def test_plan_entry_points_and_generated_wrappers():
    """Recorded launch plans (include/zs3hip.h, csrc/plan.hip): the misuse codes of the plan API (no launch, so no GPU needed), and
    the generated recording wrappers are the ones zs3_amd/build.py derives from the header as it is now -- one per entry point
    whose last parameter is `void* stream`, nothing else renamed."""
    from zs3_amd import build
    from zs3_amd._lib import lib
    L = lib()
    a, b = L.zs3_plan_create(), L.zs3_plan_create()
    assert a and b and a != b
    assert L.zs3_plan_record_begin(a) == 0 and L.zs3_plan_record_begin(a) == 0       # idempotent for the recording plan
    assert L.zs3_plan_record_begin(b) == -2 and L.zs3_plan_record_end(b) == -2       # one plan records at a time
    assert L.zs3_plan_replay(a, 0, -1) == -2                                         # not into itself
    assert L.zs3_plan_record_end(a) == 0 and L.zs3_plan_size(a) == 0 and L.zs3_plan_replay(a, 0, -1) == 0
    assert L.zs3_plan_find_op(a, b"zs3_sgd_multi_g", 0) == -1 and L.zs3_plan_failed_op(a) == -1
    assert L.zs3_plan_patch(a, 0, 0, (ctypes.c_float * 1)(1.0), 4) == -1 and L.zs3_plan_set_ptr(a, 0, 0, None) == -1
    assert L.zs3_plan_replace_u64(a, 1, 2) == 0 and L.zs3_plan_find_ptr(a, None, None, 0) == 0
    assert L.zs3_plan_destroy(a) == 0 and L.zs3_plan_destroy(b) == 0 and L.zs3_plan_destroy(0) == -1
    assert L.zs3_sgd_max_groups() == 8
    # generated sources are current
    rename_h, wrappers = (os.path.join(build.GEN, f) for f in ("plan_rename.h", "plan_wrappers.hip"))
    before = [open(p).read() for p in (rename_h, wrappers)]
    build.generate_plan_sources()
    assert [open(p).read() for p in (rename_h, wrappers)] == before, "csrc/gen is stale: run python -m zs3_amd.build and commit"
    launchers = [name for _, name, params in build.parse_header() if params and params[-1] == ("void*", "stream")]
    renamed = re.findall(r"#define (zs3_\w+) \1__impl", before[0])
    assert renamed == launchers and len(launchers) >= 60
    for name in launchers:
        assert f'extern "C" int {name}(' in before[1] and f"{name}__impl(" in before[1]
    for name, arrays in build.HOST_ARRAYS.items():       # host arrays are copied into the plan, by name
        assert name in launchers
        for arg in arrays:
            assert re.search(rf"blk_\.{arg}\[i\] = {arg}\[i\]", before[1]), (name, arg)
    # the __impl definitions stay inside the library
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", L._name], capture_output=True, text=True).stdout
    assert "__impl" not in out
