#!/usr/bin/env python3
"""bench.py -- train images/sec of the ZS3 hot path on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]          # N>1: launched by torch.distributed.run

Workload (config.workload): configs[1] of BASELINE.json -- the supervised DeepLabv3+ ResNet-101 training step of
train_pascal.py (forward, CE, backward, SGD; base_trainer.py:16-20) on synthetic 513x513 batches, 16 images per GPU,
21 classes, fp32 semantics computed as three-product 16-bit hi/lo splits on the MFMA cores (forward fp16 halves, backward bf16 halves).  Weak scaling: every rank owns 16 images;
gradients are SUM all-reduced over RCCL while backward runs (zs3_amd.parallel.GradSync).  The GMMN step
(configs[2], train_pascal_GMMN.py:139-268) is timed after the main loop and reported under "gmmn".

One JSON line on stdout (rank 0).  `roofline` prices the dominant kernel (the LDS-DMA 256x128 implicit-GEMM convolution)
with HIP events recorded around its launches inside the timed region (every 5th timed step); `cpu_baseline` times the CPU oracle
(oracle/zs3_oracle, the checked restatement of the reference) on the host cores for a bounded sample.

Further objects of the line, none of them part of `value` (each has a --*-steps flag, 0 = skip): `bf16` (the 2-byte mode of
configs[4], same process, after the timed loop), `gmmn` (configs[2]; `gmmn.script_loop` = the reference's own per-image x
per-class loop body driven through the drop-in classes), and, at N = 1 only and in child processes so that they cannot
disturb the timed loop, `shard` (what ONE rank of the 8-GPU configs[3] run executes: 8 images, 60 classes, no collectives)
and `ddp_one_rank` (the N > 1 code path -- gradient buckets, SyncBN and CE all-reduces over a one-rank RCCL group -- against the
plain step).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL / cross-process device memory on this driver

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FWD_GFLOP_PER_IMG = {21: 185.64, 60: 185.97}          # BASELINE.md section 3 (2*MAC, convs only)
TRAIN_GFLOP_PER_IMG = {21: 555.7, 60: 556.7}          # fwd + dgrad + wgrad, no dgrad for the stem
DTYPE_NOTE = {"bf16x3": "x3 split (fp32 operands as 16-bit hi+lo, 3 MFMA products per pair, fp32 accumulate; fp32 storage): forward "
                        "convolutions f16x3 (fp16 hi/lo, 2^-22-class products), data / weight gradients bf16x3 (bf16 hi/lo, 2^-16-class)",
              "bf16": "bf16 (activations and inter-layer gradients stored as bf16, plain bf16 MFMA products, fp32 accumulate; fp32 master "
                      "weights, weight gradients, BN statistics, class scores)",
              "bf16f32": "bf16 products on fp32 storage (plain bf16 MFMA products, fp32 accumulate; every tensor fp32 in HBM)"}
LAUNCH_BOUNDARY_US = 1.45         # dependent kernel boundary on one stream (MI355X_MICROARCH.md price list)
PEAK_BF16_TF = 2500.0                                 # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU")
    ap.add_argument("--size", type=int, default=513)
    ap.add_argument("--classes", type=int, default=21)
    ap.add_argument("--workload", choices=["supervised", "gmmn", "gcn_context"], default="supervised")
    ap.add_argument("--gmmn-steps", type=int, default=10, help="timed GMMN steps (configs[2]) after the supervised loop; 0 = skip")
    ap.add_argument("--dtype", choices=["bf16x3", "bf16", "bf16f32"], default="bf16x3",
                    help="bf16x3 = fp32 storage, fp32 semantics as three bf16 MFMA products (configs[1-3]); bf16 = the 2-byte mode of "
                         "configs[4]: activations and inter-layer gradients stored as bf16, plain bf16 products, fp32 accumulate / "
                         "statistics / master weights; bf16f32 = plain bf16 products on fp32 storage (the rounds 2-3 form of --dtype bf16)")
    ap.add_argument("--sync-bn", type=int, default=-1,
                    help="SynchronizedBatchNorm2d (cross-rank batch statistics, one fp64 all-reduce per layer and direction): "
                         "-1 = the reference's rule, on iff more than one GPU (train_pascal.py:279); 0 / 1 force it")
    ap.add_argument("--gmmn-pipeline", type=int, default=1,
                    help="1: the next batch's feature pass overlaps the current batch's generator loop (GMMNStep.prefetch)")
    ap.add_argument("--ddp-selftest", action="store_true",
                    help="1-GPU run with a one-rank RCCL group and the full gradient-sync plumbing (cost of the N>1 code path)")
    ap.add_argument("--host-batches", action="store_true",
                    help="supervised workload: every step takes its batch from pinned HOST memory (what the reference's "
                         "`image.cuda(), target.cuda()` does, base_trainer.py:13): the copy of step i+1 is queued on a copy stream "
                         "while step i computes.  Not the headline number (that one has its inputs resident in HBM); DESIGN.md "
                         "section 7 quotes this PCIe-inclusive rate")
    ap.add_argument("--read-loss", type=int, default=1,
                    help="1 (default): every timed step's loss value is read on the host, one step behind (what BaseTrainer does); 0: never")
    ap.add_argument("--bf16-steps", type=int, default=10,
                    help="timed steps of the 2-byte mode reported as the `bf16` object of the default line; 0 = skip")
    ap.add_argument("--shard-steps", type=int, default=10,
                    help="timed steps of `shard`: what ONE rank of BASELINE configs[3] runs (B = 8 images, 60 classes); 0 = skip")
    ap.add_argument("--ddp-steps", type=int, default=10,
                    help="timed steps of `ddp_one_rank`: the supervised step on the N > 1 code path with a one-rank RCCL group and "
                         "SyncBN (gradient buckets, 208 SyncBN all-reduces, global CE), run as a child process; 0 = skip")
    ap.add_argument("--host-steps", type=int, default=10,
                    help="timed steps of `host_batches`: the supervised step taking every batch from pinned host memory (54 MB over "
                         "PCIe per step, what base_trainer.py:13 does), child process; 0 = skip")
    ap.add_argument("--script-steps", type=int, default=3,
                    help="timed iterations of `gmmn.script_loop`: the reference's own per-image x per-class loop body "
                         "(train_pascal_GMMN.py:139-268) on the drop-in modules; 0 = skip")
    ap.add_argument("--plan", type=int, default=1,
                    help="1 (default): the supervised step runs as a recorded launch plan (zs3_amd/plan.py: recorded once after two eager "
                         "steps, then replayed from C -- bit-identical to the eager step, tests/test_gpu_plan.py); 0: every step eager "
                         "(one Python -> ctypes call per launch).  Instrumented steps (roofline events) are always eager.")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible on this node")
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    # the CPU baseline (a child process on the host cores) runs FIRST: the GPU phase that follows is then one contiguous
    # stretch at the end of the run instead of a few seconds hidden in front of a minute of CPU work
    cpu_info = cpu_baseline(args) if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # the step's streams take their hardware queues before the process group creates its own (zs3_amd.functional.warm_streams)
    from zs3_amd import functional as _Fz
    _Fz.warm_streams(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    elif args.ddp_selftest:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")

    from zs3_amd import functional as Fz
    from zs3_amd import ops
    from zs3_amd.modeling.deeplab import DeepLab
    from zs3_amd.modeling.gmmn import GMMNnetwork
    from zs3_amd.optim import SGD, Adam
    from zs3_amd.utils.loss import SegmentationLosses
    from zs3_amd.utils.lr_scheduler import LR_Scheduler
    from zs3_amd.utils.synthetic import make_batch
    from zs3_amd.gmmn_trainer import GMMNStep

    if args.sync_bn < 0:
        args.sync_bn = 1 if world > 1 else 0
    ops.PREC_DEFAULT = 3 if args.dtype == "bf16x3" else 1
    if args.dtype == "bf16":
        ops.set_storage(torch.bfloat16)
    unseen = [10, 14]
    seen = [c for c in range(args.classes) if c not in unseen]
    torch.manual_seed(1)
    model = DeepLab(num_classes=args.classes, pretrained=False, sync_bn=bool(args.sync_bn)).to(dev).train()
    # (supervised: the model takes rank 0's parameters when it arms itself at its first training forward; GMMN workloads: GMMNStep
    # does it at its first step)
    # sync_bn=True models exchange their BatchNorm sums across ranks by construction; nothing to switch on
    groups = [{"params": model.get_1x_lr_params(), "lr": 0.007}, {"params": model.get_10x_lr_params(), "lr": 0.07}]
    opt = SGD(groups, momentum=0.9, weight_decay=5e-4, nesterov=False)
    crit = SegmentationLosses(cuda=True).build_loss("ce")   # normalises over every rank's shard by itself when world > 1
    sched = LR_Scheduler("poly", 0.007, 50, 1000, verbose=False)
    multi = world > 1 or args.ddp_selftest
    if args.ddp_selftest:
        import zs3_amd.parallel as par
        par.FORCE_COLLECTIVES = True      # one-rank group: run every collective of the N>1 path anyway
    # supervised step: the model arms its bucketed gradient all-reduce itself at its first training forward
    # (zs3_amd.parallel.ensure_data_parallel, the path a reference script gets); the GMMN step has its own two small exchanges
    sync, sync_bytes = None, None
    batch = make_batch(args.batch, args.size, args.classes, unseen, seed=1 + rank, device=dev)
    image, label = batch["image"], batch["label"]

    host = None
    if args.host_batches:      # double-buffered pinned host batch -> device, one step ahead, on its own stream
        host = {"image": image.cpu().pin_memory(), "label": label.cpu().pin_memory(), "stream": torch.cuda.Stream(),
                "bufs": [(torch.empty_like(image), torch.empty_like(label)) for _ in range(2)], "ready": [None, None]}

        def stage(slot):
            with torch.cuda.stream(host["stream"]):
                host["bufs"][slot][0].copy_(host["image"], non_blocking=True)
                host["bufs"][slot][1].copy_(host["label"], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            host["ready"][slot] = ev
        stage(0)

    from zs3_amd.plan import StepPlan
    plan_step = StepPlan(model, crit, opt, enabled=bool(args.plan))

    def supervised_step(i, eager=False):
        img, lab = image, label
        if host is not None:
            slot = i & 1
            torch.cuda.current_stream().wait_event(host["ready"][slot])
            img, lab = host["bufs"][slot]
            host["stream"].wait_stream(torch.cuda.current_stream())   # the other buffer's last reader (step i-1) is queued
            stage(slot ^ 1)
        sched(opt, i, 0, 0.0)
        # zero_grad -> forward -> loss -> backward -> optimizer step (base_trainer.py:16-20): eager for the first two calls of a
        # configuration, recorded on the third, replayed from C afterwards (N > 1: always eager, the collectives are torch.distributed's)
        out, loss = plan_step(img, lab, eager=eager)
        # the trainer reads the loss of every iteration (zs3_amd/base_trainer.py; the reference: base_trainer.py:21,24).  Read the way
        # the trainer reads it: copied to pinned host memory behind the step, looked at one iteration later -- every value is read,
        # no step waits for its own
        if loss_log is not None:
            loss_log.push(loss)
        return loss

    from zs3_amd.base_trainer import LossLog
    loss_log = LossLog(dev) if args.read_loss else None

    def run(step, steps, warmup):
        for i in range(warmup):
            step(i)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            last = step(warmup + i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, last

    gmmn_info = None

    def build_gmmn():
        gen = GMMNnetwork(300, 300, 256, 256).to(dev).train()
        opt_g = Adam(gen.parameters(), lr=2e-4)
        w = torch.ones(args.classes, device=dev)
        w[unseen] = 100.0
        dp = multi and args.workload != "supervised"
        crit_g = SegmentationLosses(weight=w, cuda=True).build_loss("ce")
        gb = make_batch(args.batch, args.size, args.classes, unseen, seed=101 + rank, with_label_emb=True, device=dev)
        if args.workload == "gcn_context":    # train_context_GMMN_GCNcontext.py step (BASELINE configs[4] flow)
            from zs3_amd.gcn_trainer import GCNContextStep
            from zs3_amd.modeling.gmmn import GMMNnetwork_GCN
            gcn = GMMNnetwork_GCN(300, 300, 256, 256).to(dev).train()
            stepper = GCNContextStep(model, gen, gcn, opt, opt_g, Adam(gcn.parameters(), lr=2e-4), crit_g, seen=seen,
                                     unseen=unseen, noise="device")
        else:
            stepper = GMMNStep(model, gen, opt, opt_g, crit_g, seen=seen, unseen=unseen, noise="device",
                               )
        # next_image: the trainers look one batch ahead and start its frozen-backbone feature pass next to this batch's
        # generator loop (GMMNStep.prefetch); the synthetic loader returns the same batch every time
        fn = lambda i: stepper(gb["image"], gb["label"], gb["label_emb"], next_image=gb["image"] if args.gmmn_pipeline else None)
        fn.stepper, fn.image, fn.batch = stepper, gb["image"], gb
        return fn

    def gmmn_report(gstep, steps):
        from zs3_amd import gmmn_trainer as _gt
        NL = 6 if _gt.PREP_IN_FWD1 else 7      # launches of one generator update (zs3_gmmn_mlp_fwd1_table merges the first two)
        """configs[2]: the GMMN step as a first-class line -- images/s over `steps` timed steps plus where the time goes:
        the frozen-backbone feature pass (timed alone with HIP events) and the per-(image, class) generator updates."""
        gdt, _ = run(gstep, steps, 2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.no_grad():
            model.forward_before_class_prediction(gstep.image)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(3):
                model.forward_before_class_prediction(gstep.image)
            e1.record()
        torch.cuda.synchronize()
        fwd_ms = e0.elapsed_time(e1) / 3
        step_ms = 1e3 * gdt / steps
        upd = int(getattr(gstep.stepper, "last_updates", 0))
        script = None
        if args.script_steps > 0 and args.workload != "gcn_context":
            # the reference's own loop body on the drop-in modules (same model, batch, class split; its own generator / optimizers)
            from zs3_amd.utils.loss import GMMNLoss
            torch.manual_seed(7)
            gen_s = GMMNnetwork(300, 300, 256, 256).to(dev).train()
            opt_s = Adam(gen_s.parameters(), lr=2e-4)
            w_s = torch.ones(args.classes, device=dev)
            w_s[unseen] = 100.0
            crit_s, crit_gs = SegmentationLosses(weight=w_s, cuda=True).build_loss("ce"), GMMNLoss(cuda=True).build_loss()
            fn = lambda i: script_style_gmmn_iteration(model, gen_s, opt, opt_s, crit_s, crit_gs, gstep.batch, set(float(c) for c in seen),
                                                       set(float(c) for c in unseen))
            sdt, slast = run(fn, args.script_steps, 1)
            script = {"ms_per_step": 1e3 * sdt / args.script_steps, "value": args.batch * args.script_steps / sdt, "unit": "images/sec",
                      "steps": args.script_steps, "generator_updates_per_step": slast[2], "last_classifier_loss": slast[1],
                      "note": "train_pascal_GMMN.py:139-268 as written -- eager GMMNnetwork on every pixel of a class, boolean masks, "
                              "autograd through GMMNLoss, CPU noise / sample indices, .item() per (image, class) -- on the drop-in modules; "
                              "`GMMNStep` (ms_per_step of this object) is the same iteration as fixed-shape captured updates"}
        return {"value": args.batch * steps / gdt, "unit": "images/sec", "ms_per_step": step_ms, "steps": steps, "script_loop": script,
                "workload": "train_pascal_GMMN.py step (BASELINE configs[2]): frozen DeepLabv3+ feature pass + per-(image, class) "
                            "GMMN/MMD/Adam updates + pred_conv CE/SGD, device noise",
                "breakdown": {"backbone_forward_ms": fwd_ms, "generator_updates_per_step": upd,
                              "generator_loop_and_classifier_ms": step_ms - fwd_ms,
                              "us_per_generator_update": (1e3 * (step_ms - fwd_ms) / upd) if upd else None},
                "model_tflops": args.batch * steps / gdt * 193.5 / 1e3,
                "model_frac_of_bf16_peak": args.batch * steps / gdt * 193.5 / 1e3 / PEAK_BF16_TF,
                # the step has two regimes with two different bounds: the frozen-backbone feature pass is the conv stack
                # (MFMA roofline, 185.64 GF per image forward), the generator loop is launch-latency-bound
                "roofline": {
                    "backbone_forward": {"bound": "mfma", "achieved": args.batch * FWD_GFLOP_PER_IMG.get(args.classes, 185.64) / fwd_ms, "peak": PEAK_BF16_TF,
                                         "unit": "TFLOP/s", "frac": args.batch * FWD_GFLOP_PER_IMG.get(args.classes, 185.64) / fwd_ms / PEAK_BF16_TF},
                    "generator_update": {"bound": "launch latency", "launches_per_update": NL,
                                         "us_per_update": (1e3 * (step_ms - fwd_ms) / upd) if upd else None,
                                         "launch_floor_us": NL * LAUNCH_BOUNDARY_US,
                                         "frac_of_floor": (NL * LAUNCH_BOUNDARY_US / (1e3 * (step_ms - fwd_ms) / upd)) if upd else None,
                                         "note": f"floor = {NL} dependent kernel boundaries x 1.45 us (MI355X_MICROARCH.md, "
                                                 "'boundary' row: eager = hipGraph); the per-update figure also carries the "
                                                 "classifier's CE / SGD launches of the step"}}}

    def measure_supervised(steps, warmup, roofline):
        """warm-up + timed supervised steps -> (seconds, last loss, event records of the timed steps, of the last warm-up step,
        instrumented step count)"""
        if plan_step.enabled and plan_step._plannable(image, label):
            # settle + record the launch plan of this configuration OUTSIDE the measured steps (two eager steps, one recorded: the
            # recording allocates its private pool with fresh hipMallocs).  These are extra untimed steps in front of the W warm-up steps.
            plan_step.close()
            rec0, rep0 = plan_step.recordings, plan_step.replays
            for i in range(5):
                if plan_step.recordings > rec0 and plan_step.replays > rep0:
                    break
                supervised_step(i)
        if not roofline:
            dt, last = run(supervised_step, steps, warmup)
            return dt, last, [], [], 0
        # warm-up steps: HIP events around every conv launch (per-kernel table, picks the dominant kernel);
        # timed steps: events around the dominant kernel's launches only -- an event pair is a kernel boundary the
        # GPU cannot overlap, and 450 of them per step cost ~3 % of the step
        ops.PROFILE = []
        for i in range(warmup):
            if i == warmup - 1:
                ops.PROFILE = []
            supervised_step(i, eager=True)        # (timing events around every conv launch live in the Python wrappers)
        warm_prof = ops.PROFILE
        torch.cuda.synchronize()
        by_cfg = {}
        for tag, flops, e0, e1, cfg in warm_prof:
            by_cfg[cfg] = by_cfg.get(cfg, 0.0) + e0.elapsed_time(e1)
        for group in ({41, 42},):      # both tile heights of the strip-resident kernel are one kernel
            tot = sum(by_cfg.pop(c, 0.0) for c in group)
            if tot:
                by_cfg[min(group)] = tot
        best = max(by_cfg, key=by_cfg.get) if by_cfg else None
        ops.PROFILE_CFGS = ({41, 42} if best == 41 else {best}) if best is not None else None
        timed_prof = []

        def instrument(i):
            return i % 5 == 4 or (steps < 5 and i == steps - 1)

        STRIDE = 4     # ... and in such a step every 4th launch of the dominant kernel, the phase rotating from one
        nth = [0]      # instrumented step to the next: 127 launches per step, so four instrumented steps cover each
                       # launch exactly once.  (An event pair is ~50-100 us of host time and a kernel boundary the
                       # GPU cannot overlap: 127 pairs in a step made that step ~25 ms longer.)

        # With a recorded plan the instrumented steps are replays too: the event pairs are recorded by zs3_plan_replay itself around
        # the sampled launches (zs3_plan_time_ops), on the stream each launch runs on.  The k-th convolution launch of the plan is the
        # k-th record of the eager warm-up step above (same launch order), which supplies its kernel tag and algorithmic flops.
        conv_ops = plan_step.conv_ops()
        in_replay = bool(conv_ops) and len(conv_ops) == len(warm_prof) and ops.PROFILE_CFGS is not None
        dominant = [k for k, rec in enumerate(warm_prof) if ops.PROFILE_CFGS is not None and rec[4] in ops.PROFILE_CFGS]
        timed_sets = []

        def sampled_step(i):
            if instrument(i) and in_replay:
                pick = dominant[nth[0] % STRIDE::STRIDE]
                nth[0] += 1
                ops.PROFILE, ops.PROFILE_SAMPLE = None, None
                timed_sets.append((plan_step.time_next_replay([conv_ops[k] for k in pick]), pick))
                return supervised_step(i)
            if instrument(i):
                ops.PROFILE = timed_prof
                ops.PROFILE_SAMPLE = [STRIDE, nth[0] % STRIDE, -1]
                nth[0] += 1
            else:
                ops.PROFILE, ops.PROFILE_SAMPLE = None, None
            return supervised_step(i, eager=instrument(i))
        dt, last = run(sampled_step, steps, 0)
        ops.PROFILE, ops.PROFILE_CFGS, ops.PROFILE_SAMPLE = None, None, None
        for sid, pick in timed_sets:
            for k, ms in zip(pick, plan_step.timed_ms(sid, len(pick))):
                timed_prof.append((warm_prof[k][0], warm_prof[k][1], _Duration(ms), None, warm_prof[k][4]))
        measure_supervised.replayed_instrumented = bool(timed_sets)
        return dt, last, timed_prof, warm_prof, sum(1 for i in range(steps) if instrument(i))

    bf16_info = None
    if args.workload == "supervised":
        dt, last, prof, warm_prof, instrumented = measure_supervised(args.steps, args.warmup, not args.no_roofline)
        # resolve the instrumentation's timing events NOW and let them go: ~500 unread HIP timing events held across the phases below
        # made whichever phase came third 25-35 % slower (shard 35.8 instead of 28 ms, or bf16 39.7 instead of 29.4: the events' signals
        # are a finite pool that the side-stream waits of the weight-gradient launches share)
        roof_main = roofline_of(prof, warm_prof, instrumented, args.dtype) if (rank == 0 and prof) else None
        prof, warm_prof = [], []
        inst_replayed = bool(getattr(measure_supervised, "replayed_instrumented", False))
        n_eager = 0 if inst_replayed else instrumented
        executor = {"plan": bool(plan_step.recorded_ops), "recorded_launches": plan_step.recorded_ops,
                    "timed_steps_replayed": args.steps - n_eager if plan_step.recorded_ops else 0,
                    "timed_steps_eager": n_eager if plan_step.recorded_ops else args.steps,
                    "note": "zs3_amd/plan.py: the step's entry-point calls recorded once (include/zs3hip.h, zs3_plan_*) and replayed from C, "
                            "bit-identical to the eager step; the `roofline` events of the instrumented steps are recorded by the replay "
                            "itself (zs3_plan_time_ops) around the sampled launches"}
        if args.bf16_steps > 0 and world == 1 and args.dtype == "bf16x3":
            # the 2-byte mode (BASELINE configs[4] "bf16") on the same model, optimizer and batch: activations and
            # inter-layer gradients stored as bf16, plain bf16 products
            ops.set_storage(torch.bfloat16)
            bdt, _, bprof, bwarm, binst = measure_supervised(args.bf16_steps, 3, not args.no_roofline)
            ops.set_storage(torch.float32)      # (a round trip: arithmetic and weight-gradient kernel selection come back)
            bval = args.batch * args.bf16_steps / bdt
            bf16_info = {"value": bval, "unit": "images/sec", "ms_per_step": 1e3 * bdt / args.bf16_steps, "steps": args.bf16_steps,
                         "last_loss": float(_.detach().float().item()),
                         "warmup": 3, "dtype": DTYPE_NOTE["bf16"],
                         "workload": "the supervised step of `value` (same model, optimizer, batch) in the 2-byte mode",
                         "model_tflops": bval * TRAIN_GFLOP_PER_IMG.get(args.classes, 555.7) / 1e3,
                         "model_frac_of_bf16_peak": bval * TRAIN_GFLOP_PER_IMG.get(args.classes, 555.7) / 1e3 / PEAK_BF16_TF}
            if bprof:
                bf16_info["roofline"] = roofline_of(bprof, bwarm, binst, "bf16")
            bprof = bwarm = None
        gflop_img = TRAIN_GFLOP_PER_IMG.get(args.classes, 555.7)
        sync = getattr(model, "_zs3_grad_sync", None)      # armed by the model's first training forward when world > 1
        # (per step: the buckets' own size -- replayed steps do not pass through the Python counter)
        sync_bytes = sum(f.numel() * f.element_size() for f in sync.flat) * max(1, args.steps + args.warmup) if sync is not None else None
        if args.gmmn_steps > 0 and world == 1:
            from zs3_amd.parallel import disarm_data_parallel
            disarm_data_parallel(model)   # (--ddp-selftest: the GMMN step exchanges pred_conv's gradients itself)
            gmmn_info = gmmn_report(build_gmmn(), args.gmmn_steps)
    else:
        gstep = build_gmmn()
        prof, warm_prof, instrumented, roof_main = [], [], 0, None
        dt, last = run(gstep, args.steps, args.warmup)
        gflop_img = 193.5

    value = world * args.batch * args.steps / dt
    result = {
        "metric": "train images/sec, DeepLabv3+ + GMMN 513x513",
        "value_is": ("the supervised DeepLabv3+ step (BASELINE configs[1], the configuration the metric is quoted on); the +GMMN "
                     "step (configs[2]) is reported in full under `gmmn`" if args.workload == "supervised" else args.workload + " step"),
        "value": value, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        # the last timed step's loss: a dead network (all-zero activations) runs the same kernels 10-45 % faster on this chip (lower
        # switching power, higher clocks: found in round 4), so the line carries the evidence that it was alive
        "last_loss": float(last.detach().float().item()) if torch.is_tensor(last) else (float(last[1]) if isinstance(last, tuple) else None),
        "dtype": DTYPE_NOTE[args.dtype], "data": "synthetic",
        "config": {"workload": ("train_pascal.py supervised step: DeepLabv3+ ResNet-101 fwd+CE+bwd+SGD (BASELINE configs[1]); the +GMMN step (configs[2]) is the `gmmn` object of this line"
                                if args.workload == "supervised" else
                                "train_pascal_GMMN.py step (BASELINE configs[2])" if args.workload == "gmmn" else
                                "train_context_GMMN_GCNcontext.py step (GCN-context flow of BASELINE configs[4], fp32 arithmetic as above)"),
                   "image": f"{args.size}x{args.size}", "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                   "classes": args.classes, "parallelism": f"dp{world}", "sync_bn": bool(args.sync_bn),
                   "inputs": ("pinned host memory -> HBM every step (PCIe-inclusive)" if args.host_batches and
                              args.workload == "supervised" else "resident in HBM")},
        "model_tflops": value * gflop_img / 1e3,
        "model_frac_of_bf16_peak": value * gflop_img / 1e3 / (PEAK_BF16_TF * world),
    }
    if sync is not None:
        result["rccl_bytes_per_rank_per_step"] = sync_bytes // max(1, args.steps + args.warmup)
    if gmmn_info:
        result["gmmn"] = gmmn_info
    if bf16_info:
        result["bf16"] = bf16_info
    if args.workload == "supervised":
        result["executor"] = executor
    if roof_main is not None:
        # the step's own two floors next to ms_per_step: the x3-MFMA time of the step's conv flops at 2.5 PF / 3, and
        # the step's HBM bytes (PMC passes of this command, profiles/) at the ~6.3 TB/s a streaming kernel sustains on this chip
        tfl = args.batch * gflop_img / 1e3
        mfma_ms = tfl / (PEAK_BF16_TF / (3.0 if args.dtype == "bf16x3" else 1.0)) * 1e3
        step_gb = pmc_step_bytes("bf16" if args.dtype == "bf16" else "")
        hbm_ms = step_gb / 6.3 if step_gb else None
        roof_main["step_floor_ms"] = max(mfma_ms, hbm_ms or 0.0)
        roof_main["step_floors"] = {"mfma_ms": mfma_ms, "hbm_ms": hbm_ms, "step_tflop": tfl, "step_hbm_gb": step_gb,
                                    "note": "mfma_ms = conv flops of one step (fwd + dgrad + wgrad) / (2.5 PF / products per operand "
                                            "pair); hbm_ms = HBM bytes of one step from the committed PMC passes / 6.3 TB/s"}
        result["roofline"] = roof_main
    if cpu_info is not None:
        result["cpu_baseline"] = cpu_info
    children = rank == 0 and world == 1 and not args.ddp_selftest and args.workload == "supervised" and args.dtype == "bf16x3" and \
        (args.batch, args.classes) == (16, 21)
    if children and (args.shard_steps > 0 or args.ddp_steps > 0):
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    if children and args.shard_steps > 0:
        result["shard"] = shard_of_configs3(args)
    if children and args.ddp_steps > 0:
        result["ddp_one_rank"] = ddp_one_rank(args, result["ms_per_step"])
    if children and args.host_steps > 0 and not args.host_batches:
        result["host_batches"] = host_batches_child(args, result["ms_per_step"])
    if rank == 0:
        print(json.dumps(result))
    if world > 1 or args.ddp_selftest:
        from zs3_amd import parallel as _par
        torch.cuda.synchronize()
        _par.native_shutdown()
        dist.destroy_process_group()



def script_style_gmmn_iteration(model, generator, optimizer, optimizer_generator, criterion, criterion_generator, batch, seen, unseen,
                                noise_dim=300, embed_dim=300, feature_dim=256, batch_size_generator=128):
    """One iteration written the way the reference's script writes it (train_pascal_GMMN.py:139-268; the loop
    tests/test_gpu_dropin.py drives against the oracle) and using nothing but the drop-in surface: the model's split forwards, eager
    `GMMNnetwork` calls on EVERY pixel of a class, boolean-mask indexing, autograd through `GMMNLoss`, the optimizers' own `step`,
    CPU-drawn noise and sample indices, `.item()` per (image, class).  north_star says the build "drops into
    train_pascal_GMMN.py unchanged": this is what such an unchanged script costs per iteration, next to `GMMNStep`."""
    import torch.nn.functional as F
    image, target, embedding = batch["image"], batch["label"], batch["label_emb"]
    with torch.no_grad():
        real = model.forward_before_class_prediction(image)
    fake = torch.zeros(real.shape, device=real.device)
    fh, fw = real.shape[2:]
    g_total, updates = 0.0, 0
    for n in range(image.shape[0]):
        feats = real[n].permute(1, 2, 0).reshape(-1, feature_dim)
        labels = F.interpolate(target[n].view(1, 1, *target.shape[1:]), size=(fh, fw), mode="nearest").view(-1)
        emb = F.interpolate(embedding[n].unsqueeze(0), size=(fh, fw), mode="nearest")[0].permute(1, 2, 0).reshape(-1, embed_dim)
        generated = torch.zeros_like(feats)
        present = labels.unique()
        has_unseen = bool(sum(float(c) in unseen for c in present))
        running = 0.0
        for c in present:
            if float(c) == 255:
                continue
            optimizer_generator.zero_grad()
            where = labels == c
            count = int(where.sum().item())
            noise = torch.rand((count, noise_dim)).cuda()
            made = generator(emb[where], noise.float())
            if float(c) in seen and not has_unseen:
                pick = torch.randint(low=0, high=count, size=(batch_size_generator,)).cuda()
                g_loss = criterion_generator(made[pick], feats[where][pick])
                running += g_loss.item()
                g_loss.backward()
                optimizer_generator.step()
                updates += 1
            generated[where] = made.detach()
        g_total += running / len(present)
        fake[n] = (feats if not has_unseen else generated).reshape(fh, fw, feature_dim).permute(2, 0, 1)
    optimizer.zero_grad()
    out = model.forward_class_prediction(fake.detach(), image.shape[2:])
    loss = criterion(out, target)
    loss.backward()
    optimizer.step()
    return g_total, loss.item(), updates


def _child_line(extra, timeout=180):
    """one bench.py child process on the same GPU (a clean process: its own allocator pool, caches and streams) -> its JSON line"""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--no-cpu-baseline", "--gmmn-steps", "0", "--bf16-steps", "0", "--shard-steps", "0",
           "--ddp-steps", "0", "--host-steps", "0", "--script-steps", "0", "--no-roofline"] + [str(a) for a in extra]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def shard_of_configs3(args):
    """What ONE rank of BASELINE configs[3] runs (train_context_GMMN.py's model and batch at global B = 64 on 8 GPUs): the supervised
    step on 8 images with the 60-class head (59 + background, datasets/context.py:22) -- every tile rule sees M = 8 712 / 33 800 /
    133 128 rows instead of B = 16's.  A child process (a second model and batch inside this one measured 25-35 % slow in whichever
    phase came third; a clean process does not), no collectives (one rank)."""
    try:
        d = _child_line(["--batch", 8, "--classes", 60, "--size", args.size, "--steps", args.shard_steps, "--warmup", 5])
    except Exception as e:
        return {"ms_per_step": None, "error": type(e).__name__}
    return {"value": d["value"], "unit": "images/sec", "ms_per_step": d["ms_per_step"], "steps": d["steps"], "warmup": d["warmup"],
            "last_loss": d.get("last_loss"), "batch_per_gpu": 8, "classes": 60, "model_tflops": d.get("model_tflops"),
            "workload": "the supervised step on one rank's shard of BASELINE configs[3] (global B = 64 on 8 GPUs -> 8 images per rank, "
                        "60 classes), same arithmetic as `value`; child process, same GPU, no collectives (one rank)"}


def host_batches_child(args, plain_ms):
    """The supervised step with its batch coming from pinned HOST memory every iteration (base_trainer.py:13-14: `image.cuda(),
    target.cuda()`): a child `bench.py --host-batches` -- the copy of batch i + 1 (54.7 MB) is queued on a copy stream while step i
    computes.  Never `value` (that one has its inputs resident in HBM); the PCIe-inclusive rate, next to it."""
    try:
        d = _child_line(["--host-batches", "--steps", args.host_steps, "--warmup", 3, "--batch", args.batch, "--size", args.size,
                         "--classes", args.classes])
    except Exception as e:
        return {"ms_per_step": None, "error": f"{type(e).__name__}"}
    nbytes = args.batch * (3 + 1) * args.size * args.size * 4
    return {"ms_per_step": d["ms_per_step"], "value": d["value"], "unit": "images/sec", "steps": d["steps"], "last_loss": d.get("last_loss"),
            "resident_ms_per_step": plain_ms, "overhead_ms": d["ms_per_step"] - plain_ms, "host_to_device_bytes_per_step": nbytes,
            "workload": "the supervised step of `value` with every batch copied from pinned host memory (image fp32 + label fp32), double "
                        "buffered on a copy stream; child process, same GPU"}


def ddp_one_rank(args, plain_ms):
    """The supervised step on the N > 1 CODE PATH, observed by the driver: a child `bench.py --ddp-selftest --sync-bn 1` with a
    one-rank RCCL process group -- GradSync's bucketed all-reduce (237 MB per step, launched from the weight-gradient stream),
    one fp64 all-reduce per SynchronizedBatchNorm2d layer and direction (208 per step) and the CE's global weight sum all run,
    with nobody to talk to.  What the number is: the launch / stream / collective-call overhead of the path every rank of an
    8-GPU run executes; what it is not: link time.  (No >= 2-rank run exists on this pool's one-GPU boxes.)"""
    try:
        d = _child_line(["--ddp-selftest", "--sync-bn", 1, "--steps", args.ddp_steps, "--warmup", 3, "--batch", args.batch, "--size", args.size,
                         "--classes", args.classes])
    except Exception as e:
        return {"ms_per_step": None, "error": f"{type(e).__name__}"}
    return {"ms_per_step": d["ms_per_step"], "value": d["value"], "unit": "images/sec", "steps": d["steps"], "last_loss": d.get("last_loss"),
            "plain_ms_per_step": plain_ms, "overhead_ms": d["ms_per_step"] - plain_ms,
            "rccl_bytes_per_rank_per_step": d.get("rccl_bytes_per_rank_per_step"), "sync_bn": True, "ranks": 1,
            "workload": "the supervised step of `value` through the N > 1 path: one-rank RCCL group, GradSync buckets (in-place all-reduce "
                        "from the weight-gradient stream), 208 SyncBN all-reduces, globally normalised CE; child process, same GPU"}


class _Duration:
    """a measured duration in the place of an event pair (the records of launches timed inside a plan replay)"""

    def __init__(self, ms):
        self.ms = ms

    def elapsed_time(self, _end):
        return self.ms


def roofline_of(prof, warm_prof, instrumented, dtype):
    """`roofline` object of one measured configuration: the dominant conv kernel's algorithmic flops / HIP-event time inside the
    timed steps against the dense bf16 MFMA peak, with the HBM bytes per launch from the committed rocprofv3 PMC passes"""
    torch.cuda.synchronize()

    def aggregate(records):
        agg = {}
        for tag, flops, e0, e1, _cfg in records:
            a = agg.setdefault(tag, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += flops
            a[2] += e0.elapsed_time(e1) * 1e-3
        return agg
    agg, warm = aggregate(prof), aggregate(warm_prof)
    # the strip-resident kernel is ONE source kernel (csrc/conv_halo.hip) in several template instantiations (tile height,
    # strip passes per step): the roofline prices the family, `instantiations` lists each under its rocprof name
    fam = {k: v for k, v in agg.items() if k.startswith("conv_halo_kernel<")}
    if fam:
        agg = {k: v for k, v in agg.items() if k not in fam}
        agg["conv_halo<%d>" % (3 if dtype == "bf16x3" else 1)] = [sum(v[i] for v in fam.values()) for i in range(3)]
    tag = max(agg, key=lambda k: agg[k][2])
    n, fl, sec = agg[tag]
    ach = fl / sec / 1e12
    traffic, traffic_src = pmc_traffic(tag, "bf16" if dtype == "bf16" else "")
    return {
        "bound": "mfma", "achieved": ach, "peak": PEAK_BF16_TF, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TF,
        "traffic": traffic, "traffic_source": traffic_src, "kernel": tag, "launches": n, "avg_launch_us": 1e6 * sec / n,
        "instrumented_timed_steps": instrumented,
        "instantiations": ({k: {"launches": v[0], "avg_launch_us": 1e6 * v[2] / v[0], "tflops": v[1] / v[2] / 1e12}
                            for k, v in sorted(fam.items())} if fam and tag.startswith("conv_halo<") else None),
        "note": "achieved = algorithmic 2*M*N*K flops of the sampled launches / their HIP-event time (events around every "
                "4th launch of this kernel, rotating phase, in every 5th timed step)" + ("; each product costs 3 16-bit MFMAs (f16x3 forward / bf16x3 backward), so MFMA-issue utilisation "
                "is 3x this fraction" if dtype == "bf16x3" else "; one bf16 MFMA per product"),
        "all_conv_igemm_last_warmup_step": {k: {"launches": v[0], "tflops": v[1] / v[2] / 1e12, "ms": 1e3 * v[2]}
                                            for k, v in warm.items()},
    }

def pmc_step_bytes(variant=""):
    """HBM GB moved by one training step (read + write) from the newest committed PMC passes of this command"""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic%s.json" % ("_" + variant if variant else ""))))
    if not found:
        return None
    d = json.load(open(found[-1]))
    if "per_step_read_GB" not in d:
        return None
    return float(d["per_step_read_GB"]) + float(d["per_step_write_GB"])


def pmc_traffic(tag, variant=""):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/r1_pmc_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs, FETCH doubled per the
    gfx950 note of MI355X_MICROARCH.md).  PMC collection cannot run inside the timed process, hence the file."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic%s.json" % ("_" + variant if variant else ""))))
    if not found:
        return None, None
    path = found[-1]            # the latest round's passes
    import re
    kernels = json.load(open(path))["kernels"]
    src = f"profiles/{os.path.basename(path)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
    hm = re.match(r"conv_halo<(\d)>", tag)
    if hm:   # the strip-resident kernel is one source kernel in several instantiations (tile height, strip passes per step)
        precs = ("3", "4") if hm.group(1) == "3" else (hm.group(1),)     # <3>: the x3 family = bf16x3 (data gradient) + f16x3 (forward) instantiations
        fam = [v for k, v in kernels.items() if any(k.startswith(f"conv_halo_kernel<{q},") for q in precs)]
        n = sum(v["launches"] for v in fam)
        if not n:
            return None, None
        return sum(v["launches"] * (v["read_bytes_per_launch"] + v["write_bytes_per_launch"]) for v in fam) / n, src
    m = re.match(r"conv_igemm(_ws|_dma)?<(\d+),(\d+),(\d)(?:,pipe(\d))?>", tag)
    if not m:
        return None, None
    name = (f"conv_igemm{m.group(1)}_kernel<{m.group(4)}, false>" if m.group(1) == "_dma" else f"conv_igemm{m.group(1)}_kernel<{m.group(4)}>") if m.group(1) else f"conv_igemm_kernel<{m.group(2)}, {m.group(3)}, {m.group(4)}, {m.group(5)}>"
    k = kernels.get(name) or kernels.get(name.replace(", false>", ">"))
    if not k:
        return None, None
    return k["read_bytes_per_launch"] + k["write_bytes_per_launch"], src


def cpu_baseline(args):
    """The checked CPU restatement of the reference (oracle/, kind "port") on this node's host cores, in a child process
    with a time limit: the oracle's supervised step in torch CPU fp32 at B=2, 513x513, one warm-up + three timed steps,
    median.  One leg with min(usable cores, 64) threads: rounds 1-2 also tried one thread per core on the 256-core hosts of
    this pool and it never finished four steps in 45 s (the thread pool oversubscribes B=2 convolutions), so that leg is
    gone.  `cores` = the threads actually used (the bench contract), `host_cores` = what the host offers."""
    import subprocess
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads, limit = min(usable, 64), 60
    code = (
        "import sys, time, json, torch\n"
        f"sys.path.insert(0, {os.path.join(ROOT, 'oracle')!r})\n"
        "import zs3_oracle as zo\n"
        "threads, classes, size, bsz = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 2\n"
        "torch.set_num_threads(threads)\n"
        "torch.manual_seed(1)\n"
        "m = zo.DeepLab(num_classes=classes, pretrained=False).train()\n"
        "groups = [{'params': m.get_1x_lr_params(), 'lr': 0.007}, {'params': m.get_10x_lr_params(), 'lr': 0.07}]\n"
        "opt = torch.optim.SGD(groups, momentum=0.9, weight_decay=5e-4)\n"
        "crit = zo.SegmentationLosses().build_loss('ce')\n"
        "b = zo.make_synthetic_batch(bsz, size, classes, seed=1, with_label_emb=False)\n"
        "zo.supervised_step(m, opt, crit, b['image'], b['label'])\n"
        "times = []\n"
        "for _ in range(3):\n"
        "    t0 = time.perf_counter()\n"
        "    zo.supervised_step(m, opt, crit, b['image'], b['label'])\n"
        "    times.append(time.perf_counter() - t0)\n"
        "print(json.dumps({'median_s': sorted(times)[1], 'bsz': bsz}))\n")
    t0 = time.perf_counter()
    try:
        r = subprocess.run([sys.executable, "-c", code, str(threads), str(args.classes), str(args.size)],
                           capture_output=True, text=True, timeout=limit)
        d = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:  # timeout or a failed child
        return {"value": None, "unit": "images/sec", "cores": threads, "threads": threads, "host_cores": usable, "kind": "port",
                "sample": f"oracle supervised step did not finish 4 steps in {limit} s with {threads} threads ({type(e).__name__})"}
    return {"value": d["bsz"] / d["median_s"], "unit": "images/sec", "cores": threads, "threads": threads, "host_cores": usable,
            "kind": "port", "wall_s": time.perf_counter() - t0,
            "sample": f"oracle supervised step (fwd+CE+bwd+SGD), B={d['bsz']} at {args.size}x{args.size}, 1 warm-up + 3 timed "
                      f"steps, median; torch CPU fp32 with {threads} threads on a host with {usable} usable cores "
                      f"({os.cpu_count()} logical); run before the GPU phase of this same bench.py process.  Why not one thread per "
                      "core (BASELINE.md section 4): on the 256-core hosts of this pool torch's thread pool oversubscribes B = 2 "
                      "convolutions -- 256 threads did not finish four steps in 45 s in rounds 1-2, 64 is the fastest setting found"}


if __name__ == "__main__":
    main()
