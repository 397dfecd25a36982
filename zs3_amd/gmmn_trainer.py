"""GMMN training step of ZS3 on MI355X -- the loop body of zs3/train_pascal_GMMN.py:139-268
(train_context_GMMN.py is identical).

Semantics kept from the reference: frozen-backbone feature pass in train() mode under no_grad; per image,
per class (ascending label order) a generator call; an MMD + Adam step for seen classes of images without
unseen pixels, on `batch_size_generator` rows sampled with replacement (the same indices for fake and real
rows); generated features written back for images that contain an unseen class, real features kept otherwise
(`real_seen_features`); one CE/SGD step of `pred_conv` on the stitched feature batch; generator_loss_batch
divides by len(unique labels) including 255.

What is MI355X-native here:
* class masks come from one stable device sort per batch instead of boolean indexing (no per-class host sync);
* for images whose generated features are discarded anyway (no unseen pixel => real features are used) the
  generator only runs on the *sampled* rows -- rows of an MLP are independent and the noise / dropout of a sampled
  row are keyed on the class pixel it was drawn from (duplicate samples share them, like z[random_idx] of the
  reference), so the Adam update is the one the reference computes -- which makes the whole per-(image, class)
  update fixed-shape;
* that fixed-shape update (gather + cat, two MFMA row-GEMMs, MMD fwd/bwd, MLP backward, fused Adam, weight
  re-split) is captured ONCE into a hipGraph and replayed per (image, class): ~30 launches become one replay;
  dropout / noise seeds and the Adam step count live in device memory so that replays differ;
* the scalar losses are read back once per step instead of `.item()` per (image, class).

Multi-GPU (`group=`): every rank runs the frozen-backbone forward and the generator updates of its own
images; once per step the generator parameters are averaged over the ranks (one 0.88 MB all-reduce) and the `pred_conv`
gradients are reduced (one <= 62 KB all-reduce) before the SGD step.  The generator loop is a sequential online process with a
data-dependent number of Adam steps per image, so trajectory parity with the reference is claimed at one GPU only; the
number of collectives per step is fixed (two), whatever each rank's images contain.

noise="cpu" draws z and the sample indices from the CPU default generator exactly like the reference
(:216,:229) -- bit-parity mode for tests; noise="device" draws z with the counter-based device RNG.
"""
import ctypes
import os

import torch

from . import functional as Fz
from . import ops
from ._lib import F, I, P, check, lib, require_gpu, stream


def _rows_gemm(x, wp, b, act=Fz.ACT_NONE, leak=0.2):
    """One Linear layer of the generator on rows.  The output keeps the rows' own element type -- the generators work on fp32 rows in
    either storage mode: without `out_dtype` the launcher allocates the activation storage type, and in the
    2-byte mode the fp32-only row kernels behind this call (zs3_scatter_rows, zs3_gather_rows, the MMD) then read a bf16 buffer as
    fp32, twice past its end (found in round 6 as a once-in-a-few-runs memory fault of the 2-byte GCN-context test: the generated
    features of images with an unseen class)."""
    n, c = x.shape
    y, _ = ops.conv2d_fwd(x.view(1, 1, n, c), wp, shift=b, act=act, leak=leak, out_dtype=x.dtype)
    return y.view(n, wp.cout)


PREP_IN_FWD1 = True   # zs3_gmmn_mlp_fwd1_table: six launches per update instead of seven
# Captured chains of generator updates (table mode): powers of two up to ZS3_GMMN_CHAIN.  A chain boundary costs a graph launch
# (~100-200 us of idle queue when the host is not far enough ahead), so longer chains = fewer boundaries per step.
CHAIN_MAX = max(1, int(os.environ.get("ZS3_GMMN_CHAIN", "32")))
CHAIN_SIZES = tuple(1 << k for k in range(CHAIN_MAX.bit_length() - 1, -1, -1))


# ASPP's atrous branches side by side on the lanes (functional.py) inside this step's frozen-backbone feature pass.  Off: the pass shares
# the chip with the generator loop of the previous batch, whose 64-workgroup launches wait for free CU slots; three streams of big
# convolutions instead of one leave it fewer (same box, interleaved, tools/probe/r6p.sh: 26.18 / 26.21 ms per step with the lanes,
# **25.00 / 24.94 without**; the supervised step, where nothing latency-bound runs beside the pass, keeps them: +2.6 ms without)
FEATURE_LANES = False
# ... and its persistent pointwise launches (conv_pw.hip: 34 per pass, two resident workgroups per CU, each launch holding its CUs for
# 60-140 us) on 192 x 2 workgroups instead of 256 x 2: some CUs stay free for the loop all through those launches.  Same box, interleaved
# (tools/probe/r6i.sh, r6i2.sh): 25.59 / 25.77 -> 24.49 / 24.69 ms on one box, 24.77 / 25.08 -> 24.40 / 24.45 on another (160-224 within
# 0.2 ms of that; 128: 25.2-25.4).  The supervised step keeps 256 (192 there: 42.00 / 42.03 against 42.18 / 42.04, noise).  0 = leave alone.
FEATURE_PW_WGS = 192
# ... and its strip-resident launches with more tiles than this (the decoder's 3x3 layers at 129^2: 2774 tiles, layer 2's at 65^2) on this
# many workgroups that walk the tiles (zs3_conv_halo_set_wgs).  A launch handed out in one go on half the chip: beside it the
# update chain's launches start as if the chip were idle (tools/probe/queue_gate.py: 6.5 us per launch against 47 beside the one-tile-
# per-workgroup launch, 4.8 alone; the convolution itself ~6 % longer at 224 workgroups, ~13 % at 192).  GMMN step, same box x3: 24.18 /
# 24.31 / 24.24 ms one workgroup per tile, 22.95 / 22.64 / 22.85 at 224, 22.47 / 22.27 / 22.63 at 192; another box: 22.48 / 22.73 at 192,
# 21.89 / 22.01 / 22.18 / 21.86 at 160, 21.67 / 21.52 at 144, **21.56 / 21.53 at 128**, 21.86 / 21.89 at 96 (the pass is not the step's critical
# path, the loop is: the decoder's launches may take half the chip).  The same treatment of the
# register-staged kernel (tile loop, grids sized for 224 CUs) bought nothing (23.5 against 23.1-23.5) and was not kept.  0 = one per tile.
FEATURE_HALO_WGS = 128
# The caps above are where the bench's configuration balances (225 updates per step beside one pass: the pass and the loop end together).
# Another class count, crop or batch shifts the balance -- with a short loop the capped pass would be the critical path (the GCN-context
# step: +4 ms) -- so a GMMNStep steers between these (pointwise, strip) settings, one notch per step at most, by what the previous step's
# events say: the pass ended more than FEATURE_SLACK_MS after loop + classifier -> a notch towards the whole chip, the loop ended later
# than the pass -> a notch towards the caps above.  Results do not depend on the notch (the capped launches are bit-identical).
FEATURE_NOTCHES = ((0, 0), (224, 224), (192, 192), (192, 160), (FEATURE_PW_WGS, FEATURE_HALO_WGS))
FEATURE_ADAPT = True
FEATURE_SLACK_MS = 0.4
FEATURE_PLAN = os.environ.get("ZS3_PLAN", "1") == "1"   # the frozen-backbone feature pass replayed from a recorded plan (plan.ForwardPlan)


class GMMNStep:
    _hook_reads_generator = True   # does _after_image() need the generator's weights up to date? (GCNContextStep: no)

    @property
    def group(self):
        """the process group of this step's two exchanges (None: single process), resolved when asked: "auto" follows
        torch.distributed's state"""
        from .parallel import resolve_group
        return resolve_group(self._group_arg)

    @property
    def grad_reduce(self):
        """"sum" when the criterion normalises by the global batch / valid-pixel weight (SegmentationLosses' group), else "mean" """
        if self._grad_reduce_arg is not None:
            return self._grad_reduce_arg
        from .parallel import resolve_group
        owner = getattr(self.criterion, "__self__", None)
        return "sum" if resolve_group(getattr(owner, "group", None)) is not None else "mean"

    def __init__(self, model, generator, optimizer, optimizer_generator, criterion, *, seen, unseen, noise_dim=300,
                 embed_dim=300, feature_dim=256, batch_size_generator=128, real_seen_features=True,
                 sigma=(2, 5, 10, 20, 40, 80), noise="device", use_graph=True, group="auto", grad_reduce=None,
                 context_aware=False, fused_mlp=True):
        """group: "auto" (replicas with parameter averaging as soon as torch.distributed runs with more than one rank; resolved at
        every step) | None (single process) | True (default process group) | a torch.distributed group.
        grad_reduce: "sum" when `criterion` already normalises by the global batch / valid-pixel weight
        (SegmentationLosses(group=...)), "mean" when it normalises per rank; default: picked from the criterion.
        context_aware: the generator's second input is the image's mean embedding over labelled pixels instead of
        noise (train_context_GMMN_GCNcontext.py:345-348)."""
        self.model = model.module if hasattr(model, "module") else model
        if torch.cuda.is_available():
            Fz.warm_streams()        # (hardware queues in a fixed order, the feature stream on its own: functional.warm_streams)
        self.generator = generator
        self.optimizer, self.optimizer_generator = optimizer, optimizer_generator
        self.criterion = criterion
        self.seen, self.unseen = set(int(s) for s in seen), set(int(u) for u in unseen)
        self.noise_dim, self.embed_dim, self.feature_dim = noise_dim, embed_dim, feature_dim
        self.bsg, self.real_seen_features = batch_size_generator, real_seen_features
        self.sigma = tuple(float(s) for s in sigma)
        self.noise = noise
        self.context_aware = bool(context_aware)
        if self.context_aware and noise_dim != embed_dim:
            raise ValueError("context_aware feeds the mean embedding where the noise goes: noise_dim must equal embed_dim")
        self._group_arg = group
        if grad_reduce not in (None, "sum", "mean"):
            raise ValueError("grad_reduce must be 'sum' or 'mean'")
        self._grad_reduce_arg = grad_reduce
        self._dp_ready = False
        self.bytes_reduced = 0
        if not isinstance(generator.model, torch.nn.Sequential):
            raise NotImplementedError("GMMNStep needs the hidden-layer generator (hidden_size > 0)")
        from .optim import Adam
        self.fused_adam = isinstance(optimizer_generator, Adam)
        self.use_graph = use_graph and self.fused_adam
        # the update's MLP forward / backward on the latency-shaped kernels of csrc/gmmn.hip (6 launches per update in table mode
        # with fused Adam, 16 on the general kernels); False keeps the general conv kernels (the A/B reference in tests)
        self.fused_mlp = bool(fused_mlp)
        self._st = None       # static buffers (allocated at first call)
        self._graph = None
        self._update_graphs = {}     # captured update chains by (mode, chain length)
        self._feat_stream = None   # side stream of the pipelined feature pass (prefetch)
        self._prefetched = None    # (image, features, ready event) of the next batch
        self.last_updates = 0
        self._sig = (ctypes.c_float * len(self.sigma))(*self.sigma)

    # ------------------------------------------------------------------ helpers
    def _layers(self):
        m = self.generator.model
        return m[0], m[1], m[2], m[3]

    def _resplit(self):
        """bf16 hi/lo operands of the two Linear layers, rewritten in place (static addresses for the graph)."""
        lin1, _, _, lin2 = self._layers()
        st = self._st
        for lin, wp in ((lin1, st["wp1"]), (lin2, st["wp2"])):
            check(lib().zs3_prep_weight(P(lin.weight), P(wp.f_pk), P(wp.t_pk), I(wp.cout), I(1), I(wp.cin), I(wp.cin_pad),
                                        I(wp.cout_pad), stream()), "zs3_prep_weight")

    def _alloc(self, dev, b, npix):
        lin1, _, _, lin2 = self._layers()
        s, d, e, nz = self.bsg, self.feature_dim, self.embed_dim, self.noise_dim
        i64 = dict(dtype=torch.int64, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        st = {
            "emb": torch.zeros((npix, e), **f32), "real": torch.zeros((b * npix, d), **f32), "emb_all": None, "upd_host": None,
            "pix_local": torch.zeros(s, **i64), "pix_global": torch.zeros(s, **i64), "ridx": torch.zeros(s, **i64),
            "z": torch.zeros((s, nz), **f32), "loss": torch.zeros(1, **f32), "one": torch.ones(1, **f32),
            "seed_dev": torch.zeros(1, **i64), "step_dev": torch.zeros(1, **i64),
            "loss_ring": torch.zeros(4096, **f32), "slot_dev": torch.zeros(1, **i64),
            "wp1": ops.prep_weight(lin1.weight, need_t=True), "wp2": ops.prep_weight(lin2.weight, need_t=True),
            "ring": torch.zeros((512, s), dtype=torch.int64).pin_memory(), "ring_pos": 0,
            "seed_base": Fz.next_seed(), "shape": (b, npix), "ident": torch.arange(s, **i64),
        }
        self._st = st
        for name, prm in (("dw1", lin1.weight), ("db1", lin1.bias), ("dw2", lin2.weight), ("db2", lin2.bias)):
            st[name] = torch.zeros(prm.shape, **f32)
        if self.fused_adam:
            opt = self.optimizer_generator
            opt._step_dev = None
            for group in opt.param_groups:
                for p in group["params"]:
                    state = opt.state[p]
                    if len(state) == 0:
                        state["step"] = torch.tensor(0.0)
                        state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                        state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            any_p = opt.param_groups[0]["params"][0]
            st["step_dev"].fill_(int(opt.state[any_p]["step"]))
            # one-launch Adam (+ in-place re-split of the two Linear weights) when the four generator tensors form a
            # single param group with plain contiguous storage
            st["adam_multi"] = None
            grads = {lin1.weight: st["dw1"], lin1.bias: st["db1"], lin2.weight: st["dw2"], lin2.bias: st["db2"]}
            planes = {lin1.weight: st["wp1"], lin2.weight: st["wp2"]}
            groups = [g for g in opt.param_groups if any(p in grads for p in g["params"])]
            if len(groups) == 1 and all(p.is_contiguous() for p in grads):
                chunk = lib().zs3_adam_chunk()
                recs, bmap = [], []
                for e, (p, g) in enumerate(grads.items()):
                    state, wp = opt.state[p], planes.get(p)
                    recs.append((p.data_ptr(), g.data_ptr(), state["exp_avg"].data_ptr(), state["exp_avg_sq"].data_ptr(),
                                 p.numel(), wp.f_pk.data_ptr() if wp else 0, wp.t_pk.data_ptr() if wp else 0,
                                 wp.cout if wp else 0, wp.cin if wp else 1, wp.cin_pad if wp else 32,
                                 wp.cout_pad if wp else 32))
                    bmap.extend((e, c) for c in range((p.numel() + chunk - 1) // chunk))
                st["adam_multi"] = (torch.tensor(recs, dtype=torch.int64).to(dev),
                                    torch.tensor(bmap, dtype=torch.int32).to(dev), len(bmap), groups[0])
                # the same update fused into the weight-gradient launch (zs3_gmmn_mlp_wgrad_adam): per layer
                # {weight, exp_avg, exp_avg_sq, bias, bias exp_avg, bias exp_avg_sq, f_pk, t_pk}; layer 1's transposed planes
                # are never read (the generator's input needs no gradient) and are not maintained
                def _state(lin, wp, with_t):
                    sw, sb = opt.state[lin.weight], opt.state[lin.bias]
                    ptrs = [lin.weight, sw["exp_avg"], sw["exp_avg_sq"], lin.bias, sb["exp_avg"], sb["exp_avg_sq"], wp.f_pk,
                            wp.t_pk if with_t else None]
                    return (ctypes.c_void_p * 8)(*[None if t is None else t.data_ptr() for t in ptrs])
                st["adam_state2"], st["adam_state1"] = _state(lin2, st["wp2"], True), _state(lin1, st["wp1"], False)
                st["done"] = torch.zeros(1, dtype=torch.int32, device=dev)
                st["adam_bc"] = torch.ones(2, dtype=torch.float32, device=dev)   # bias corrections of the running update

    def _adam_signature(self):
        """identity of everything the captured update holds raw pointers to / mirrors on the device"""
        if not self.fused_adam:
            return None
        opt = self.optimizer_generator
        sig = []
        for group in opt.param_groups:
            for p in group["params"]:
                stt = opt.state.get(p, {})
                sig.append((p.data_ptr(), stt["exp_avg"].data_ptr() if "exp_avg" in stt else 0,
                            stt["exp_avg_sq"].data_ptr() if "exp_avg_sq" in stt else 0, int(stt["step"]) if "step" in stt else -1))
        return tuple(sig)

    # ------------------------------------------------------------------ the fixed-shape sampled-row update
    def _sampled_update(self, training):
        """gather -> generator -> MMD -> generator backward -> Adam -> weight re-split, all on S = bsg rows that sit
        in static buffers.  Captured into a hipGraph; every launch is on the current (capture) stream."""
        st = self._st
        lin1, lrelu, drop, lin2 = self._layers()
        s, d = self.bsg, self.feature_dim
        width = self.embed_dim + self.noise_dim
        use_drop = training and drop.p > 0
        dseed = st["seed_base"] ^ 0x5DEECE66D
        wp1, wp2 = st["wp1"], st["wp2"]
        device_noise = self.noise != "cpu" and not self.context_aware
        fused = (self.fused_mlp and s <= 128 and width % 4 == 0 and self.embed_dim % 4 == 0 and wp1.cin_pad <= 640 and
                 wp2.cin_pad <= 640 and wp2.cout_pad <= 640 and d % 4 == 0 and wp1.cout % 4 == 0)
        t = (2 * s + 31) // 32
        gmat = torch.empty((2 * s, 2 * s), dtype=torch.float32, device=st["emb"].device)
        tile = torch.empty(2 * t * t, dtype=torch.float64, device=st["emb"].device)
        fused_adam = fused and st.get("adam_multi") is not None and st.get("adam_state2") is not None
        adam_b1, adam_b2 = st["adam_multi"][3]["betas"] if fused_adam else (0.9, 0.999)
        bc_ready = False     # Adam's bias corrections of this update already on the device (written by zs3_gmmn_prep)
        if fused:
            dev = st["emb"].device
            hid = wp1.cout
            x = torch.empty((s, width), dtype=torch.float32, device=dev)
            h = torch.empty((s, hid), dtype=torch.float32, device=dev)
            hd = torch.empty((s, hid), dtype=torch.float32, device=dev)
            gen_s = torch.empty((s, d), dtype=torch.float32, device=dev)
            real_s = torch.empty((s, d), dtype=torch.float32, device=dev)
            if device_noise and st.get("table_mode") and PREP_IN_FWD1:
                # table-driven, SIX launches per update: the first GEMM's workgroups read the update's row of the step's device
                # table themselves (sample indices -> pixel -> embedding row + noise), publish pix_global / ridx for the later
                # launches and the Adam bias corrections -- what zs3_gmmn_prep did in a launch of its own
                check(lib().zs3_gmmn_mlp_fwd1_table(P(st["upd_table"]), I(st["upd_table"].stride(0)), P(st["slot_dev"]),
                                                    P(st["order_flat"]), P(st["emb_all"]), I(st["emb_all"].stride(0)),
                                                    I(self.embed_dim), I(self.noise_dim), P(wp1.f_pk), I(wp1.cin_pad // 32),
                                                    P(lin1.bias), P(x), I(width), P(h), P(hd), I(hid), I(s), I(hid),
                                                    F(lrelu.negative_slope), F(drop.p if use_drop else 0.0),
                                                    ctypes.c_ulonglong(st["seed_base"]), ctypes.c_ulonglong(dseed), P(st["seed_dev"]),
                                                    P(st["pix_global"]), P(st["ridx"]), P(st["step_dev"] if fused_adam else None),
                                                    F(adam_b1), F(adam_b2), P(st["adam_bc"] if fused_adam else None), stream()),
                      "zs3_gmmn_mlp_fwd1_table")
                bc_ready = fused_adam
            elif device_noise and st.get("table_mode"):
                # table-driven: the update's (image, class) and sample indices come from row slot_dev[0] of the step's
                # device table (no host argument, nothing to copy per update); x is assembled once, the GEMM reads it plain
                check(lib().zs3_gmmn_prep(P(st["upd_table"]), I(st["upd_table"].stride(0)), P(st["slot_dev"]), P(st["order_flat"]),
                                          P(st["emb_all"]), I(st["emb_all"].stride(0)), I(self.embed_dim), I(self.noise_dim), P(x),
                                          I(width), P(st["pix_global"]), P(st["ridx"]), I(s), ctypes.c_ulonglong(st["seed_base"]),
                                          P(st["seed_dev"]), P(st["step_dev"] if fused_adam else None), F(adam_b1), F(adam_b2),
                                          P(st["adam_bc"] if fused_adam else None), stream()), "zs3_gmmn_prep")
                bc_ready = fused_adam
                check(lib().zs3_gmmn_mlp_fwd1(P(x), I(width), P(None), P(st["ridx"]), I(width), I(0), P(wp1.f_pk),
                                              I(wp1.cin_pad // 32), P(lin1.bias), P(None), I(width), P(h), P(hd), I(hid), I(s),
                                              I(hid), F(lrelu.negative_slope), F(drop.p if use_drop else 0.0),
                                              ctypes.c_ulonglong(0), ctypes.c_ulonglong(dseed), P(st["seed_dev"]), stream()),
                      "zs3_gmmn_mlp_fwd1")
            elif device_noise:   # rows gathered and noise drawn inside the first GEMM's operand load
                check(lib().zs3_gmmn_mlp_fwd1(P(st["emb"]), I(st["emb"].stride(0)), P(st["pix_local"]), P(st["ridx"]),
                                              I(self.embed_dim), I(self.noise_dim), P(wp1.f_pk), I(wp1.cin_pad // 32),
                                              P(lin1.bias), P(x), I(width), P(h), P(hd), I(hid), I(s), I(hid),
                                              F(lrelu.negative_slope), F(drop.p if use_drop else 0.0),
                                              ctypes.c_ulonglong(st["seed_base"]), ctypes.c_ulonglong(dseed), P(st["seed_dev"]),
                                              stream()), "zs3_gmmn_mlp_fwd1")
            else:                # noise from the host (reference RNG stream) / context vector: z sits in st["z"]
                x = ops.gather_cat(st["emb"], st["pix_local"], self.embed_dim, st["z"], self.noise_dim, width)
                ident = st["ident"]
                # Cb = 0: the whole row is "embedding" = the already assembled x
                check(lib().zs3_gmmn_mlp_fwd1(P(x), I(width), P(ident), P(st["ridx"]), I(width), I(0), P(wp1.f_pk),
                                              I(wp1.cin_pad // 32), P(lin1.bias), P(None), I(width), P(h), P(hd), I(hid), I(s),
                                              I(hid), F(lrelu.negative_slope), F(drop.p if use_drop else 0.0),
                                              ctypes.c_ulonglong(0), ctypes.c_ulonglong(dseed), P(st["seed_dev"]), stream()),
                      "zs3_gmmn_mlp_fwd1")
            check(lib().zs3_gmmn_mlp_fwd2(P(hd), I(hid), P(wp2.f_pk), I(wp2.cin_pad // 32), P(lin2.bias), P(gen_s), I(d), I(s),
                                          I(d), I(hid), P(st["real"]), I(st["real"].stride(0)), P(st["pix_global"]), P(real_s),
                                          stream()), "zs3_gmmn_mlp_fwd2")
        else:
            if device_noise:   # noise drawn inside the gather (same stream as zs3_uniform on a [S, noise_dim] tensor)
                x = ops.gather_cat_noise(st["emb"], st["pix_local"], self.embed_dim, self.noise_dim, width, s, st["seed_base"],
                                         seed_dev=st["seed_dev"], noise_key=st["ridx"])   # z[random_idx]: duplicates share noise
            else:
                x = ops.gather_cat(st["emb"], st["pix_local"], self.embed_dim, st["z"], self.noise_dim, width)
            h = _rows_gemm(x, st["wp1"], lin1.bias, Fz.ACT_LEAKY, lrelu.negative_slope)
            hd = ops.dropout(h, drop.p, dseed, row_idx=st["ridx"], seed_dev=st["seed_dev"]) if use_drop else h
            gen_s = _rows_gemm(hd, st["wp2"], lin2.bias)
            real_s = ops.gather_rows(st["real"], st["pix_global"])
        check(lib().zs3_mmd_fwd(P(gen_s), I(d), P(real_s), I(d), I(s), I(d), self._sig, I(len(self.sigma)), P(gmat), P(tile),
                                None, stream()), "zs3_mmd_fwd")          # the loss value is finalised by the update epilogue
        dgen = torch.empty_like(gen_s)
        # fused_adam: the loss value goes to loss_ring[slot] here, the counters are advanced by the wgrad + Adam launch
        check(lib().zs3_mmd_bwd_ws(P(gen_s), I(d), P(real_s), I(d), I(s), I(d), P(gmat), P(tile), P(st["one"]), P(dgen), I(d),
                                   P(st["loss_ring"]) if fused_adam else None, P(st["slot_dev"]) if fused_adam else None,
                                   I(st["loss_ring"].numel()), stream()), "zs3_mmd_bwd_ws")
        # generator backward on the sampled rows, gradients into static buffers
        if fused:
            dpre = torch.empty((s, wp1.cout), dtype=torch.float32, device=dgen.device)
            check(lib().zs3_gmmn_mlp_dgrad(P(dgen), I(d), P(wp2.t_pk), I(wp2.cout_pad // 32), P(h), I(h.stride(0)), P(st["ridx"]),
                                           P(dpre), I(wp1.cout), I(s), I(wp2.cin), I(wp2.cout), F(lrelu.negative_slope),
                                           F(drop.p if use_drop else 0.0), ctypes.c_ulonglong(dseed), P(st["seed_dev"]),
                                           stream()), "zs3_gmmn_mlp_dgrad")
            if fused_adam:   # gradients consumed tile by tile: Adam, the new weights' bf16 planes and the update's counters
                group = st["adam_multi"][3]
                b1, b2 = group["betas"]
                check(lib().zs3_gmmn_mlp_wgrad_adam(P(dgen), I(d), P(hd), I(hd.stride(0)), I(wp2.cout), I(wp2.cin), P(dpre),
                                                    I(wp1.cout), P(x), I(width), I(wp1.cout), I(wp1.cin), I(s),
                                                    st["adam_state2"], st["adam_state1"], I(wp2.cin_pad), I(wp2.cout_pad),
                                                    I(wp1.cin_pad), I(wp1.cout_pad), F(group["lr"]), F(b1), F(b2),
                                                    F(group["eps"]), F(group["weight_decay"]), P(st["slot_dev"]),
                                                    P(st["step_dev"]), P(st["seed_dev"]), ctypes.c_long(1 << 24), P(st["done"]),
                                                    P(st["adam_bc"] if bc_ready else None), stream()), "zs3_gmmn_mlp_wgrad_adam")
                return
            check(lib().zs3_gmmn_mlp_wgrad(P(dgen), I(d), P(hd), I(hd.stride(0)), I(wp2.cout), I(wp2.cin), P(st["dw2"]),
                                           P(st["db2"]), P(dpre), I(wp1.cout), P(x), I(width), I(wp1.cout), I(wp1.cin),
                                           P(st["dw1"]), P(st["db1"]), I(s), stream()), "zs3_gmmn_mlp_wgrad")
        else:
            ops.conv2d_wgrad(dgen.view(1, 1, s, -1), hd.view(1, 1, s, -1), wp2.cout, wp2.cin, 1, 1, out=st["dw2"])
            ops.colsum(dgen, out=st["db2"])
            dhd = ops.conv2d_dgrad(dgen.view(1, 1, s, -1), wp2, (1, s), out_dtype=dgen.dtype).view(s, -1)     # (fp32 rows: see _rows_gemm)
            dpre = ops.dropout_act_bwd(dhd, h, drop.p if use_drop else 0.0, dseed, lrelu.negative_slope, row_idx=st["ridx"],
                                       seed_dev=st["seed_dev"])
            ops.conv2d_wgrad(dpre.view(1, 1, s, -1), x.view(1, 1, s, -1), wp1.cout, wp1.cin, 1, 1, out=st["dw1"])
            ops.colsum(dpre, out=st["db1"])
        opt = self.optimizer_generator
        multi = st.get("adam_multi")
        if multi is not None:
            table, bmap, nblk, group = multi
            b1, b2 = group["betas"]
            check(lib().zs3_adam_multi(P(table), P(bmap), I(nblk), F(group["lr"]), F(b1), F(b2), F(group["eps"]),
                                       F(group["weight_decay"]), P(st["step_dev"]), stream()), "zs3_adam_multi")
            self._epilogue(tile)
            return
        grads = {lin1.weight: st["dw1"], lin1.bias: st["db1"], lin2.weight: st["dw2"], lin2.bias: st["db2"]}
        for group in opt.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                g = grads.get(p)
                if g is None:
                    continue
                state = opt.state[p]
                check(lib().zs3_adam_step(P(p), P(g.contiguous()), P(state["exp_avg"]), P(state["exp_avg_sq"]),
                                          ctypes.c_long(p.numel()), F(group["lr"]), F(b1), F(b2), F(group["eps"]),
                                          F(group["weight_decay"]), I(0), P(st["step_dev"]), stream()), "zs3_adam_step")
        self._epilogue(tile)
        self._resplit()

    def _epilogue(self, tile):
        """loss value -> loss_ring[slot++], Adam step and RNG position advanced: one launch"""
        st = self._st
        check(lib().zs3_gmmn_update_epilogue(P(tile), I(self.bsg), P(st["loss_ring"]), P(st["slot_dev"]),
                                             I(st["loss_ring"].numel()), P(st["step_dev"]), P(st["seed_dev"]),
                                             ctypes.c_long(1 << 24), stream()), "zs3_gmmn_update_epilogue")

    def _run_sampled_update(self, training, count=1):
        """`count` consecutive updates.  Table mode: every update reads its own row of the step's device table, so captured
        chains of 32 ... 1 updates are replayed back to back with no host work in between; otherwise one update."""
        if not self.use_graph:
            for _ in range(count):
                self._sampled_update(training)
        else:
            table_mode = bool(self._st.get("table_mode"))
            left = count
            for size in (CHAIN_SIZES if table_mode else (1,)):
                while left >= size:
                    key = (training, self.noise, self.context_aware, table_mode, size)
                    g = self._update_graphs.get(key)
                    if g is None:
                        g = torch.cuda.CUDAGraph()
                        torch.cuda.synchronize()
                        with torch.cuda.graph(g):
                            for _ in range(size):
                                self._sampled_update(training)
                        self._update_graphs[key] = g
                    g.replay()
                    left -= size
        # host-side mirror of the step count (state_dict compatibility with torch.optim.Adam)
        opt = self.optimizer_generator
        for group in opt.param_groups:
            for p in group["params"]:
                if p in opt.state and "step" in opt.state[p]:
                    opt.state[p]["step"] += count
        Fz.invalidate_planes(*[p for g_ in opt.param_groups for p in g_["params"]])
        self._st["adam_sig"] = self._adam_signature()

    # ------------------------------------------------------------------ eager generator pieces (fallback + unseen images)
    def _generator_forward(self, x, training):
        lin1, lrelu, drop, lin2 = self._layers()
        st = self._st
        h = _rows_gemm(x, st["wp1"], lin1.bias, Fz.ACT_LEAKY, lrelu.negative_slope)
        seed, hd = None, h
        if training and drop.p > 0:
            seed = Fz.next_seed()
            hd = ops.dropout(h, drop.p, seed)
        return _rows_gemm(hd, st["wp2"], lin2.bias), h, hd, seed

    def _eager_update_rows(self, x, h, hd, seed, ridx, d_out):
        """Reference-order fallback (torch.optim.Adam supplied by the caller): gradients from the sampled rows."""
        lin1, lrelu, drop, lin2 = self._layers()
        st = self._st
        s = ridx.shape[0]
        wp1, wp2 = st["wp1"], st["wp2"]
        hd_s = ops.gather_rows(hd, ridx)
        dw2 = ops.conv2d_wgrad(d_out.view(1, 1, s, -1), hd_s.view(1, 1, s, -1), wp2.cout, wp2.cin, 1, 1)
        db2 = ops.colstats(d_out)[:, 0].sum(0)
        dhd = ops.conv2d_dgrad(d_out.view(1, 1, s, -1), wp2, (1, s), out_dtype=d_out.dtype).view(s, -1)     # (fp32 rows: see _rows_gemm)
        if seed is not None:
            dhd = ops.dropout(dhd, drop.p, seed, row_idx=ridx)
        h_s = ops.gather_rows(h, ridx)
        dpre = torch.empty_like(dhd)
        ops.bn_act_bwd(dhd, h_s, None, None, None, None, None, None, dres=dpre, act=Fz.ACT_LEAKY, leak=lrelu.negative_slope,
                       want_dy=False)
        x_s = ops.gather_rows(x, ridx)
        dw1 = ops.conv2d_wgrad(dpre.view(1, 1, s, -1), x_s.view(1, 1, s, -1), wp1.cout, wp1.cin, 1, 1)
        db1 = ops.colstats(dpre)[:, 0].sum(0)
        lin1.weight.grad, lin1.bias.grad = dw1.view(lin1.weight.shape), db1
        lin2.weight.grad, lin2.bias.grad = dw2.view(lin2.weight.shape), db2
        self.optimizer_generator.step()
        self._resplit()

    # ------------------------------------------------------------------ extension points (GCN-context step)
    def _before_images(self, label_maps):
        """called once per step with the [B, fh, fw] label maps at feature resolution, before the step's host read-back"""

    def _after_image(self, i, label_map, real_rows_i, has_unseen):
        """called after image i's generator updates; self._st["emb"] holds its embedding rows at feature resolution"""

    def _join_side_work(self):
        pass

    def _extra_classifier_terms(self):
        """called after the CE backward of the stitched batch, before the gradient exchange and the SGD step"""

    def _replica_parameters(self):
        """parameters averaged over the ranks once per step in multi-GPU runs"""
        return list(self.generator.parameters())

    def _replica_modules(self):
        return [self.generator]

    # ------------------------------------------------------------------ frozen-backbone feature pass, pipelined
    def _feature_shaping(self):
        """(ASPP lanes, persistent pointwise workgroups, strip-kernel workgroups) of this step's frozen feature pass: how much of the chip
        a pass that runs beside the generator loop leaves to it.  Tuned for THIS step's balance (the loop is its critical path);
        subclasses with another balance override it (GCNContextStep: the pass is the critical path there)."""
        if not FEATURE_ADAPT:
            return FEATURE_LANES, FEATURE_PW_WGS, FEATURE_HALO_WGS
        pw, halo = FEATURE_NOTCHES[self.__dict__.setdefault("_feature_notch", len(FEATURE_NOTCHES) - 1)]
        return FEATURE_LANES, pw, halo

    def _steer_feature_shaping(self, loop_end, step_end):
        """one notch per step from the events of the step that just ended: `loop_end` / `step_end` on the main stream (behind the
        generator loop / behind the classifier update), the prefetched pass's completion event on the feature stream"""
        pf = self._prefetched
        if not FEATURE_ADAPT or pf is None or len(pf) < 4 or type(self)._feature_shaping is not GMMNStep._feature_shaping:
            return
        notch = self.__dict__.setdefault("_feature_notch", len(FEATURE_NOTCHES) - 1)
        done = pf[3]
        if not done.query():                                   # the pass is still running behind the step's last launch: it is the critical path
            notch -= 1
        else:
            if step_end.elapsed_time(done) > FEATURE_SLACK_MS:   # (the read-back that follows waited for the main stream only)
                notch -= 1
            elif loop_end.elapsed_time(done) < -FEATURE_SLACK_MS:  # the pass ended before the loop did: the loop is the critical path
                notch += 1
        self.__dict__["_feature_notch"] = min(max(notch, 0), len(FEATURE_NOTCHES) - 1)

    def _features_eager(self, image):
        # [B, fh, fw, D]; in the 2-byte mode the backbone hands over bf16 features: the generator loop, the MMD kernels and the
        # cluster graphs work on fp32 rows (273 MB at B = 16: one cast pass, ~0.1 ms)
        lanes = Fz.ASPP_CONCURRENT
        if not self._feature_shaping()[0]:
            Fz.ASPP_CONCURRENT = False
        try:
            return ops.cast(ops.nhwc(self.model.forward_before_class_prediction(image)), torch.float32)
        finally:
            Fz.ASPP_CONCURRENT = lanes

    def _features(self, image):
        """The frozen-backbone feature pass as a recorded launch plan (zs3_amd.plan.ForwardPlan): ~330 launches whose Python enqueue
        took as long as the pass runs on the GPU (14-15 ms: the next batch's pass could not start overlapping the generator loop
        until the host was done with it) replayed with one C call."""
        # (the persistent pointwise kernel's launches size themselves when they are enqueued -- recorded or replayed -- from this setting)
        # -- only for a pass that runs BESIDE a generator loop (prefetch): alone on the chip it takes all of it
        beside = bool(self.__dict__.get("_prefetching"))
        _, pw_wgs, halo_wgs = self._feature_shaping()
        prev = lib().zs3_conv_pw_set_wgs(pw_wgs) if pw_wgs and beside else 0
        prev_halo = lib().zs3_conv_halo_set_wgs(halo_wgs) if halo_wgs and beside else None
        try:
            with torch.no_grad():
                if not FEATURE_PLAN:
                    return self._features_eager(image)
                fp = self.__dict__.get("_feature_plan")
                if fp is None:
                    from .plan import ForwardPlan
                    fp = self.__dict__["_feature_plan"] = ForwardPlan(self._features_eager, [self.model])
                return fp(image)
        finally:
            if prev:
                lib().zs3_conv_pw_set_wgs(prev)
            if prev_halo is not None:
                lib().zs3_conv_halo_set_wgs(prev_halo)

    def prefetch(self, image):
        """Start the feature pass of the NEXT batch on a side stream.  The backbone is frozen in this step (only `pred_conv`
        and the generator train, train_pascal_GMMN.py:262-264), so the next batch's features do not depend on anything the
        current batch's generator loop or classifier update writes: the big convolutions of batch t+1 fill the chip while
        the latency-bound per-(image, class) update chain of batch t runs.  Batches are still visited in loader order, so
        the BatchNorm running statistics see the same sequence of updates."""
        require_gpu(image)
        dev = image.device
        if self._feat_stream is None:
            # (a CU-masked stream that leaves 16-64 CUs to the generator loop -- hipExtStreamCreateWithCUMask -- was measured:
            # 61-65 ms per step instead of 38.5: the masked queue slows the convolutions far more than the loop gains)
            self._feat_stream = Fz.feature_stream(dev)
        self._feat_stream.wait_stream(torch.cuda.current_stream(dev))    # the image (and the previous feature pass) are ready
        with torch.cuda.stream(self._feat_stream):
            _tick("feat-begin", self._feat_stream)
            self.__dict__["_prefetching"] = True
            try:
                real = self._features(image)
            finally:
                self.__dict__["_prefetching"] = False
            _tick("feat-end", self._feat_stream)
            done = torch.cuda.Event()
            done.record()
            timed = torch.cuda.Event(enable_timing=True)      # (for _steer_feature_shaping; the wait below uses `done`)
            timed.record()
        self._prefetched = (image, real, done, timed)

    def _take_features(self, image):
        pf, self._prefetched = self._prefetched, None
        if pf is not None and pf[0] is image:
            main = torch.cuda.current_stream(image.device)
            main.wait_event(pf[2])
            pf[1].record_stream(main)
            return pf[1]
        if pf is not None:
            # the prefetched pass already updated the BN running statistics (train() mode, train_pascal_GMMN.py:136,154) and
            # drew its dropout seeds for a batch that is now dropped: recomputing silently would count one batch twice
            import warnings
            warnings.warn("GMMNStep: the batch passed to prefetch()/next_image is not the tensor object of this call; its "
                          "feature pass is discarded (the BatchNorm running statistics have seen that batch once already)",
                          RuntimeWarning, stacklevel=3)
        return self._features(image)

    # ------------------------------------------------------------------ one iteration
    def __call__(self, image, target, embedding=None, table=None, next_image=None):
        """embedding: the reference's `label_emb` [B, embed_dim, H, W] (zs3/dataloaders/datasets/base.py:45-51), or
        table: the [num_classes, embed_dim] class-embedding table itself -- then the per-pixel embedding rows are looked
        up on the device at feature resolution (embed(nearest(label)) == nearest(embed(label)) exactly),
        which avoids materialising and shipping 316 MB per 513x513 sample."""
        if (embedding is None) == (table is None):
            raise ValueError("pass exactly one of `embedding` (label_emb) or `table`")
        require_gpu(image, target, embedding, table)
        model, dev = self.model, image.device
        b = image.shape[0]
        if not self._dp_ready and self.group is not None:
            # first step of a multi-rank run (every rank is here): all replicas start from rank 0's segmentation model and
            # generator(s) -- nn.DataParallel's replicate guarantee, made once instead of at every forward
            from .parallel import broadcast_parameters
            pg = None if self.group is True else self.group
            broadcast_parameters(model, group=pg)
            for mod in self._replica_modules():
                broadcast_parameters(mod, group=pg)
            if self._st is not None:
                self._resplit()
            self._dp_ready = True
        real = self._take_features(image)
        _tick("take")
        if next_image is not None:      # the caller already knows the next batch: overlap its feature pass with this loop
            self.prefetch(next_image)   # (queued before the loop: 38.5 ms per step; after it: 39.9 ms; with the loop on a
            # high-priority stream: 101 ms -- HIP priority streams misbehave on this runtime, as in round 1)
            _tick("prefetch")
        fh, fw, d = real.shape[1], real.shape[2], real.shape[3]
        npix = fh * fw
        if self._st is None or self._st["shape"] != (b, npix) or self._st.get("adam_sig") != self._adam_signature():
            # (re)build the static buffers, pointer tables and captured updates: first call, another batch shape, or the
            # generator's optimizer state was replaced behind our back (load_state_dict: new moment tensors, another step count)
            self._alloc(dev, b, npix)
            self._graph = None
            self._update_graphs = {}
            self._st["adam_sig"] = self._adam_signature()
        st = self._st
        self._resplit()   # the generator may have been changed from outside (load_state_dict, another optimizer)
        real_rows = real.reshape(b, npix, d)
        st["real"].copy_(real_rows.reshape(b * npix, d))
        fake = torch.empty((b, fh, fw, d), dtype=torch.float32, device=dev)
        fake_rows = fake.view(b, npix, d)
        # labels at feature resolution (nearest), per-image class histogram, pixels grouped by class: one launch, then the one
        # host sync of the step's head
        tgt_l, tgt_cls, hist, order = ops.label_order(target, (fh, fw))                           # [B, npix] int64 each
        self._before_images(tgt_l.view(b, fh, fw))
        hist_h = hist.cpu().tolist()
        _tick("hist")
        if table is not None:
            table_f = table.contiguous().float()
        training = self.generator.training
        n_mmd = int(sum(1 for i in range(b) for c in range(255) if hist_h[i][c] > 0))
        mmd_losses = torch.zeros(max(n_mmd, 1), dtype=torch.float32, device=dev)
        mmd_slots, slot = [], 0
        ring_slots, n_ring = [], 0
        st["slot_dev"].zero_()
        # table mode (device noise, fused kernels): all sample indices of the step are drawn up front in the reference's
        # order (torch.randint per (image, class), train_pascal_GMMN.py:229) and shipped in ONE copy together with each
        # update's class segment; the captured update reads its row by a device counter (zs3_gmmn_prep)
        table_mode = bool(self.noise == "device" and not self.context_aware and self.fused_mlp and self.fused_adam and
                          self.use_graph and self.real_seen_features and self.bsg <= 128)
        st["table_mode"] = table_mode
        if table_mode:
            rows = []
            for i in range(b):
                cls_i = [c for c in range(256) if hist_h[i][c] > 0]
                if any(c in self.unseen for c in cls_i):
                    continue
                off = 0
                for c in cls_i:
                    n_c = int(hist_h[i][c])
                    if c != 255 and c in self.seen:
                        rows.append((i * npix + off, i * npix, n_c))
                    off += n_c
            need = max(len(rows), 1)
            if st.get("upd_host") is None or st["upd_host"].shape[0] < need:
                cap = max(need, 256)          # static addresses: the captured updates hold these pointers
                st["upd_host"] = torch.zeros((cap, self.bsg + 2), dtype=torch.int64).pin_memory()
                st["upd_table"] = torch.zeros((cap, self.bsg + 2), dtype=torch.int64, device=dev)
                self._update_graphs = {}
            host = st["upd_host"]
            if rows:    # with replacement, uniform on [0, n_c): floor(U * n_c) with U from the CPU generator, one call per step
                meta = torch.tensor(rows, dtype=torch.int64)
                n_c = meta[:, 2:3].double()
                host[:len(rows), :self.bsg] = torch.minimum((torch.rand((len(rows), self.bsg), dtype=torch.float64) * n_c).long(),
                                                            meta[:, 2:3] - 1)
                host[:len(rows), self.bsg:] = meta[:, :2]
            st["upd_table"][:need].copy_(host[:need], non_blocking=True)
            if st.get("emb_all") is None:
                st["emb_all"] = torch.empty((b * npix, self.embed_dim), dtype=torch.float32, device=dev)
                st["order_flat"] = torch.empty(b * npix, dtype=torch.int64, device=dev)
            st["order_flat"].copy_(order.reshape(-1))
            if table is not None:     # one lookup for the whole batch (label 255 -> class 0 like base.py:47-48)
                check(lib().zs3_gather_rows(P(table_f), I(self.embed_dim), P(tgt_cls), P(st["emb_all"]), I(self.embed_dim),
                                            ctypes.c_long(b * npix), I(self.embed_dim), stream()), "zs3_gather_rows")
        # pinned staging ring of the sample indices: one row per update of THIS step (the host syncs once, at the end of the
        # step, so a row must not be reused before that); every copy of the previous step completed at its read-back
        if n_mmd > st["ring"].shape[0]:
            st["ring"] = torch.zeros((n_mmd, self.bsg), dtype=torch.int64).pin_memory()
        st["ring_pos"] = 0
        _tick("table")
        pending = 0        # table mode: sampled updates queued but not yet replayed
        # a subclass hook after every image forces the queued generator updates out first only if it reads the generator
        per_image_hook = type(self)._after_image is not GMMNStep._after_image and self._hook_reads_generator
        for i in range(b):
            classes = [c for c in range(256) if hist_h[i][c] > 0]
            has_unseen = any(c in self.unseen for c in classes)
            use_real = self.real_seen_features and not has_unseen
            if table_mode:     # this image's rows of the batch-wide embedding buffer
                st["emb"] = st["emb_all"][i * npix:(i + 1) * npix]
            if embedding is not None:
                check(lib().zs3_nearest_rows(P(embedding[i].contiguous()), I(self.embed_dim), I(embedding.shape[2]),
                                             I(embedding.shape[3]), I(fh), I(fw), P(st["emb"]), I(self.embed_dim), stream()),
                      "zs3_nearest_rows")
            elif not table_mode:  # label 255 -> class 0 like the dataloader (base.py:47-48); those rows are never used
                check(lib().zs3_gather_rows(P(table_f), I(self.embed_dim), P(tgt_cls[i]), P(st["emb"]), I(self.embed_dim),
                                            ctypes.c_long(npix), I(self.embed_dim), stream()), "zs3_gather_rows")
            if has_unseen and pending:   # the eager generator calls below read the weights: every queued update first
                self._run_sampled_update(training, pending)
                pending = 0
            if use_real:
                fake_rows[i].copy_(real_rows[i])
            else:
                fake_rows[i].zero_()   # ignore-label pixels keep zero features (:198, :242)
            ctx = None
            if self.context_aware:    # mean embedding over the labelled pixels (255 sorts last in `order`)
                n_valid = npix - int(hist_h[i][255])
                ctx = ops.colsum(ops.gather_rows(st["emb"], order[i, :n_valid])) / max(n_valid, 1)
                st["z"].copy_(ctx.view(1, -1).expand(self.bsg, -1))
            if has_unseen and self.fused_adam:
                # No class of an image that contains an unseen class trains the generator (:224); all that is left of its
                # class loop is "generate features for every labelled pixel" (:242).  Rows of the MLP are independent and
                # `order` lists the pixels class by class, so ONE generator call over all labelled pixels replaces one per
                # class (6 launches per image instead of 6 per class).  The CPU noise stream is drawn class by class in the
                # reference's order.
                n_valid = npix - int(hist_h[i][255])
                if n_valid > 0:
                    idx_all = order[i, :n_valid]
                    if ctx is not None:
                        z = ctx.view(1, -1).expand(n_valid, -1).contiguous()
                    elif self.noise == "cpu":
                        z = torch.cat([torch.rand((int(hist_h[i][c]), self.noise_dim)) for c in classes if c != 255]).to(dev)
                    else:
                        z = ops.uniform((n_valid, self.noise_dim), Fz.next_seed(), dev)
                    x = ops.gather_cat(st["emb"], idx_all, self.embed_dim, z, self.noise_dim, self.embed_dim + self.noise_dim)
                    fake_all, _, _, _ = self._generator_forward(x, training)
                    if not use_real:
                        ops.scatter_rows(fake_all, idx_all, fake_rows[i])
                classes = []      # nothing left for the per-class loop below
            off = 0
            for c in classes:
                n_c = int(hist_h[i][c])
                idx_c = order[i, off:off + n_c]
                off += n_c
                if c == 255:
                    continue
                do_mmd = c in self.seen and not has_unseen
                sampled_only = do_mmd and use_real and self.fused_adam
                z_cpu = torch.rand((n_c, self.noise_dim)) if (self.noise == "cpu" and ctx is None) else None
                if sampled_only and table_mode:      # indices already drawn into the step's table
                    pending += 1
                    ring_slots.append((slot, n_ring))
                    n_ring += 1
                    mmd_slots.append((slot, len(classes)))
                    slot += 1
                    continue
                ridx_cpu = torch.randint(low=0, high=n_c, size=(self.bsg,)) if do_mmd else None
                if sampled_only:
                    ring = st["ring"][st["ring_pos"]]
                    st["ring_pos"] += 1
                    ring.copy_(ridx_cpu)
                    st["ridx"].copy_(ring, non_blocking=True)
                    check(lib().zs3_sample_rows(P(idx_c), P(st["ridx"]), ctypes.c_long(i * npix), P(st["pix_local"]),
                                                P(st["pix_global"]), I(self.bsg), stream()), "zs3_sample_rows")
                    if z_cpu is not None:
                        st["z"].copy_(z_cpu[ridx_cpu])
                    self._run_sampled_update(training)
                    ring_slots.append((slot, n_ring))     # value lands in loss_ring[n_ring] (written by the update epilogue)
                    n_ring += 1
                    mmd_slots.append((slot, len(classes)))
                    slot += 1
                    continue
                # full-class generator call (image with an unseen class, or caller-supplied optimizer)
                if ctx is not None:
                    z = ctx.view(1, -1).expand(n_c, -1).contiguous()
                else:
                    z = z_cpu.to(dev) if z_cpu is not None else ops.uniform((n_c, self.noise_dim), Fz.next_seed(), dev)
                x = ops.gather_cat(st["emb"], idx_c, self.embed_dim, z, self.noise_dim, self.embed_dim + self.noise_dim)
                fake_c, h, hd, seed = self._generator_forward(x, training)
                if do_mmd:
                    ridx = ridx_cpu.to(dev)
                    s = self.bsg
                    gen_s = ops.gather_rows(fake_c, ridx)
                    real_s = ops.gather_rows(real_rows[i], idx_c[ridx])
                    t = (2 * s + 31) // 32
                    gmat = torch.empty((2 * s, 2 * s), dtype=torch.float32, device=dev)
                    tile = torch.empty(2 * t * t, dtype=torch.float64, device=dev)
                    loss = mmd_losses[slot:slot + 1]
                    check(lib().zs3_mmd_fwd(P(gen_s), I(d), P(real_s), I(d), I(s), I(d), self._sig, I(len(self.sigma)),
                                            P(gmat), P(tile), P(loss), stream()), "zs3_mmd_fwd")
                    dgen = torch.empty_like(gen_s)
                    check(lib().zs3_mmd_bwd(P(gen_s), I(d), P(real_s), I(d), I(s), I(d), P(gmat), P(loss), P(st["one"]),
                                            P(dgen), I(d), stream()), "zs3_mmd_bwd")
                    self._eager_update_rows(x, h, hd, seed, ridx, dgen)
                    st["adam_sig"] = self._adam_signature()   # the eager optimizer step advanced the host step count
                    mmd_slots.append((slot, len(classes)))
                    slot += 1
                if not use_real:
                    ops.scatter_rows(fake_c, idx_c, fake_rows[i])
            # table mode: the sampled updates of consecutive images are replayed together (a replay costs ~100 us of launch
            # latency on top of its kernels: 48 replays per step became ~12), unless a subclass hooks in after every image
            if pending and (per_image_hook or i == b - 1):
                self._run_sampled_update(training, pending)
                pending = 0
            self._after_image(i, tgt_l[i].view(fh, fw), real_rows[i], has_unseen)
        self._join_side_work()       # (subclasses: per-image work that was queued on another stream)
        loop_end = None
        if FEATURE_ADAPT and self._prefetched is not None:
            loop_end = torch.cuda.Event(enable_timing=True)
            loop_end.record()
        _tick("loop")
        pg = None if self.group is True else self.group
        if self.group is not None:   # generator replicas -> their average (parameters only; Adam moments stay per rank)
            from .parallel import all_reduce_tensors
            gen_params = [p.data for p in self._replica_parameters()]
            self.bytes_reduced += all_reduce_tensors(gen_params, group=pg, average=True)
            Fz.invalidate_planes(*self._replica_parameters())
            self._resplit()
        # ---- classifier update on the stitched features (only pred_conv receives gradients)
        self.optimizer.zero_grad()
        out = model.forward_class_prediction(ops.nchw(fake), image.shape[2:])
        closs = self.criterion(out, target)
        closs.backward()
        self._extra_classifier_terms()
        if self.group is not None:
            grads = [p.grad for g_ in self.optimizer.param_groups for p in g_["params"] if p.grad is not None]
            self.bytes_reduced += all_reduce_tensors(grads, group=pg, average=self.grad_reduce == "mean")
            # one rank's f16x3 overflow (the backbone's train-mode BatchNorm forward raises the flag) reaches
            # every rank inside the summed gradients: all ranks skip this classifier step and fall back together
            from .parallel import exchange_range_flag
            exchange_range_flag(dev, pg)
        self.optimizer.step()
        _tick("classifier")
        if ring_slots:
            if n_ring > st["loss_ring"].numel():
                raise RuntimeError("more generator updates in one step than the loss ring holds")
            dst = torch.tensor([a_ for a_, _ in ring_slots], dtype=torch.int64, device=dev)
            mmd_losses.index_copy_(0, dst, st["loss_ring"][:n_ring])
        # the single read-back of the step; the f16x3 range flag travels with it (this step never goes through LossLog: without
        # this look a raised flag would make the fused SGD skip every later classifier step in silence -- ADVICE r5)
        tail = [closs.detach().reshape(1)]
        if ops._range_flags:
            tail.append(ops.range_flag(dev).float())
        step_end = None
        if loop_end is not None:
            step_end = torch.cuda.Event(enable_timing=True)
            step_end.record()
        vals = torch.cat((mmd_losses, *tail)).cpu()
        _tick("readback")
        if step_end is not None:
            self._steer_feature_shaping(loop_end, step_end)
        if len(tail) == 2:
            if float(vals[-1]) != 0.0:
                Fz.check_forward_range(flag_value=1)     # lowers the flag, warns, bf16x3 forward products from here on
            vals = vals[:-1]
        g_batch = sum(float(vals[sl]) / nuniq for sl, nuniq in mmd_slots)
        self.last_updates = len(mmd_slots)
        return g_batch, float(vals[-1]), out


_TICKS = [] if os.environ.get("ZS3_GMMN_TICKS") else None      # host time stamps inside GMMNStep.__call__ (tools/probe)


def _tick(tag, stream_obj=None):
    if _TICKS is not None:
        import time
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(stream_obj) if stream_obj is not None else ev.record()
        _TICKS.append((tag, time.perf_counter(), ev))


class GMMNTrainer:
    """Trainer with the attribute/driver surface of zs3/train_pascal_GMMN.py:21-311 (`training(epoch, args)`),
    built from injected pieces (model, generator, optimizers, criteria, loaders, scheduler, writer...)."""

    def __init__(self, args, model, generator, optimizer, optimizer_generator, criterion, train_loader, scheduler,
                 writer=None, noise="device"):
        self.args, self.model, self.generator = args, model, generator
        self.optimizer, self.optimizer_generator = optimizer, optimizer_generator
        self.criterion, self.train_loader, self.scheduler, self.writer = criterion, train_loader, scheduler, writer
        self.best_pred = 0.0
        self.step_fn = GMMNStep(model, generator, optimizer, optimizer_generator, criterion,
                                seen=args.seen_classes_idx_metric, unseen=args.unseen_classes_idx_metric,
                                noise_dim=args.noise_dim, embed_dim=args.embed_dim, feature_dim=args.feature_dim,
                                batch_size_generator=args.batch_size_generator,
                                real_seen_features=args.real_seen_features, noise=noise)

    def training(self, epoch, args=None):
        train_loss = 0.0
        self.model.train()
        num_img_tr = len(self.train_loader)
        # one batch of lookahead: the next batch's image goes to the device early so that its (frozen-backbone) feature
        # pass overlaps this batch's generator loop
        it = iter(self.train_loader)
        sample, i, staged = next(it, None), 0, None
        while sample is not None:
            following = next(it, None)
            if len(sample["image"]) <= 1:
                sample, i = following, i + 1
                continue
            image = staged if staged is not None else sample["image"].cuda()
            target, embedding = sample["label"].cuda(), sample["label_emb"].cuda()
            staged = following["image"].cuda() if (following is not None and len(following["image"]) > 1) else None
            self.scheduler(self.optimizer, i, epoch, self.best_pred)
            g_loss, c_loss, _ = self.step_fn(image, target, embedding, next_image=staged)
            sample, i_done, i = following, i, i + 1
            train_loss += c_loss
            if self.writer is not None:
                self.writer.add_scalar("train/total_loss_iter", c_loss, i_done + num_img_tr * epoch)
                self.writer.add_scalar("train/generator_loss", g_loss, i_done + num_img_tr * epoch)
        if self.writer is not None:
            self.writer.add_scalar("train/total_loss_epoch", train_loss, epoch)
        return train_loss
