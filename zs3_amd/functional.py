"""Differentiable ops of the hot path: torch.autograd.Function wrappers around the HIP kernels.

All activations flowing between these functions are NHWC tensors [N, H, W, C] (fp32, channel
contiguous).  Each Function is one *fused layer* (conv + BatchNorm + residual + ReLU, ...), so a
training step is ~150 autograd nodes instead of ~700 ATen ops.  Nothing here falls back to torch
compute: tensors must live on the GPU and libzs3hip.so must be present.
"""
import ctypes
import os
import random
import weakref

import torch

from . import ops
from ._lib import require_gpu

ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2

# ---------------------------------------------------------------------------------- side stream for weight gradients
# A layer's wgrad only feeds the optimizer, while its dgrad feeds the next layer of the backward chain.  Many layers
# of this network launch fewer workgroups than the 256 CUs can hold, so the wgrad kernels run on a second HIP stream
# next to the dgrad / BatchNorm chain; the main stream joins it once, at the end of backward.
WGRAD_SIDE_STREAM = True
# The side stream of a layer's weight gradient waits for the main stream at the point where dy exists (behind bn_act_bwd), not
# behind the layer's data-gradient launch: dgrad and wgrad of one layer both only READ dy, so they may run side by side.
EARLY_WGRAD_FORK = False
# Side streams inside a hipGraph capture (GraphedStep): the side stream joins the capture through its wait on the capturing stream
# and is joined back by the end-of-backward callback, so the captured graph keeps the dgrad / wgrad overlap of the eager step.
CAPTURE_SIDE_STREAMS = False   # (probe: tools/probe/graph_probe.py -- 36.2 ms with, 36.8 without for fwd + bwd in the 2-byte mode: hipGraph replay does not overlap the branches; off)
FUSE_BN_BWD_STATS = True   # BN-backward sums produced by the sole consumer's dgrad epilogue (BnLink)
DROPOUT_FUSED = True   # nn.Dropout behind conv+BN+ReLU inside the BN-apply pass
# BN-apply + ReLU in the sole consumer's operand path (conv_bn_act: next_conv).  Same-box A/B, ms per step: off 46.64 / 46.87, on
# 46.51 / 46.47, on for 3x3 consumers only 46.68 / 46.67 (tools/probe/ab_env.sh; per layer: tools/probe/defer_bench.py)
DEFER_BN_APPLY = os.environ.get("ZS3_DEFER_BN", "1") == "1"
LAZY_SKIP_GRAD = True      # identity blocks: the skip gradient dA*mask is applied by conv1's dgrad epilogue, never stored
_lazy_skip = {}            # data_ptr of a block-output gradient -> (tensor, sign bits) it still has to be masked with
# The layers' weight gradients are independent of each other: with WGRAD_STREAMS > 1 they go round-robin over a small pool
# of side streams, each launch sized (split-K) for its share of the chip (ZS3_WGRAD_CUS, read by zs3_conv_wgrad_plan) --
# fewer K splits per layer = less split-K slab traffic, the same number of workgroups in flight.
# (round 6: the two streams of rounds 3-5 had been sharing ONE hardware queue all along -- see warm_streams -- so they were one
# stream in effect; on two queues of their own they starve the main chain, 52-60 ms.  One stream: 43.9-44.1 against 43.9-44.2.)
WGRAD_STREAMS = max(1, int(os.environ.get("ZS3_WGRAD_STREAMS", "1")))
_side = {}
_side_next = {}
_join_armed = [False]


# torch.cuda.current_stream() / `with torch.cuda.stream(s)` / Stream.wait_stream() spend 9-15 us each in device-index bookkeeping
# and fresh Event objects; a weight-gradient launch used all three (4 ms of host time per training step, measured with
# tools/probe/host_profile.py -- the 2-byte mode's step is host-bound).  The same calls on the raw bindings:
_raw_device = getattr(torch._C, "_cuda_getDevice", None)
_raw_current = getattr(torch._C, "_cuda_getCurrentStream", None)
_raw_set = getattr(torch._C, "_cuda_setStream", None)
_stream_objs = {}      # (stream_id, device_index, device_type) -> torch.cuda.Stream


def _current_stream():
    if _raw_current is None or _raw_device is None:
        return torch.cuda.current_stream()
    sd = _raw_current(_raw_device())
    s = _stream_objs.get(sd)
    if s is None:
        s = _stream_objs[sd] = torch.cuda.Stream(stream_id=sd[0], device_index=sd[1], device_type=sd[2])
    return s


def _set_stream(s):
    if _raw_set is None:
        torch.cuda.set_stream(s)
    else:
        _raw_set(stream_id=s.stream_id, device_index=s.device_index, device_type=s.device_type)


def _wait_for(waiter, producer):
    """waiter.wait_stream(producer) as ONE call of the library (zs3_stream_wait: event record + stream wait on one reusable event per
    waiting stream).  Under the C ABI since round 6 so that a recorded plan (zs3_amd/plan.py) carries the step's cross-stream
    dependencies; it is also the cheapest spelling (torch's Event.record + Stream.wait_event: two calls of 5-10 us)."""
    ops.check(ops.lib().zs3_stream_wait(waiter.cuda_stream, producer.cuda_stream), "zs3_stream_wait")


# A recorded plan replays the step's launches with the addresses of the recording step, without the caching allocator in between.
# While a plan records (zs3_amd/plan.py sets this), tensors that a SIDE stream reads are therefore not handed to the allocator's
# record_stream bookkeeping -- it makes a block reusable when the HOST has seen the side stream's event complete, a fact about the
# recording run's timing that no replay repeats -- but kept alive here until the end-of-backward join, after which the main stream
# (where they were allocated) is ordered behind every side-stream reader.
PLAN_RECORDING = False
ASPP_LANES_LAST = True    # the lane branches are the LAST nodes created in ASPP's forward = the first to run in its backward
ASPP_CONCURRENT = os.environ.get("ZS3_ASPP_LANES", "1") == "1"   # ASPP's two heavy atrous branches on lanes (modeling/aspp.py)
_plan_keep = []
PLAN_EPOCH = [0]     # bumped whenever buffers a plan may have recorded are dropped (weight planes, mode switches): plans re-record


def wgrad_streams(device):
    key = device.index
    if key not in _side:
        _side[key] = [torch.cuda.Stream(device=device) for _ in range(WGRAD_STREAMS)]
        _side_next[key] = 0
    return _side[key]


def feature_stream(device):
    """the stream of the GMMN step's prefetched feature pass (gmmn_trainer.GMMNStep.prefetch): one per device, created here so that
    warm_streams can give it a hardware queue of its own"""
    key = device.index
    if key not in _feature:
        _feature[key] = torch.cuda.Stream(device=device)
    return _feature[key]


_feature = {}


def wgrad_stream(device):
    """the side stream the next weight-gradient launch goes to (round-robin over the pool)"""
    pool = wgrad_streams(device)
    i = _side_next[device.index]
    _side_next[device.index] = (i + 1) % len(pool)
    return pool[i]


def warm_streams(device=None):
    """Touch the streams of the step in a fixed order -- current (main) stream, the two ASPP lanes, the weight-gradient stream(s) --
    so that each gets its hardware queue NOW.

    Why: the HIP runtime multiplexes streams onto at most GPU_MAX_HW_QUEUES (default 4) hardware queues, handed out in order of first
    use, later streams sharing the least-used one; kernels of streams that share a queue run in submission order.  WHICH streams
    share decides the step (measured with `rocprofv3 --kernel-trace`, tools/probe/db_queues.py, B = 16 at 513^2): main, lane, lane,
    weight gradient on four queues of their own: 42.4-43.8 ms; the weight-gradient stream on the MAIN stream's queue (what the
    one-rank N > 1 selftest got through round 6, its streams being created after the process group's): 48.5-49.4 ms, every data
    gradient followed by a 40-55 us hole on the main stream; two weight-gradient streams on two queues of their own: 52-60 ms (their
    launches cover all 256 CUs and starve the main chain); raising GPU_MAX_HW_QUEUES beside RCCL's own streams: 58-86 ms.
    Call it before torch.distributed.init_process_group / the first communicator (bench.py does; a model does at its first forward,
    which is early enough in a single-process run).  Idempotent and cheap (four 4-byte fills)."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if device.index in _warmed:
        return
    _warmed.add(device.index)
    scratch = torch.empty(64, dtype=torch.float32, device=device)
    # (the GMMN step's feature stream takes the fourth queue; the weight-gradient stream, the next one created, then shares it --
    # the runtime hands a fifth stream the LAST of the least-used queues -- which is idle in the supervised step and carries one
    # small launch per step in the GMMN step)
    order = [torch.cuda.current_stream(device)] + list(lane_streams(device)) + [feature_stream(device)] + list(wgrad_streams(device))
    for st in order:
        ops.check(ops.lib().zs3_fill_zero(scratch.data_ptr(), ctypes.c_long(256), st.cuda_stream), "zs3_fill_zero")
        st.synchronize()


_warmed = set()


def join_wgrad_stream():
    """Make the current stream wait for every wgrad launched on the side streams (called at the end of backward)."""
    _join_armed[0] = False
    _wgrad_side_of.clear()
    for pool in _side.values():
        for st in pool:
            _wait_for(torch.cuda.current_stream(st.device), st)



# ---------------------------------------------------------------------------------- lanes: independent layers side by side
# ASPP's branches are independent and each launches fewer workgroups than the chip has CUs (138 tiles on 256 CUs at 33 x 33): run
# side by side they are worth 3 ms of a 44 ms step (same-box A/B, round 6).  Through round 5 the module ran them under
# `torch.cuda.stream(...)`: autograd then runs each branch's backward on the stream of its forward and orders the streams with
# events of its own -- dependencies no recorded plan carries.  A LANE is the same overlap kept inside this library: a fused layer
# whose cfg names a lane switches the current stream INSIDE its Function.forward / backward (after autograd has noted the node's
# stream, before it looks again), the lane first waits for the main stream (its operand is ready there), and the main stream
# waits for the lanes where the results meet: lanes_join() in the module's forward, _Fork.backward / the end-of-backward
# callback in the backward pass.  Every dependency is a zs3_stream_wait call: recorded and replayed like a launch.
# Allocator: tensors a layer allocates inside its lane belong to the lane's pool and are reused in the lane's own stream order;
# what crosses (the branch input and output-gradient slices from main, the branch's input gradient to main) is ordered by the
# fork wait at the lane's next entry and by the joins.
_lanes = {}
_lanes_open = {}          # device index -> lanes that have work the main stream has not waited for yet
_lanes_armed = [False]


def lane_streams(device, n=2):
    key = device.index
    if key not in _lanes or len(_lanes[key]) < n:
        _lanes[key] = [torch.cuda.Stream(device=device) for _ in range(n)]
    return _lanes[key]


def lanes_fork(*lanes):
    """the lanes wait for what the current (main) stream holds NOW: layers that enter them with `lane_forked` set start from this
    point, whatever the main stream is given in between (ASPP: the main stream's own branches are enqueued before the lanes')"""
    main = _current_stream()
    for lane in lanes:
        if lane is not None:
            _wait_for(lane, main)


def _enter_lane(lane, wait=True):
    main = _current_stream()
    if wait:
        _wait_for(lane, main)
    _set_stream(lane)
    _lanes_open.setdefault(lane.device_index, set()).add(lane)
    return main


def lanes_join():
    """the current (main) stream waits for every lane that has taken work since the last join"""
    if not _lanes_open:
        return
    main = _current_stream()
    for lane in _lanes_open.pop(main.device_index, ()):
        _wait_for(main, lane)


def _lanes_join_at_end_of_backward():
    _lanes_armed[0] = False
    for idx in list(_lanes_open):
        for lane in _lanes_open.pop(idx, ()):
            _wait_for(torch.cuda.current_stream(lane.device), lane)


# ---------------------------------------------------------------------------------- weight-plane cache
_planes = {}


def forward_is_f16x3(prec, bn):
    """Does this fused layer's forward convolution run f16x3 (fp16 hi/lo operands, prec 4)?  Under the default three-product
    arithmetic, when the layer normalises with BATCH statistics: then its input is the output of a batch-normalised layer (or the
    image) -- O(1) by construction, inside fp16's exponent range -- and this is the mode whose conditioning needs the 2^-22
    products.  Eval-mode / frozen BatchNorm and layers without one keep bf16x3 forward products: an
    untrained network in eval() mode lets activations grow to 1e4 and beyond (running statistics 0 / 1 normalise nothing), which
    fp16 cannot hold, and eval mode is well-conditioned anyway (logits 3e-5 from the reference)."""
    return bool(ops.fwd_f16() and prec in (None, 3, 4) and bn is not None and bn["training"])


def check_forward_range(device=None, flag_value=None):
    """Range guard of the f16x3 forward.  fp16 hi/lo operands hold |x| <= 65504 (weights: |w| < 1023); the
    rule that selects them (batch-statistics layers: inputs are batch-normalised activations or the image) keeps real networks far
    inside that, but nothing in the data enforces it.  A layer whose operands overflowed produces non-finite batch sums, its
    BatchNorm finalize raises the device's sticky flag (ops.range_flag) and skips its running-statistics update -- as does every
    finalize launch that finds the flag already up (the layers downstream of the overflow see a dead layer's zeros, the steps queued
    before the host looks see garbage) -- and the fused SGD skips its update while the flag is up: the affected steps are lost like a
    loss scaler's overflow steps; parameters, momentum and (up to the raising layer's own racing channel groups) running
    statistics keep their last good values.
    This function is the host's half: called by LossLog / the trainers where they read the loss anyway (one iteration late,
    `flag_value` = the copied flag) or directly (device given: a synchronising read), it lowers the flag and switches the
    forward arithmetic to bf16x3 (fp32's exponent range) for every launch from then on.  Returns True when it fell back."""
    import warnings
    if flag_value is None:
        if device is None or (device.type, device.index) not in ops._range_flags:
            return False
        flag_value = int(ops.range_flag(device).item())
    if not flag_value:
        return False
    for flag in ops._range_flags.values():
        flag.zero_()          # (stream order: the steps queued before this line still see it raised and skip their update)
    if ops.FWD_F16:     # (a later call that finds an older step's copy of the raised flag only lowers it again: one warning per fallback)
        ops.FWD_F16 = False
        PLAN_EPOCH[0] += 1
        _planes.clear()
        _refresh_tables.clear()
        _defer_choice.clear()
        _in_affine_choice.clear()
        ops._TILE_CHOICE.clear()
        warnings.warn("zs3_amd: a forward convolution multiplied operands beyond fp16's range (non-finite batch statistics); the "
                      "affected steps were skipped and forward products fall back to the bf16 split (ZS3_FWD_F16=0) from here on",
                      RuntimeWarning, stacklevel=2)
    return True


def weight_planes(w, need_t=True, f16=False):
    """hi/lo planes of a parameter, recomputed only when the parameter changed (optimizer step,
    load_state_dict): keyed on (storage pointer, autograd version counter).  f16: the forward plane as fp16 hi/lo (the layer's
    forward launches run prec 4, `forward_is_f16x3`); the data-gradient plane is bf16 hi/lo either way."""
    key = id(w)
    fmt = 1 if f16 else 0
    ver = (w.data_ptr(), w._version, tuple(w.shape), need_t, fmt)
    hit = _planes.get(key)
    if hit is not None and hit[0] == ver and hit[2]() is w:
        return hit[1]
    if hit is not None and hit[0][:3] == ver[:3] and hit[0][3] and not need_t and hit[0][4] == fmt and hit[2]() is w:
        return hit[1]
    wp = ops.prep_weight(w, need_t=need_t, f16_forward=bool(fmt))
    _planes[key] = (ver, wp, weakref.ref(w, lambda _r, k=key: _planes.pop(k, None)))
    return wp


def invalidate_planes(*params):
    """Drop cached planes of parameters that were updated through raw pointers (our fused optimizers)."""
    for w in params:
        if _planes.pop(id(w), None) is not None:
            PLAN_EPOCH[0] += 1       # a recorded plan may hold these buffers' addresses


# ---------------------------------------------------------------------------------- gradient buckets (parallel.GradSync)
_grad_buffers = {}


def register_grad_buffer(param, flat_view):
    """flat_view: 1-D fp32 view (param.numel() elements) of a GradSync bucket, or None to forget the parameter."""
    if flat_view is None:
        _grad_buffers.pop(id(param), None)
    else:
        _grad_buffers[id(param)] = (weakref.ref(param), flat_view)


_handed = set()          # parameters that already received their bucket slice in the running backward pass
_handed_armed = [False]
_wgrad_side_of = {}      # id(weight) -> side stream of its first weight-gradient launch in the running backward pass


def _end_of_backward():
    _handed.clear()
    _handed_armed[0] = False


_graph_task_id = getattr(torch._C, "_current_graph_task_id", None)


def _reset_backward_state():
    """The per-backward bookkeeping is normally cleared by engine callbacks at the end of the pass; those do not run when backward
    raises (an out-of-memory error the caller catches and retries).  A fused layer's FORWARD outside any autograd pass means no
    backward of ours is in flight, so it clears whatever a dead pass left behind -- otherwise grad_buffer() would answer None for
    every registered parameter from then on and GradSync would silently fall back to its copy path.  A forward that runs INSIDE a
    live backward (activation checkpointing / recomputation, double backward) must leave the bookkeeping alone: clearing it
    mid-pass would hand a shared weight's bucket slice out twice and drop the stream waits of its two launches."""
    if _graph_task_id is not None and _graph_task_id() != -1:
        return
    if _handed_armed[0] or _handed:
        _end_of_backward()
    if _join_armed[0]:
        join_wgrad_stream()
    if _wgrad_side_of:
        _wgrad_side_of.clear()
    if _lanes_armed[0]:
        _lanes_join_at_end_of_backward()


def grad_buffer(param):
    """The bucket slice a fresh gradient of `param` should be written into, or None.  The slice is handed out only when the
    parameter has no .grad yet (otherwise autograd accumulates into the existing one and the slice may already hold it) and
    only ONCE per backward pass: a weight used by two layers gets two wgrad launches (possibly on two streams of the wgrad
    pool), AccumulateGrad sums the two contributions after both exist, and two launches writing one slice would leave
    2 x the last one there.  The second request gets None -> a private buffer.  (Gradients returned by torch.autograd.grad for a
    registered parameter still alias its slice: they are valid until the next backward writes it.)"""
    ent = _grad_buffers.get(id(param))
    if ent is None or ent[0]() is not param or not param.is_leaf or param.grad is not None or id(param) in _handed:
        return None
    _handed.add(id(param))
    if not _handed_armed[0]:
        try:     # only legal while the autograd engine is running a backward pass -- which is where wgrad launches come from
            torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward)
            _handed_armed[0] = True
        except RuntimeError:
            _handed.discard(id(param))
    return ent[1]


_refresh_tables = {}


def refresh_planes(*params):
    """After an optimizer step through raw pointers: recompute the cached planes of every parameter that has any, in ONE
    launch (zs3_prep_weight_multi) into the existing plane buffers, instead of one zs3_prep_weight per layer at its next
    forward.  Parameters whose memory is not the [Cout][KH][KW][Cin] storage the planes were built from are invalidated."""
    if ops.PREC_DEFAULT == 0:   # exact-fp32 test mode: the one-launch refresh writes bf16 planes
        invalidate_planes(*params)
        return
    todo = []
    for w in params:
        hit = _planes.get(id(w))
        if hit is None:
            continue
        ver, wp, ref = hit
        w4 = w if w.dim() == 4 else w[:, :, None, None]
        if ref() is not w or ver[0] != w.data_ptr() or ver[2] != tuple(w.shape) or not ver[3] or \
                not w4.permute(0, 2, 3, 1).is_contiguous():
            _planes.pop(id(w), None)
            continue
        todo.append((w, wp, ref))
    if not todo:
        return
    dev = todo[0][0].device
    key = (dev.index,) + tuple((w.data_ptr(), wp.f_pk.data_ptr(), wp.t_pk.data_ptr(), wp.f_fmt) for w, wp, _ in todo)
    tab = _refresh_tables.get(dev.index)
    if tab is None or tab[0] != key:
        recs, bmap = [], []
        for e, (w, wp, _) in enumerate(todo):
            taps = wp.kh * wp.kw
            recs.append((w.data_ptr(), wp.f_pk.data_ptr(), wp.t_pk.data_ptr(), wp.cout, taps | ((1 << 32) if wp.f_fmt == 1 else 0),
                         wp.cin, wp.cin_pad, wp.cout_pad))   # bit 32 of the taps word: forward plane as fp16 hi/lo
            bmap.extend((e, c) for c in range(ops.lib().zs3_prep_chunks(ops.I(wp.cout_pad), ops.I(taps), ops.I(wp.cin_pad))))
        table = torch.tensor(recs, dtype=torch.int64).to(dev)
        blockmap = torch.tensor(bmap, dtype=torch.int32).to(dev)
        tab = (key, table, blockmap, len(bmap))
        _refresh_tables[dev.index] = tab
    ops.check(ops.lib().zs3_prep_weight_multi(ops.P(tab[1]), ops.P(tab[2]), ops.I(tab[3]), ops.stream()),
              "zs3_prep_weight_multi")
    for w, wp, ref in todo:
        _planes[id(w)] = ((w.data_ptr(), w._version, tuple(w.shape), True, 1 if wp.f_fmt == 1 else 0), wp, ref)


# ---------------------------------------------------------------------------------- RNG for dropout
_rng = None


def manual_seed(seed):
    global _rng
    _rng = random.Random(int(seed))


def next_seed():
    global _rng
    if _rng is None:
        _rng = random.Random(torch.initial_seed())
    return _rng.getrandbits(63)


def _pad_channels(t, mult):
    """NHWC/rows tensor -> same logical tensor whose row stride is a multiple of `mult` floats with zero
    padding behind the last channel (what the dgrad/wgrad kernels need to read whole chunks)."""
    c = t.shape[-1]
    ld = ops._rows(t)[2] if t.stride(-1) == 1 else -1
    if ld > 0 and ld % mult == 0 and (c % mult == 0):
        return t
    cp = (c + mult - 1) // mult * mult
    if t.is_cuda and t.dtype in (torch.float32, torch.bfloat16) and t.stride(-1) == 1:
        try:
            return ops.pad_rows(t, cp)          # one launch of the library (a recorded plan carries it)
        except AssertionError:
            pass                                # rows without a uniform stride: the general copy below
    buf = torch.zeros(t.shape[:-1] + (cp,), dtype=t.dtype, device=t.device)
    buf[..., :c].copy_(t)
    return buf[..., :c]


def _dense_rows(t):
    if t.stride(-1) != 1:
        return t.contiguous()
    try:
        ops._rows(t)
        return t
    except AssertionError:
        return t.contiguous()


class BnLink:
    """Hand-off between a fused conv+BN(+ReLU) layer and the *only* consumer of its output.  The consumer's dgrad epilogue
    already holds the finished gradient dA of this layer in registers, so it also produces this layer's BN-backward sums
    (sum dz, sum dz*xhat; zs3_conv_igemm_bnstats) and the separate statistics pass over dA is skipped.  The model code
    promises the sole-consumer property (`input_has_one_consumer=True`); the producer double-checks that the gradient it
    receives is the very buffer the sums were computed for."""
    __slots__ = ("y", "mean", "istd", "msc", "msh", "mbits", "partial", "for_ptr")

    def __init__(self):
        self.y = self.mean = self.istd = self.msc = self.msh = self.mbits = self.partial = None
        self.for_ptr = 0


def _mask_bits_for(y, act, residual, need_grad):
    """Residual layers: the ReLU mask cannot be recomputed from y alone, and re-reading the output `a` twice in the backward
    pass costs 8 bytes per element; affine_act writes the sign bits (1/16 of that) instead."""
    if not need_grad or act == ACT_NONE or residual is None:
        return None
    m, c, _ = ops._rows(y)
    return torch.empty(m * (c // 4), dtype=torch.uint8, device=y.device)


# ---------------------------------------------------------------------------------- conv + BN + act
class _ConvBnAct(torch.autograd.Function):
    """y = conv(x, w) ; a = act(bn(y) + bias + residual).  One node for the whole fused layer.

    reference: Bottleneck.forward (resnet.py:33-53), _ASPPModule.forward (aspp.py:25-29), ASPP.forward
    (aspp.py:111-114), Decoder (decoder.py:30-32, last_conv), pred_conv (decoder.py:26), nn.Linear (gmmn.py)."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, bias, residual, cfg):
        lane = cfg.get("lane")
        if lane is None:
            return _ConvBnAct._forward(ctx, x, weight, gamma, beta, bias, residual, cfg)
        main = _enter_lane(lane, wait=not cfg.get("lane_forked"))
        try:
            return _ConvBnAct._forward(ctx, x, weight, gamma, beta, bias, residual, cfg)
        finally:
            _set_stream(main)

    @staticmethod
    def backward(ctx, dA, dskip=None):
        lane = ctx.cfg.get("lane")
        if lane is None:
            return _ConvBnAct._backward(ctx, dA, dskip)
        main = _enter_lane(lane)
        if not _lanes_armed[0]:     # whoever consumes this layer's gradients on the main stream -- _Fork.backward, else the optimizer
            _lanes_armed[0] = True
            torch.autograd.Variable._execution_engine.queue_callback(_lanes_join_at_end_of_backward)
        ctx.outer_stream = main      # where autograd runs the consumers of this node's results inside this pass (AccumulateGrad)
        # parameter gradients that are READ inside this pass (added into an existing .grad by a second backward before the optimizer
        # step, or flowing on through a non-leaf parameter): autograd does that on the main stream right behind this node
        read_now = any(t is not None and t.requires_grad and (not t.is_leaf or t.grad is not None) for t in ctx.saved_tensors[1:3])
        try:
            return _ConvBnAct._backward(ctx, dA, dskip)
        finally:
            _set_stream(main)
            if read_now:
                _wait_for(main, lane)

    @staticmethod
    def _forward(ctx, x, weight, gamma, beta, bias, residual, cfg):
        require_gpu(x, weight)
        x = _dense_rows(x)
        need_grad = cfg.get("need_grad", True)  # grad mode is always off inside Function.forward: decided by the caller
        bn = cfg.get("bn")
        wp = weight_planes(weight, need_t=True, f16=forward_is_f16x3(cfg.get("prec"), bn))
        stride, pad, dil, act = cfg["stride"], cfg["pad"], cfg["dil"], cfg["act"]
        geom = cfg.get("geom")  # explicit (n,h,w,ldx,...) for the stem's overlapping-window view
        out = cfg.get("out")
        leak = cfg.get("leak", 0.2)
        prec = 4 if wp.f_fmt == 1 else cfg.get("prec")    # forward products: f16x3 where the plane was prepared for it
        drop = cfg.get("drop")   # (p, seed): nn.Dropout behind this layer's activation, fused into affine_act / bn_act_bwd
        y = a = st = mbits = None
        in_aff = cfg.get("in_affine")   # x is the RAW conv output of the producing layer: its BN-apply + ReLU runs in our producers
        if in_aff is not None and not _applies_in_affine(x, wp, stride, pad, dil, prec, bn, residual, need_grad):
            # the producer handed its raw output over expecting a training-mode consumer on the producer-converting kernels; this
            # consumer runs otherwise (eval-mode BatchNorm with a residual epilogue, another precision, ...): apply the BatchNorm +
            # ReLU here as the stored pass the producer skipped, and carry on as if nothing had been deferred
            x = ops.affine_act(x, in_aff[0], in_aff[1], act=ACT_RELU)
            in_aff = None
        # element type of this layer's output: the caller's (the class scores stay fp32), else the input's own -- the stem (fp32
        # image in) is where the storage type ops.ACT_DTYPE enters the network, everything behind it inherits it, and the
        # generators (nn.Linear / GraphConvolution on fp32 rows, whose kernels -- csrc/gmmn.hip, gcn.hip, the MMD loss -- and
        # whose products-used-as-weights are fp32) stay fp32 in either mode
        odt = cfg.get("out_dtype") or (ops.ACT_DTYPE if geom is not None else x.dtype)
        conv = (lambda **k: ops.conv2d_fwd(x, wp, stride, pad, dil, prec=prec, in_affine=in_aff, out_dtype=odt, **k)) if geom is None else (
            lambda **k: ops.conv_igemm(x, wp.f_pk, prec=prec, out_dtype=odt, **geom, **k))
        defer = bool(cfg.get("defer_out"))
        if bn is not None and bn["training"]:
            y, part = conv(want_stats=True)
            count = y.shape[0] * y.shape[1] * y.shape[2]
            if count <= 1:
                raise ValueError("Expected more than 1 value per channel when training, got input size "
                                 f"{(y.shape[0], y.shape[3], y.shape[1], y.shape[2])}")
            if bn.get("sync") is not None:
                from .parallel import combine_bn_partials
                part, count = combine_bn_partials(part, count, None if bn["sync"] is True else bn["sync"])
            st = ops.bn_fwd_finalize(part, count, gamma, beta, bn["eps"], bn["momentum"], bn["running_mean"],
                                     bn["running_var"], bn.get("nbt"),
                                     range_flag=ops.range_flag(x.device) if wp.f_fmt == 1 else None)   # f16x3 launch: range guard
            if defer:
                # the one consumer of this layer applies scale / shift / ReLU in its own operand path (forward and weight
                # gradient): no BN-apply pass, no activation tensor -- the raw conv output is what travels on
                a = y
            else:
                mbits = _mask_bits_for(y, act, residual, need_grad)
                a = ops.affine_act(y, st[2], st[3], res=residual, out=out, act=act, leak=leak, mask_out=mbits, drop=drop)
        elif bn is not None:
            st = ops.bn_eval_affine(gamma, beta, bn["running_mean"], bn["running_var"], bn["eps"])
            if need_grad:
                y, _ = conv()
                mbits = _mask_bits_for(y, act, residual, need_grad)
                a = ops.affine_act(y, st[2], st[3], res=residual, out=out, act=act, leak=leak, mask_out=mbits, drop=drop)
            else:
                a, _ = conv(scale=st[2], shift=st[3], res=residual, act=act, leak=leak, out=out)
                if drop is not None:   # conv epilogue path (no gradient needed): the dropout stays a pass of its own
                    a = ops.dropout(a, drop[0], drop[1], out=a)
        else:
            a, _ = conv(shift=bias, res=residual, act=act, leak=leak, out=out)
        ctx.cfg = cfg
        ctx.has_bn = bn is not None
        ctx.bn_training = bool(bn is not None and bn["training"])
        ctx.has_bias = bias is not None
        ctx.has_res = residual is not None
        # identity block whose skip tensor is conv1's pass-through alias: hand the *unmasked* output gradient back as the
        # skip gradient and let conv1's dgrad epilogue apply the ReLU mask (saves writing and re-reading dz)
        ctx.lazy_dres = bool(LAZY_SKIP_GRAD and residual is not None and mbits is not None and act == ACT_RELU and
                             bn is not None and out is None and getattr(residual, "_zs3_skip_alias", False))
        ctx.x_shape = tuple(x.shape)
        # ReLU mask in backward: recomputed from y (y*scale + shift > 0) when there is no residual, so `a` is not re-read
        ctx.mask_from_y = bool(bn is not None and y is not None and act == ACT_RELU and residual is None)
        keep_a = act != ACT_NONE and not ctx.mask_from_y and mbits is None
        ctx.save_for_backward(x, weight, gamma, y, a if keep_a else None, st, mbits)
        ctx.x_affine = in_aff
        if defer:
            cfg["deferred"] = (st[2], st[3])
        ctx.pass_through = bool(cfg.get("pass_through"))
        link = cfg.get("out_link")
        # (also under SyncBN since round 6: the consumer's epilogue sums dz, dz * xhat of THIS rank's rows with the global mean / invstd
        # this layer's finalize produced; the backward all-reduces them like the sums of the separate pass -- 113 full reads of the
        # gradient per step that the N > 1 path used to make and the one-GPU path did not)
        if link is not None and ctx.bn_training and need_grad and act in (ACT_NONE, ACT_RELU):
            link.y, link.mean, link.istd, link.mbits = y, st[0], st[1], mbits
            if ctx.mask_from_y:
                link.msc, link.msh = st[2], st[3]
        else:
            cfg["out_link"] = None
        if ctx.pass_through:
            return a, x.view(x.shape)   # the block input again, as the skip connection: its gradient comes back to us
        return a

    @staticmethod
    def _backward(ctx, dA, dskip=None):
        x, weight, gamma, y, a, st, mbits = ctx.saved_tensors
        cfg = ctx.cfg
        msc, msh = (st[2], st[3]) if ctx.mask_from_y else (None, None)
        stride, pad, dil, act = cfg["stride"], cfg["pad"], cfg["dil"], cfg["act"]
        leak, prec, geom = cfg.get("leak", 0.2), cfg.get("prec"), cfg.get("geom")
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dA = _dense_rows(dA)
        wp = weight_planes(weight, need_t=True, f16=forward_is_f16x3(prec, cfg.get("bn")))   # (the forward's planes: no re-split)
        dgamma = dbeta = dbias = dres = None
        lazy = ctx.lazy_dres and ctx.has_bn and ctx.needs_input_grad[5] and ops._rows(dA)[2] == dA.shape[-1]
        if ctx.has_res and ctx.needs_input_grad[5] and not lazy:
            dres = torch.empty(dA.shape, dtype=dA.dtype, device=dA.device)
        m = dA.shape[0] * dA.shape[1] * dA.shape[2] if dA.dim() == 4 else dA.shape[0]
        if ctx.has_bn:
            link = cfg.get("out_link")
            if link is not None and link.partial is not None and link.for_ptr == dA.data_ptr() and \
                    ops._rows(dA)[2] == dA.shape[-1]:
                part = link.partial      # summed by the consumer's dgrad epilogue
            else:
                part = ops.bn_bwd_stats(dA, a, y, st[0], st[1], msc, msh, mbits, drop=cfg.get("drop"))
            if link is not None:
                link.partial = link.y = link.mbits = None
            sync = (cfg.get("bn") or {}).get("sync") if ctx.bn_training else None
            # dgamma / dbeta straight into the parameters' slices of the data-parallel gradient bucket, when there is one (GradSync):
            # the all-reduce then needs no pack / unpack copies for the 226 BatchNorm vectors either
            gout = (grad_buffer(gamma), grad_buffer(cfg.get("bn_beta"))) if gamma is not None and cfg.get("bn_beta") is not None \
                else (None, None)
            if sync is not None:
                from .parallel import combine_bn_partials_bwd
                # per-rank dgamma / dbeta (summed later by GradSync) come out of the pack launch of the exchange where the library
                # issues it; c1, c2 from the global sums and count
                gpart, gcount, dgamma, dbeta = combine_bn_partials_bwd(part, m, None if sync is True else sync, out=gout)
                if dgamma is None:
                    fin_l = ops.bn_bwd_finalize(part, m, False, out=gout)
                    dgamma, dbeta = fin_l[0], fin_l[1]
                fin_g = ops.bn_bwd_finalize(gpart, gcount, True, want_param_grads=False)
                c1, c2 = fin_g[2], fin_g[3]
            else:
                fin = ops.bn_bwd_finalize(part, m, ctx.bn_training, out=gout)
                dgamma, dbeta = fin[0], fin[1]
                c1, c2 = (fin[2], fin[3]) if ctx.bn_training else (None, None)
            dy = ops.bn_act_bwd(dA, a, y, st[0], st[1], gamma, c1, c2, dres=dres, act=act, leak=leak, mask_scale=msc,
                                mask_shift=msh, mask_bits=mbits, drop=cfg.get("drop"))
            if lazy:
                if len(_lazy_skip) > 256:
                    _lazy_skip.clear()
                _lazy_skip[dA.data_ptr()] = (dA, mbits)
                dres = dA
        else:
            cout = dA.shape[-1]
            vec_ok = cout % 4 == 0 and ops._rows(dA)[2] % 4 == 0
            if act != ACT_NONE:
                if vec_ok:
                    dz = torch.empty(dA.shape, dtype=dA.dtype, device=dA.device)
                    ops.bn_act_bwd(dA, a, None, None, None, None, None, None, dres=dz, act=act, leak=leak, want_dy=False)
                else:
                    dz = dA * _act_grad(a, act, leak)
            else:
                dz = dA
            if dres is not None:
                dres = dz
            want_bias = ctx.has_bias and ctx.needs_input_grad[4]
            dy = _pad_channels(dz, 8) if (need_x or need_w or (want_bias and not vec_ok)) else dz
            if want_bias:
                # column sums in two launches of the library: per-chunk partial sums, then the chunks (rows of [chunks][2][C], every
                # second one) added in order -- no tensor-library reduction on the path (a recorded plan replays both)
                bout = grad_buffer(cfg.get("bias_param")) if cfg.get("bias_param") is not None else None
                if vec_ok:
                    dbias = _sum_partials(ops.colstats(dz), cout, bout)
                else:  # odd channel count (21 classes): reduce the zero-padded 8-aligned copy with the HIP kernel
                    full = dy.as_strided(dy.shape[:-1] + (ops._rows(dy)[2],), dy.stride(), dy.storage_offset())
                    dbias = _sum_partials(ops.colstats(full), cout, bout)
        dx = dw = None
        lazy_bits = None
        early_side = None
        if EARLY_WGRAD_FORK and need_w and need_x and geom is None and WGRAD_SIDE_STREAM and (
                CAPTURE_SIDE_STREAMS or not torch.cuda.is_current_stream_capturing()):
            early_side = wgrad_stream(dy.device)
            _wait_for(early_side, _current_stream())        # dy (and x) are ready here; the data gradient below only reads them
        if dskip is not None:
            ent = _lazy_skip.pop(dskip.data_ptr(), None)
            if ent is not None and tuple(ent[0].shape) == tuple(dskip.shape):
                lazy_bits = ent[1]
        if need_x:
            if geom is None:
                fuse = (dskip is not None and tuple(dskip.shape) == ctx.x_shape and dskip.is_contiguous()
                        and dskip.shape[-1] == wp.cin)
                if lazy_bits is not None and not (fuse and wp.cin % 4 == 0):
                    # cannot mask inside the epilogue: materialise the masked skip gradient first
                    dz = torch.empty(dskip.shape, dtype=dskip.dtype, device=dskip.device)
                    ops.bn_act_bwd(dskip, None, None, None, None, None, None, None, dres=dz, act=ACT_RELU, want_dy=False,
                                   mask_bits=lazy_bits)
                    dskip, lazy_bits = dz, None
                    fuse = (tuple(dskip.shape) == ctx.x_shape and dskip.shape[-1] == wp.cin)
                # identity blocks: the skip gradient is accumulated by the dgrad epilogue instead of a separate add kernel
                in_link = cfg.get("in_link")
                # skip gradient: accumulated in place (already masked), or -- lazy -- read through its ReLU mask as `res`
                # and overwritten in place (each element is read and written by the same thread)
                skip_kw = dict(out=dskip if fuse else None, accumulate=fuse and lazy_bits is None)
                if lazy_bits is not None:
                    skip_kw.update(res=dskip, res_mask_bits=lazy_bits)
                if in_link is not None and in_link.y is not None and (dskip is None or fuse) and \
                        wp.cin == ctx.x_shape[-1] and wp.cin % 4 == 0:
                    dx, part_in = ops.conv2d_dgrad(dy, wp, (ctx.x_shape[1], ctx.x_shape[2]), stride, pad, dil, prec=prec,
                                                   bn_bwd=(in_link.y, in_link.mean, in_link.istd, in_link.msc,
                                                           in_link.msh, in_link.mbits), out_dtype=x.dtype, **skip_kw)
                    in_link.partial, in_link.for_ptr = part_in, dx.data_ptr()
                # (the decoder's 304-channel concat: its data gradient's third 128-column tile multiplies 48 valid columns -- 21 % of
                # the launch on padding.  Two launches into channel slices, 256 columns on the big tiles + 48 on 64-wide tiles, were
                # measured in round 5: 45.12 against 45.11 ms per step -- nothing; the wasted matrix work is not what the backward
                # pass waits for.)
                else:
                    dx = ops.conv2d_dgrad(dy, wp, (ctx.x_shape[1], ctx.x_shape[2]), stride, pad, dil, prec=prec, out_dtype=x.dtype,
                                          **skip_kw)
                if dskip is not None and not fuse:
                    dx = dx + dskip
                if dx.shape[-1] != ctx.x_shape[-1]:  # x carried pad channels
                    dx = _pad_channels(dx, ctx.x_shape[-1])
                    dx = dx.as_strided(ctx.x_shape, dx.stride(), dx.storage_offset())
            else:
                raise RuntimeError("the stem convolution has no data gradient (its input is the image)")
        elif dskip is not None and ctx.needs_input_grad[0]:
            if lazy_bits is not None:
                dz = torch.empty(dskip.shape, dtype=dskip.dtype, device=dskip.device)
                ops.bn_act_bwd(dskip, None, None, None, None, None, None, None, dres=dz, act=ACT_RELU, want_dy=False,
                               mask_bits=lazy_bits)
                dskip = dz
            dx = dskip
        if need_w:
            side = early_side if early_side is not None else (wgrad_stream(dy.device) if (WGRAD_SIDE_STREAM and geom is None and (
                CAPTURE_SIDE_STREAMS or not torch.cuda.is_current_stream_capturing())) else None)
            if side is not None:
                main = _current_stream()
                if early_side is None:
                    _wait_for(side, main)       # dy (and x) are ready on the main stream
                if PLAN_RECORDING:
                    _plan_keep.append((dy, x))  # (see PLAN_RECORDING: alive until the join, no allocator bookkeeping)
                else:
                    dy.record_stream(side)      # keep their memory from being recycled while the side stream reads it
                    x.record_stream(side)
                _set_stream(side)
                try:
                    dw = ops.conv2d_wgrad(dy, x, wp.cout, wp.cin, wp.kh, wp.kw, stride, pad, pad, dil, prec=prec,
                                          out=_bucket_out(weight, wp), x_affine=ctx.x_affine)
                finally:
                    _set_stream(main)
                outer = getattr(ctx, "outer_stream", None) or main    # (a layer on a lane: `main` is the lane, autograd's stream is outer)
                dw.record_stream(outer)         # (allocated on the side stream, read on the main stream by the optimizer)
                first_side = _wgrad_side_of.get(id(weight))
                if first_side is not None:
                    # a weight used by two layers: autograd ADDS this contribution to the first one on the main stream as soon as
                    # both exist -- before the end-of-backward join -- so the main stream has to see both launches finished
                    _wait_for(outer, first_side)
                    _wait_for(outer, side)
                else:
                    _wgrad_side_of[id(weight)] = side
                if not (weight.is_leaf and weight.grad is None):
                    # the gradient is READ inside this backward pass -- accumulated into an existing .grad (a second
                    # backward before the optimizer step) or propagated through a non-leaf weight (a transposed /
                    # computed operand) -- so the main stream cannot wait for the end-of-backward join
                    _wait_for(outer, side)
                if not _join_armed[0]:
                    _join_armed[0] = True
                    torch.autograd.Variable._execution_engine.queue_callback(join_wgrad_stream)
            elif geom is None:
                dw = ops.conv2d_wgrad(dy, x, wp.cout, wp.cin, wp.kh, wp.kw, stride, pad, pad, dil, prec=prec,
                                      out=_bucket_out(weight, wp), x_affine=ctx.x_affine)
            if geom is None:
                dw = dw.permute(0, 3, 1, 2)  # logical OIHW, channels_last memory like the parameter
                if weight.dim() == 2:
                    dw = dw.reshape(weight.shape)
            else:
                dw = cfg["wgrad"](dy, x)
        return dx, dw, dgamma, dbeta, dbias, dres, None


def _sum_partials(part, ncols=None, out=None):
    """[chunks][2][C] partial column sums -> totals of the first `ncols` columns of the first plane (zs3_colsum over rows of stride
    2C), into `out` (a gradient-bucket slice) when given"""
    chunks, _, c = part.shape
    ncols = ncols or c
    out = out.view(ncols) if out is not None and out.numel() == ncols else torch.empty(ncols, dtype=torch.float32, device=part.device)
    ops.check(ops.lib().zs3_colsum(ops.P(part), ops.I(2 * c), ops.I(chunks), ops.I(ncols), ops.P(out), ops.stream()), "zs3_colsum")
    return out


def _applies_in_affine(x, wp, stride, pad, dil, prec, bn, residual, need_grad):
    """Will THIS layer's forward run on a kernel whose producers transform the operand (and its weight gradient likewise)?  The
    launch that decides is the one _ConvBnAct.forward is about to make: with batch statistics (or a gradient to keep) the conv
    has a store-only epilogue; an eval-mode no-grad layer fuses scale / shift / residual into it, and a residual makes it a
    loading epilogue, which the persistent pointwise kernel leaves to the register-staged ones."""
    n, h, w_, _ = x.shape
    ldx = ops._rows(x)[2]
    if wp.cin % 4 or ldx % 4 or x.dtype != torch.float32:
        return False
    prec = ops.PREC_DEFAULT if (prec is None or ops.PREC_DEFAULT == 0) else prec
    store_only = residual is None or (bn is not None and (bn["training"] or need_grad))
    key = (n, h, w_, ldx, wp.cin, wp.cout, wp.kh, wp.kw, stride, pad, dil, prec, store_only, need_grad, ops.HALO, ops.HALO_BM, ops.PW,
           ops.PW_FORCE, ops.WGRAD_STRIP, ops.WGRAD_PW)
    hit = _in_affine_choice.get(key)
    if hit is None:
        ho, wo = ops.conv_out_size(h, wp.kh, stride, pad, dil), ops.conv_out_size(w_, wp.kw, stride, pad, dil)
        tile = ops._choose_tile(0, x.shape, n * ho * wo, ho, wo, wp.cin_pad, min(ops._round_up(wp.cin, 4), ldx), ldx, wp.kh, wp.kw,
                                stride, pad, pad, dil, wp.cout, False, prec, 1 if store_only else 0)
        hit = tile in (41, 42, 51, 52) and not (tile in (51, 52) and wp.cin % 32) and (not need_grad or ops._wgrad_plan(
            n, h, w_, ho, wo, wp.kh, wp.kw, stride, pad, pad, dil, wp.cout, wp.cin)[0] in ("strip", "pw"))
        _in_affine_choice[key] = hit
    return hit


_in_affine_choice = {}


def _bucket_out(weight, wp):
    """wgrad output buffer inside the data-parallel gradient bucket (zero-copy all-reduce), when one is registered"""
    buf = grad_buffer(weight)
    if buf is None or buf.numel() != wp.cout * wp.kh * wp.kw * wp.cin:
        return None
    w4 = weight if weight.dim() == 4 else weight[:, :, None, None]
    if not w4.permute(0, 2, 3, 1).is_contiguous():   # the slice is laid out like the parameter's own storage
        return None
    return buf.view(wp.cout, wp.kh, wp.kw, wp.cin)


def _act_grad(a, act, leak):
    if act == ACT_RELU:
        return (a > 0).to(a.dtype)
    return torch.where(a > 0, torch.ones_like(a), torch.full_like(a, leak))


_defer_choice = {}


def _consumer_applies(x, weight, stride, pad, dil, prec, next_conv):
    """Does `next_conv` (the one consumer of this layer's output) run forward and weight gradient on producer-converting kernels
    that can apply this layer's BatchNorm + ReLU themselves?  Decided once per (output geometry, consumer)."""
    n, h, w_, _ = x.shape
    cout, _, kh, kw = weight.shape if weight.dim() == 4 else (*weight.shape, 1, 1)
    oshape = (n, ops.conv_out_size(h, kh, stride, pad, dil), ops.conv_out_size(w_, kw, stride, pad, dil), cout)
    key = (oshape, tuple(next_conv.weight.shape), next_conv.stride[0], next_conv.padding[0], next_conv.dilation[0],
           next_conv.bias is None, prec, ops.PREC_DEFAULT, ops.HALO, ops.HALO_BM, ops.PW, ops.PW_FORCE, ops.WGRAD_STRIP, ops.WGRAD_PW,
           ops.ACT_DTYPE)
    hit = _defer_choice.get(key)
    if hit is None:
        nw = next_conv.weight
        f16 = bool(ops.fwd_f16() and prec in (None, 3, 4))   # (asked by a layer that normalises with batch statistics: its consumer does too)
        hit = _defer_choice[key] = bool(
            nw.dim() == 4 and next_conv.bias is None and nw.shape[1] == cout and cout % 4 == 0 and
            ops.consumer_applies_bn(oshape, cout, weight_planes(nw, need_t=True, f16=f16), next_conv.stride[0], next_conv.padding[0],
                                    next_conv.dilation[0], prec))
    return hit


def conv_bn_act(x, weight, bn=None, bias=None, residual=None, stride=1, pad=0, dil=1, act=ACT_NONE, out=None,
                leak=0.2, prec=None, geom=None, wgrad=None, pass_through=False, input_has_one_consumer=False, dropout=None,
                next_conv=None, out_dtype=None, lane=None, lane_forked=False):
    """bn: a BatchNorm module-like object with weight/bias/running_mean/running_var/eps/momentum/training, or None.
    input_has_one_consumer: promise that `x` feeds nothing but this layer (and, with pass_through, the skip tensor this
    layer hands back), which lets this layer's dgrad produce the BN-backward sums of the layer that made `x` (BnLink).
    next_conv: the conv module that is the ONLY consumer of this layer's output.  When that conv's forward and weight gradient
    run on kernels whose producer waves convert the operand (ops.consumer_applies_bn), this layer hands over its raw conv
    output tagged with the BatchNorm scale / shift, and the consumer applies BN + ReLU on the way into LDS (DEFER_BN_APPLY)."""
    # dropout: (p, training) of an nn.Dropout that follows this layer's activation.  conv + BN + ReLU layers without a residual
    # (every dropout of the network sits behind one: aspp.py:100, decoder.py:19,23) take it into the BN-apply pass and its
    # backward (same mask as the stand-alone kernel, same position in the seed stream); anything else gets the separate pass.
    if _handed_armed[0] or _join_armed[0] or _lanes_armed[0]:
        _reset_backward_state()
    drop, drop_after = None, None
    if dropout is not None and dropout[1] and dropout[0] > 0.0:
        if dropout[0] >= 1.0:
            drop_after = (1.0, None)
        elif bn is not None and residual is None and act == ACT_RELU and not pass_through and DROPOUT_FUSED:
            drop = (float(dropout[0]), next_seed())
        else:
            drop_after = (float(dropout[0]), None)
    cfg = {"stride": stride, "pad": pad, "dil": dil, "act": act, "out": out, "leak": leak, "prec": prec, "geom": geom,
           "out_dtype": out_dtype, "lane": lane,    # lane: a side stream this layer runs on, forward and backward (see lane_streams)
           "lane_forked": bool(lane_forked),        # the forward's wait for the main stream was made by lanes_fork() already
           "drop": drop,
           "wgrad": wgrad, "pass_through": pass_through,
           "in_link": getattr(x, "_zs3_bn_link", None) if (input_has_one_consumer and FUSE_BN_BWD_STATS) else None,
           # (no hand-off behind a fused dropout: the consumer's dgrad sees the gradient of the DROPPED activation)
           "out_link": BnLink() if (bn is not None and out is None and FUSE_BN_BWD_STATS and drop is None) else None,
           "need_grad": torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in
                                                        (x, weight, bias, residual, getattr(bn, "weight", None)))}
    x_defer = getattr(x, "_zs3_defer", None)
    if x_defer is not None:
        if geom is not None or not input_has_one_consumer:
            raise RuntimeError("a deferred BatchNorm output reached a layer that does not apply it")
        cfg["in_affine"] = x_defer
    cfg["bias_param"] = bias
    gamma = beta = None
    if bn is not None:
        gamma, beta = bn.weight, bn.bias
        cfg["bn_beta"] = beta
        use_batch = bn.training or bn.running_mean is None
        cfg["defer_out"] = bool(
            DEFER_BN_APPLY and next_conv is not None and use_batch and act == ACT_RELU and residual is None and out is None and
            drop is None and drop_after is None and geom is None and x.is_cuda and
            _consumer_applies(x, weight, stride, pad, dil, prec, next_conv))
        mom = bn.momentum
        nbt = None
        if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
            if mom is None:   # cumulative moving average needs the count on the host
                bn.num_batches_tracked.add_(1)
                mom = 1.0 / float(bn.num_batches_tracked)
            else:
                nbt = bn.num_batches_tracked   # incremented by the finalize kernel
        cfg["bn"] = {"training": use_batch, "nbt": nbt, "eps": bn.eps, "momentum": mom if mom is not None else 0.0,
                     "sync": getattr(bn, "_zs3_sync_group", None) if use_batch else None,   # eval never touches torch.distributed
                     "running_mean": bn.running_mean if (bn.training and bn.track_running_stats) or not use_batch else None,
                     "running_var": bn.running_var if (bn.training and bn.track_running_stats) or not use_batch else None}
    res = _ConvBnAct.apply(x, weight, gamma, beta, bias, residual, cfg)
    if cfg.get("deferred") is not None:
        (res[0] if pass_through else res)._zs3_defer = cfg["deferred"]
    if cfg.get("out_link") is not None:
        (res[0] if pass_through else res)._zs3_bn_link = cfg["out_link"]
    if pass_through:
        res[1]._zs3_skip_alias = True   # lets the block's last layer know who will receive its skip gradient
    if drop_after is not None:
        if pass_through:
            raise NotImplementedError("dropout behind a pass-through layer")
        res = _dropout_pass(res, drop_after[0], True)
    return res


# ---------------------------------------------------------------------------------- standalone BN (+act)
class _BnAct(torch.autograd.Function):
    """BatchNorm (+ReLU) on an NHWC / [M, C] tensor that was not produced by one of our convs."""

    @staticmethod
    def forward(ctx, y, gamma, beta, cfg):
        y = _dense_rows(y)
        m = y.numel() // y.shape[-1]
        if cfg["training"]:
            if m <= 1:
                raise ValueError(f"Expected more than 1 value per channel when training, got input size {tuple(y.shape)}")
            part, count = ops.colstats(y), m
            if cfg.get("sync") is not None:   # SynchronizedBatchNorm2d under more than one rank: global sums and count
                from .parallel import combine_bn_partials
                part, count = combine_bn_partials(part, count, None if cfg["sync"] is True else cfg["sync"])
            st = ops.bn_fwd_finalize(part, count, gamma, beta, cfg["eps"], cfg["momentum"], cfg["running_mean"],
                                     cfg["running_var"], cfg.get("nbt"))
        else:
            st = ops.bn_eval_affine(gamma, beta, cfg["running_mean"], cfg["running_var"], cfg["eps"])
        a = ops.affine_act(y, st[2], st[3], act=cfg["act"], out=cfg.get("out"))
        ctx.cfg = cfg
        ctx.save_for_backward(y, gamma, a if cfg["act"] != ACT_NONE else None, st)
        return a

    @staticmethod
    def backward(ctx, dA):
        y, gamma, a, st = ctx.saved_tensors
        cfg = ctx.cfg
        dA = _dense_rows(dA)
        m = y.numel() // y.shape[-1]
        part = ops.bn_bwd_stats(dA, a, y, st[0], st[1])
        fin = ops.bn_bwd_finalize(part, m, cfg["training"], out=(grad_buffer(gamma), grad_buffer(cfg.get("beta_param"))))
        c1, c2 = (fin[2], fin[3]) if cfg["training"] else (None, None)
        if cfg["training"] and cfg.get("sync") is not None:
            from .parallel import combine_bn_partials   # dgamma / dbeta stay per rank (summed by GradSync); c1, c2 are global
            gpart, gcount = combine_bn_partials(part, m, None if cfg["sync"] is True else cfg["sync"])
            fin_g = ops.bn_bwd_finalize(gpart, gcount, True, want_param_grads=False)
            c1, c2 = fin_g[2], fin_g[3]
        dy = ops.bn_act_bwd(dA, a, y, st[0], st[1], gamma, c1, c2, act=cfg["act"])
        return dy.reshape(y.shape), fin[0], fin[1], None


def bn_act(y, bn, act=ACT_NONE, out=None):
    use_batch = bn.training or bn.running_mean is None
    mom = bn.momentum
    nbt = None
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        if mom is None:
            bn.num_batches_tracked.add_(1)
            mom = 1.0 / float(bn.num_batches_tracked)
        else:
            nbt = bn.num_batches_tracked
    track = bn.training and bn.track_running_stats
    cfg = {"training": use_batch, "nbt": nbt, "eps": bn.eps, "momentum": mom if mom is not None else 0.0, "act": act, "out": out,
           "sync": getattr(bn, "_zs3_sync_group", None) if use_batch else None,
           "running_mean": bn.running_mean if track or not use_batch else None,
           "running_var": bn.running_var if track or not use_batch else None, "beta_param": bn.bias}
    return _BnAct.apply(y, bn.weight, bn.bias, cfg)


class _Relu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _dense_rows(x)
        a = ops.affine_act(x, act=ACT_RELU)
        ctx.save_for_backward(a)
        return a

    @staticmethod
    def backward(ctx, dA):
        (a,) = ctx.saved_tensors
        dA = _dense_rows(dA)
        out = torch.empty(dA.shape, dtype=dA.dtype, device=dA.device)
        ops.bn_act_bwd(dA, a, None, None, None, None, None, None, dy=None, dres=out, act=ACT_RELU, want_dy=False)
        return out


def relu(x):
    return _Relu.apply(x)


# ---------------------------------------------------------------------------------- pooling / resize
class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, stride, pad):
        x = _dense_rows(x)
        out, idx = ops.maxpool_fwd(x, k, stride, pad)
        ctx.save_for_backward(idx)
        ctx.geom = (x.shape[1], x.shape[2], k, stride, pad)
        return out

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        h, w, k, stride, pad = ctx.geom
        return ops.maxpool_bwd(_dense_rows(dy), idx, (h, w), k, stride, pad), None, None, None


def max_pool(x, k=3, stride=2, pad=1):
    return _MaxPool.apply(x, k, stride, pad)


class _Bilinear(torch.autograd.Function):
    """F.interpolate(mode='bilinear', align_corners=True) on NHWC; `out` may be a channel slice of a concat buffer."""

    @staticmethod
    def forward(ctx, x, size, out, grad_pad):
        x = _dense_rows(x)
        ctx.in_hw = (x.shape[1], x.shape[2])
        ctx.grad_pad = grad_pad
        return ops.bilinear_fwd(x, size, out=out)

    @staticmethod
    def backward(ctx, dout):
        dout = _dense_rows(dout)
        n, _, _, c = dout.shape
        h, w = ctx.in_hw
        out = None
        if ctx.grad_pad and c % ctx.grad_pad:
            cp = (c + ctx.grad_pad - 1) // ctx.grad_pad * ctx.grad_pad
            out = ops.zeros((n, h, w, cp), dout.dtype, dout.device)[..., :c]
        return ops.bilinear_bwd(dout, (h, w), out=out), None, None, None


def bilinear(x, size, out=None, grad_pad=8):
    return _Bilinear.apply(x, tuple(int(s) for s in size), out, grad_pad)


class _GlobalAvgPool(torch.autograd.Function):
    """nn.AdaptiveAvgPool2d((1,1)) (aspp.py:85): NHWC [N,H,W,C] -> [N,1,1,C]."""

    @staticmethod
    def forward(ctx, x):
        x = _dense_rows(x)
        n, h, w, c = x.shape
        ctx.shape = (n, h, w, c)
        return ops.group_colsum(x, n, 1.0 / (h * w)).view(n, 1, 1, c)

    @staticmethod
    def backward(ctx, dy):
        n, h, w, c = ctx.shape
        dy = dy.reshape(n, c).contiguous()
        return ops.affine_act(dy, alpha=1.0 / (h * w), div=h * w, out_shape=(n, h, w, c))


def global_avg_pool(x):
    return _GlobalAvgPool.apply(x)


class _Broadcast(torch.autograd.Function):
    """Bilinear resize of a 1x1 map (align_corners=True) = broadcast (aspp.py:109), written into `out`."""

    @staticmethod
    def forward(ctx, x, size, out):
        n, _, _, c = x.shape
        ctx.n = n
        x2 = x.reshape(n, c).contiguous()
        return ops.affine_act(x2, div=size[0] * size[1], out=out, out_shape=(n, size[0], size[1], c))

    @staticmethod
    def backward(ctx, dout):
        dout = _dense_rows(dout)
        c = dout.shape[-1]
        return ops.group_colsum(dout, ctx.n).view(ctx.n, 1, 1, c), None, None


def broadcast_to(x, size, out=None):
    return _Broadcast.apply(x, tuple(size), out)


class _Fork(torch.autograd.Function):
    """x -> n aliases of x for n consumers; the backward adds the n gradients in ONE launch (zs3_sum_n) in consumer order,
    where autograd would add them pairwise (n-1 passes over the tensor: ASPP's input has five consumers, aspp.py:104-108)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)
        return tuple(x.view(x.shape) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        lanes_join()      # consumers that ran on lanes (ASPP's branches): their gradients meet here, on the main stream
        gs = [g for g in grads if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return gs[0], None
        dense = [_dense_rows(g) for g in gs]
        same = all(g.shape == dense[0].shape and g.stride() == dense[0].stride() and g.dtype == dense[0].dtype and g.is_cuda
                   and g.dtype in (torch.float32, torch.bfloat16)
                   and g.data_ptr() % 16 == 0 for g in dense)   # zs3_sum_n reads float4: an offset view takes the plain adds
        n = dense[0].numel()
        if not same or len(dense) > 8 or n % 4 or ops._rows(dense[0])[2] != dense[0].shape[-1]:
            out = dense[0]
            for g in dense[1:]:
                out = out + g
            return out, None
        out = torch.empty_like(dense[0])
        ptrs = (ctypes.c_void_p * len(dense))(*[g.data_ptr() for g in dense])
        ops.check(ops.lib().zs3_sum_n(ptrs, ops.I(len(dense)), ops.P(out), ctypes.c_long(n),
                                      ops.I(3 if out.dtype == torch.bfloat16 else 0), ops.stream()), "zs3_sum_n")
        return out, None


FORK_SUM = True   # False: leave the fan-out gradients to autograd's pairwise accumulation


def fork(x, n):
    """n aliases of x, one per consumer (see _Fork); a plain tuple of x when nothing needs a gradient"""
    if n < 2 or not FORK_SUM or not (torch.is_grad_enabled() and x.requires_grad):
        return (x,) * n
    return _Fork.apply(x, n)


class _CatSlices(torch.autograd.Function):
    """torch.cat along channels (aspp.py:110, decoder.py:37) without a copy: the producers wrote straight
    into channel slices of `buffer`; this node only ties the slices to the buffer for autograd."""

    @staticmethod
    def forward(ctx, buffer, *parts):
        ctx.widths = [p.shape[-1] for p in parts]
        return buffer.view(buffer.shape)

    @staticmethod
    def backward(ctx, dbuf):
        grads, c0 = [], 0
        for w in ctx.widths:
            grads.append(dbuf[..., c0:c0 + w])
            c0 += w
        return (None, *grads)


def cat_slices(buffer, parts):
    return _CatSlices.apply(buffer, *parts)


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        x = _dense_rows(x)
        ctx.p, ctx.seed = p, seed
        return ops.dropout(x, p, seed)

    @staticmethod
    def backward(ctx, dy):
        return ops.dropout(_dense_rows(dy), ctx.p, ctx.seed), None, None


def dropout(x, p, training):
    if not training or p <= 0.0:
        return x
    if p >= 1.0:
        return x * 0.0
    return _Dropout.apply(x, float(p), next_seed())


_dropout_pass = dropout   # conv_bn_act has a keyword of that name
