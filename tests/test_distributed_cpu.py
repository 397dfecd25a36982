"""world_size-2 gloo tests of the N>1 path (zs3_amd.parallel): bucketed gradient SUM all-reduce driven by
grad hooks + end-of-backward callback, parameter broadcast, SyncBN partial-sum combination."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _zero_copy_and_global_ce(rank, world):
    """(1) a layer that writes its weight gradient straight into the GradSync bucket slice (what the fused HIP layers do
    through functional.grad_buffer): autograd adopts the slice as .grad and the all-reduce runs in place, no pack / unpack;
    (2) the product's global CE normalisation (utils.loss.global_ce_normalise) + GradSync's SUM on two DIFFERENT shards
    equals the single-process gradient of the CE over the concatenated batch (loss.py:31-46 on the gathered batch)."""
    import torch.nn.functional as F
    from zs3_amd import functional as Fz
    from zs3_amd.parallel import GradSync
    from zs3_amd.utils.loss import global_ce_normalise

    class BucketLinear(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            ctx.save_for_backward(x, w)
            return x @ w.t()

        @staticmethod
        def backward(ctx, dy):
            x, w = ctx.saved_tensors
            buf = Fz.grad_buffer(w)
            dw = buf.view(w.shape) if buf is not None else torch.empty_like(w)
            torch.mm(dy.t(), x, out=dw)
            return dy @ w, dw

    torch.manual_seed(3)
    classes, feat = 5, 6
    w1 = torch.nn.Parameter(torch.randn(7, feat) * 0.3)
    w2 = torch.nn.Parameter(torch.randn(classes, 7) * 0.3)
    bias = torch.nn.Parameter(torch.zeros(classes))
    cw = torch.tensor([1.0, 2.0, 0.5, 100.0, 1.0])
    sync = GradSync([w1, w2, bias], bucket_mb=0.0001)
    torch.manual_seed(40)
    xs = [torch.randn(3, 4, 4, feat) for _ in range(world)]          # [B, H, W, feat] per rank, different per rank
    ts = [torch.randint(0, classes, (3, 4, 4)) for _ in range(world)]
    ts[0][0, 0, :] = 255
    ts[1][2, 1:, :] = 255

    def logits(x, a, b, c, fn):
        return fn(torch.relu(fn(x.reshape(-1, feat), a)), b) + c

    for it in range(2):
        for p in (w1, w2, bias):
            p.grad = None
        z = logits(xs[rank], w1, w2, bias, BucketLinear.apply)
        t = ts[rank].reshape(-1)
        keep = t != 255
        nll = F.cross_entropy(z[keep], t[keep], reduction="none")
        wt = cw[t[keep]]
        s_local = (wt * nll).sum()
        ws = torch.tensor([0.0, float(wt.sum()), float(s_local)])
        loss_glob, batch_glob = global_ce_normalise(ws, xs[rank].shape[0], True)
        assert batch_glob == 3 * world
        (s_local / ws[1] / batch_glob).backward()      # this rank's share of the global loss
        assert sync._in_place(w1) and sync._in_place(w2) and not sync._in_place(bias)
        # single process, concatenated batch: CE = sum(w*nll)/sum(w) over valid pixels, then / B
        ref = [p.detach().clone().requires_grad_(True) for p in (w1, w2, bias)]
        zc = logits(torch.cat(xs), *ref, lambda a_, b_: a_ @ b_.t())
        tc = torch.cat(ts).reshape(-1)
        lc = F.cross_entropy(zc, tc, weight=cw, ignore_index=255, reduction="mean") / (3 * world)
        gs = torch.autograd.grad(lc, ref)
        assert abs(float(loss_glob) - float(lc)) < 1e-6 * abs(float(lc))
        for p, g_ in zip((w1, w2, bias), gs):
            assert torch.allclose(p.grad, g_, rtol=1e-5, atol=1e-7), (p.shape, (p.grad - g_).abs().max())
    sync.remove()
    assert Fz.grad_buffer(w1) is None


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from zs3_amd.parallel import GradSync, broadcast_parameters, combine_bn_partials
        torch.manual_seed(100 + rank)  # different init per rank: broadcast must fix it
        net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 4, 1), torch.nn.Flatten(),
                                  torch.nn.Linear(4 * 6 * 6, 5), torch.nn.Linear(5, 3))
        net[0].weight.data = net[0].weight.data.contiguous(memory_format=torch.channels_last)
        unused = torch.nn.Parameter(torch.zeros(7))  # never receives a gradient
        broadcast_parameters(net)
        ref = [p.detach().clone() for p in net.parameters()]
        gathered = [torch.zeros_like(ref[0]) for _ in range(world)]
        dist.all_gather(gathered, ref[0].contiguous())
        assert torch.equal(gathered[0], gathered[1])
        sync = GradSync(list(net.parameters()) + [unused], bucket_mb=0.0005)  # several buckets
        assert len(sync.buckets) > 2
        for it in range(2):
            torch.manual_seed(7 + it)
            xs = [torch.randn(2, 3, 6, 6) for _ in range(world)]
            net.zero_grad()
            loss = net(xs[rank]).square().sum()
            from zs3_amd import ops
            flag = ops.range_flag(torch.device("cpu"))
            flag.fill_(1 if (it == 1 and rank == 1) else 0)    # rank 1's forward left fp16's range in the second iteration
            loss.backward()   # hooks launch the all-reduces; the queued callback joins them
            assert int(flag) == (1 if it == 1 else 0)           # ... and every rank knows (the fused SGD skips the step everywhere)
            flag.zero_()
            got = [p.grad.clone() for p in net.parameters()]
            # reference: sum of the per-rank gradients computed locally without hooks
            want = [torch.zeros_like(p) for p in net.parameters()]
            for r in range(world):
                clone = [p.detach().clone().requires_grad_(True) for p in net.parameters()]
                h = torch.nn.functional.conv2d(xs[r], clone[0], clone[1], padding=1).relu()
                h = torch.nn.functional.conv2d(h, clone[2], clone[3]).flatten(1)
                h = torch.nn.functional.linear(torch.nn.functional.linear(h, clone[4], clone[5]), clone[6], clone[7])
                gs = torch.autograd.grad(h.square().sum(), clone)
                want = [w + g for w, g in zip(want, gs)]
            for g_, w_ in zip(got, want):
                assert torch.allclose(g_, w_, rtol=1e-5, atol=1e-6)
            assert net[0].weight.grad.is_contiguous(memory_format=torch.channels_last)
        assert sync.bytes_reduced > 0
        sync.remove()
        _zero_copy_and_global_ce(rank, world)
        # data parallelism by construction: what DeepLab.forward / patch_replication_callback call
        from zs3_amd import parallel
        assert parallel.resolve_group("auto") is True and parallel.resolve_group(None) is None and parallel.resolve_group(False) is None
        # the criterion's "auto" group exchanges only for a logit that needs a gradient: a validation loss under no_grad (possibly on
        # one rank only) is local and never touches torch.distributed
        from zs3_amd.utils.loss import ce_group
        z = torch.zeros(1, 3, 2, 2, requires_grad=True)
        assert ce_group("auto", z) == "auto" and ce_group("auto", z.detach()) is None and ce_group(True, z.detach()) is True
        with torch.no_grad():
            assert ce_group("auto", z) is None
        # broadcast_parameters over an explicit group other than the world must not create the world-wide SyncBN communicator
        # (dist.new_group() is a collective over ALL ranks: with a true sub-group the ranks outside it would never arrive)
        sub = dist.new_group([0, 1])
        saved, parallel._bn_group = parallel._bn_group, None
        torch.manual_seed(400 + rank)
        sub_lin = torch.nn.Linear(2, 2)
        broadcast_parameters(sub_lin, group=sub)
        assert parallel._bn_group is None
        torch.manual_seed(400)
        assert torch.equal(sub_lin.weight, torch.nn.Linear(2, 2).weight)
        parallel._bn_group = saved
        torch.manual_seed(200 + rank)     # different init per rank again: arming must hand every rank rank 0's parameters
        lin = torch.nn.Linear(6, 4)
        armed = parallel.ensure_data_parallel(lin)
        assert armed is not None and parallel.ensure_data_parallel(lin) is armed and lin._zs3_grad_sync is armed
        torch.manual_seed(200)
        want0 = torch.nn.Linear(6, 4)
        assert torch.equal(lin.weight, want0.weight) and torch.equal(lin.bias, want0.bias)
        torch.manual_seed(300)
        xs = [torch.randn(5, 6) for _ in range(world)]
        lin(xs[rank]).sum().backward()    # no GradSync line here: the hooks ensure_data_parallel installed reduce the gradient
        assert torch.allclose(lin.weight.grad, sum(x.sum(0) for x in xs).expand(4, 6), rtol=1e-6, atol=1e-6)
        assert torch.allclose(lin.bias.grad, torch.full((4,), float(5 * world)))
        parallel.disarm_data_parallel(lin)
        assert lin._zs3_grad_sync is None
        lin.zero_grad()
        lin(xs[rank]).sum().backward()
        assert torch.allclose(lin.weight.grad, xs[rank].sum(0).expand(4, 6), rtol=1e-6, atol=1e-6)   # local again
        # SyncBN statistics: global sums / count from per-rank chunk partials
        part = torch.arange(2 * 2 * 4, dtype=torch.float32).reshape(2, 2, 4) * (rank + 1)
        buf, cnt = combine_bn_partials(part, 10 * (rank + 1))
        # -> the fp64 exchange buffer [global sums (2 x C) | global count]; the count travels inside it
        assert cnt is None and buf.dtype == torch.float64 and buf.shape == (2 * 4 + 1,) and buf[8].item() == 30.0
        assert torch.equal(buf[:8].view(2, 4), (torch.arange(16.).reshape(2, 2, 4).sum(0).double()) * 3)
        # the statistics travel on a process group of their own (never queued behind gradient buckets): created once, collectively
        from zs3_amd.parallel import bn_group
        grp = bn_group()
        assert grp is not True and bn_group() is grp and dist.get_world_size(grp) == world
        buf2, _ = combine_bn_partials(part, 10 * (rank + 1), grp)
        assert torch.equal(buf2, buf)
        big = torch.full((3, 2, 4), 1e8 / 3 + rank, dtype=torch.float32)     # sums that fp32 cannot hold exactly
        buf, _ = combine_bn_partials(big, 1)
        other = torch.full((3, 2, 4), 1e8 / 3 + (1 - rank), dtype=torch.float32)
        assert torch.equal(buf[:8].view(2, 4), big.double().sum(0) + other.double().sum(0)) and buf[8].item() == 2.0
        # the GMMN step's exchange: several small tensors (one of them channels_last, one a transposed view) in ONE
        # collective, mean or sum, written back in place
        from zs3_amd.parallel import all_reduce_tensors
        torch.manual_seed(50 + rank)
        a = torch.randn(6, 5)
        b = torch.randn(4, 3, 2, 2).contiguous(memory_format=torch.channels_last)
        c = torch.randn(7, 3).t()                   # non-contiguous: goes through a copy and is written back
        mine = [a.clone(), b.clone(), c.clone()]
        every = []
        for r in range(world):
            torch.manual_seed(50 + r)
            every.append([torch.randn(6, 5), torch.randn(4, 3, 2, 2), torch.randn(7, 3).t()])
        nbytes = all_reduce_tensors([a, None, b, c], average=True)
        assert nbytes == 4 * (30 + 48 + 21)
        for got, col in zip((a, b, c), zip(*every)):
            assert torch.allclose(got, sum(col) / world, rtol=1e-6, atol=1e-7)
        assert b.is_contiguous(memory_format=torch.channels_last) and c.shape == (3, 7)
        all_reduce_tensors(mine, average=False)
        for got, col in zip(mine, zip(*every)):
            assert torch.allclose(got, sum(col), rtol=1e-6, atol=1e-7)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "fail: " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_gradsync_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res
