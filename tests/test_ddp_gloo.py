"""CPU, world_size 2, gloo: the N>1 path's host logic (train_step.FlatBuckets: contiguous gradient buckets,
asynchronous all-reduce, wait) and the data-parallel semantics it implements -- per-rank (unsynced) BatchNorm,
gradients summed over ranks then divided by world size (DataParallel's reduce-add, train.py:202, with the
1/world folded into the optimizer) -- using the CPU oracle as the per-rank model."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases as C
from cases import O
from mnasnet_pytorch_amd.train_step import FlatBuckets


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    try:
        # 1) bucket mechanics on a synthetic flat buffer with an odd split
        n = 1000
        flat = O.det_uniform((n,), 100 + rank).clone()
        expect = sum(O.det_uniform((n,), 100 + r) for r in range(world))
        fb = FlatBuckets(flat, [0, 137, n])
        fb.launch(0)          # "early" bucket, launched from the engine's stage-done callback
        fb.launch(1)
        fb.wait()
        ok1 = bool(torch.allclose(flat, expect, atol=1e-6))
        # empty bucket is a no-op
        fb2 = FlatBuckets(flat.clone(), [0, 0, n])
        fb2.launch(0); fb2.launch(1); fb2.wait()
        # 2) data-parallel step semantics with the oracle as the replica: each rank sees its half of the batch,
        #    BatchNorm statistics are per rank, gradients are reduce-added
        ccf, N, H, W = False, 4, 32, 32
        st = O.init_state(ccf, C.STATE_SEED, proj_gamma=0.1)
        params = []
        seen = set()
        for k, v in st.items():
            if v.dtype.is_floating_point and "running" not in k and id(v) not in seen:
                seen.add(id(v)); v.requires_grad_(True); params.append(v)
        x = C.det_input((N, 3, H, W))
        xs = x[rank * N // world:(rank + 1) * N // world]
        y = O.features_forward(xs, st, ccf, True)
        (y * C.cotangent(tuple(y.shape), seed=50 + rank)).sum().backward()
        sizes = [p.numel() for p in params]
        flatg = torch.cat([p.grad.reshape(-1) for p in params])
        local = flatg.clone()
        split = sum(sizes[:len(sizes) // 3])
        fb3 = FlatBuckets(flatg, [0, split, flatg.numel()])
        fb3.launch(0); fb3.launch(1); fb3.wait()
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        ok2 = bool(torch.allclose(flatg, sum(gathered), rtol=1e-5, atol=1e-7))
        # per-rank BatchNorm: running stats differ across ranks (they saw different data) -- not synced
        rm = st["features.0.bn.running_mean"].clone()
        rms = [torch.zeros_like(rm) for _ in range(world)]
        dist.all_gather(rms, rm)
        ok3 = not torch.allclose(rms[0], rms[1])
        q.put((rank, ok1, ok2, ok3))
    finally:
        dist.destroy_process_group()


def test_flat_buckets_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok1, ok2, ok3 in res:
        assert ok1, "bucketed all-reduce != sum over ranks (rank %d)" % rank
        assert ok2, "gradient buckets != reduce-add of per-rank gradients (rank %d)" % rank
        assert ok3, "BatchNorm statistics must stay per-rank (DataParallel semantics)"


def test_bucket_bounds_validation():
    with pytest.raises(AssertionError):
        FlatBuckets(torch.zeros(10), [0, 11, 10])
    with pytest.raises(AssertionError):
        FlatBuckets(torch.zeros(10), [1, 5, 10])


# ---- the ORDER in which Trainer launches its buckets, with a stub engine (no GPU): train_step.BucketSchedule -----------------
def _stub_layout():
    """flat gradient layout of a head (50 elements) + 8 engine stages, later stages first, as engine.stage_ranges gives it"""
    sizes = {7: 400, 6: 300, 5: 120, 4: 60, 3: 30, 2: 20, 1: 10, 0: 10}
    ranges, off = {}, 0
    for st in sorted(sizes, reverse=True):
        ranges[st] = [off, off + sizes[st]]
        off += sizes[st]
    return 50, ranges, off


def _sched_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    try:
        from mnasnet_pytorch_amd.train_step import BucketSchedule
        n_head, ranges, n_eng = _stub_layout()
        n = n_head + n_eng
        flat = torch.zeros(n)
        sched = BucketSchedule(flat, n_head, ranges, early_bucket_stage=5)
        oks = []
        for step in range(2):                   # two steps: begin_step() must re-arm the early launch
            flat.zero_()
            sched.begin_step()
            launch_seen_at = {}
            orig = sched.buckets.launch

            def launch(i, _orig=orig):
                # what the flat buffer looked like when bucket i was handed to the collective: stages written so far
                launch_seen_at[i] = [st for st in ranges if float(flat[n_head + ranges[st][0]]) != 0.0]
                _orig(i)
            sched.buckets.launch = launch
            # stub engine: the head's gradients exist before the features backward; then stages 7..0 in backward order, each
            # writing ITS slice of the flat buffer right before its stage-done callback (as Program.run_backward does)
            flat[:n_head] = 1.0 + rank
            for st in sorted(ranges, reverse=True):
                a, b = ranges[st]
                flat[n_head + a:n_head + b] = (st + 1) * (1.0 + rank) + step
                sched.on_stage_done(st)
            sched.finish()
            sched.buckets.launch = orig
            # ordering: bucket 0 right after stage 5's callback, before stage 4 wrote anything; bucket 1 at the end
            want_log = [("stage", 7), ("stage", 6), ("stage", 5), ("launch", 0)] + [("stage", s) for s in (4, 3, 2, 1, 0)] + [("launch", 1)]
            oks.append(sched.log == want_log)
            oks.append(sorted(launch_seen_at[0]) == [5, 6, 7] and sorted(launch_seen_at[1]) == list(range(8)))
            # result: every element is the sum over ranks of what the ranks wrote
            tot = sum(1.0 + r for r in range(world))
            exp = torch.empty(n)
            exp[:n_head] = tot
            for st in ranges:
                a, b = ranges[st]
                exp[n_head + a:n_head + b] = (st + 1) * tot + step * world
            oks.append(bool(torch.allclose(flat, exp)))
        oks.append(sched.join_stages == {5} and sched.buckets.bounds == [0, n_head + ranges[5][1], n])
        # a model without late stages (engine rooted at an early stage): everything goes at the end, nothing deadlocks
        s2 = BucketSchedule(torch.ones(40), 0, {1: [0, 30], 0: [30, 40]}, early_bucket_stage=5)
        s2.begin_step(); s2.on_stage_done(1); s2.on_stage_done(0); s2.finish()
        oks.append(s2.log == [("stage", 1), ("stage", 0), ("launch", 0), ("launch", 1)] and s2.early_stage is None)
        oks.append(bool(torch.allclose(s2.buckets.flat, torch.full((40,), float(world)))))
        q.put((rank, oks))
    finally:
        dist.destroy_process_group()


def test_bucket_schedule_order_world2_gloo():
    """Trainer's overlap logic without a GPU: bucket 0 (head + features.5..7) is all-reduced from the stage-done callback of
    stage 5 -- after its gradients exist, before stage 4's backward is enqueued -- bucket 1 at the end; begin_step() re-arms."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sched_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, oks in res:
        assert all(oks), (rank, oks)


# ---- config 5's straggler case: every rank runs a DIFFERENT input shape (mixed-across-ranks clusters) ------------------------
def _mixed_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    try:
        import time
        from mnasnet_pytorch_amd.train_step import BucketSchedule
        ccf = False
        shapes = [(2, 3, 32, 48), (2, 3, 48, 32)]          # rank 0 / rank 1: different "programs" (datasets.py:331-335 clusters)
        st = O.init_state(ccf, C.STATE_SEED, proj_gamma=0.1)
        # unique trainable tensors grouped by features.<stage>, flat layout = later stages first (engine.stage_ranges)
        seen, by_stage = set(), {}
        for k, v in st.items():
            if v.dtype.is_floating_point and "running" not in k and id(v) not in seen:
                seen.add(id(v)); v.requires_grad_(True)
                by_stage.setdefault(int(k.split(".")[1]), []).append(v)
        ranges, off = {}, 0
        for stg in sorted(by_stage, reverse=True):
            n_ = sum(p.numel() for p in by_stage[stg])
            ranges[stg] = [off, off + n_]
            off += n_
        n_head = 7
        flat = torch.zeros(n_head + off)
        sched = BucketSchedule(flat, n_head, ranges, early_bucket_stage=5)
        oks = []
        for step in range(2):
            x = C.det_input(shapes[(rank + step) % 2], seed=3 + step)      # the shapes swap between steps
            for ps in by_stage.values():
                for p in ps:
                    p.grad = None
            y = O.features_forward(x, st, ccf, True)
            (y * C.cotangent(tuple(y.shape), seed=60 + rank)).sum().backward()
            local = {stg: torch.cat([p.grad.reshape(-1) for p in ps]) for stg, ps in by_stage.items()}
            flat.zero_()
            sched.begin_step()
            flat[:n_head] = 1.0 + rank
            for stg in sorted(ranges, reverse=True):       # backward order; the "slow" rank of this step lags behind
                if (rank + step) % 2 == 1:
                    time.sleep(0.02)
                a, b = ranges[stg]
                flat[n_head + a:n_head + b] = local[stg]
                sched.on_stage_done(stg)
            sched.finish()
            # expected: reduce-add over ranks of what every rank computed on ITS shape
            mine = torch.cat([torch.full((n_head,), 1.0 + rank)] + [local[stg] for stg in sorted(ranges, reverse=True)])
            gathered = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(gathered, mine)
            oks.append(bool(torch.allclose(flat, sum(gathered), rtol=1e-5, atol=1e-6)))
            oks.append([e for e in sched.log if e[0] == "launch"] == [("launch", 0), ("launch", 1)])
            oks.append(sched.log.index(("launch", 0)) == sched.log.index(("stage", 5)) + 1)
        q.put((rank, oks))
    finally:
        dist.destroy_process_group()


def test_mixed_shape_ranks_world2_gloo():
    """BASELINE configs[4] across ranks (SURVEY 8(e) caveat): the ranks of one step hold DIFFERENT resolution clusters, i.e. run
    different launch lists and reach their stage-done callbacks at different times.  The gradient layout does not depend on the
    input shape, so the two buckets must still be all-reduced in the same order with the same bounds on every rank and the
    result must be the reduce-add of the per-rank gradients -- whichever rank is the straggler (they swap between the steps)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mixed_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, oks in res:
        assert all(oks), (rank, oks)


def test_bench_multi_gpu_spawn_is_a_child_process():
    """bench.py --gpus N (plain python) must start torch.distributed.run as a CHILD before anything touches the GPU and never
    exec: checked on the source (the GPU box refuses an exec from a process that initialised HIP)."""
    src = open(os.path.join(os.path.dirname(__file__), "..", "bench.py")).read()
    head = src[src.index("def main():"):src.index("torch.cuda.set_device")]
    assert "subprocess.run(cmd)" in head and "torch.distributed.run" in head
    assert "os.exec" not in src and "execv" not in src
