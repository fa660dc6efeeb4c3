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
