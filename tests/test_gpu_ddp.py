"""-m gpu: the RCCL path end to end on ONE GPU (world_size 1, backend nccl): process-group init, rank-0 parameter
broadcast, the engine's stage-done callback launching bucket 0 early, bucket 1 + wait, fused Adam with 1/world.
With one rank the all-reduce is the identity, so the result must equal the non-distributed Trainer's.
(The multi-rank arithmetic is covered on CPU/gloo by tests/test_ddp_gloo.py; the driver runs N = 2, 4, 8.)"""
import os
import socket

import pytest
import torch

import cases as C
from test_gpu_train import _no_dropout, build

pytestmark = pytest.mark.gpu


def test_trainer_distributed_world1():
    import torch.distributed as dist
    from mnasnet_pytorch_amd.train_step import Trainer
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        x = C.det_input((4, 3, 64, 64)).cuda()
        target = torch.tensor([1, 3, 5, 7]).cuda()
        m1 = build("512", proj_gamma=0.1).train(); _no_dropout(m1)
        t1 = Trainer(m1, lr=1e-3)
        l1 = [float(t1.step(x, target)) for _ in range(3)]
        m2 = build("512", proj_gamma=0.1).train(); _no_dropout(m2)
        t2 = Trainer(m2, lr=1e-3, distributed=True)
        assert t2.buckets is not None and t2.buckets.n == 2 and t2.world == 1
        # bucket 0 = head + late stages: most of the bytes
        assert t2.buckets.bounds[1] > 0.7 * t2.buckets.bounds[2]
        launched = []
        orig = t2.buckets.launch
        t2.buckets.launch = lambda i: (launched.append(i), orig(i))[1]
        l2 = [float(t2.step(x, target)) for _ in range(3)]
        assert launched[:2] == [0, 1]                   # early bucket first (from the stage-done callback), then the rest
        assert abs(l1[0] - l2[0]) <= 5e-3 * abs(l1[0]) and abs(l1[2] - l2[2]) <= 3e-2 * abs(l1[2]), (l1, l2)
        t2.sync_buffers()
    finally:
        dist.destroy_process_group()
