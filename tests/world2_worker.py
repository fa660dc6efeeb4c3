"""Child process of tests/test_gpu_world2.py: ONE data-parallel rank of the real step -- Trainer(distributed=True) over the HIP
engine, stage-done callback, two-bucket asynchronous all-reduce, fused Adam with 1/world -- on cuda:0.  Two of these run side
by side on the same GPU with backend "gloo" on device tensors (RCCL refuses two ranks on one device); everything except the
transport is what `bench.py --gpus 2` runs.   python world2_worker.py <rank> <world> <port> <side 0|1|2=graphs> <out.pt> [optimizer]"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))


def rank_batch(rank, N=32, S=64, mixed=False):
    """Different data on every rank (closed form: the parent process rebuilds the same batches for its emulation).  mixed: the
    ranks of one step run DIFFERENT input shapes (BASELINE config 5's straggler case: rank 0 a 64x96 cluster, rank 1 a 96x64 one)."""
    import torch
    import cases as C
    H, W = ((S, S + 32) if rank % 2 == 0 else (S + 32, S)) if mixed else (S, S)
    x = C.det_input((N, 3, H, W), seed=C.INPUT_SEED + 17 * rank)
    t = (torch.arange(N) * 3 + rank) % 10
    return x, t


def main():
    rank, world, port, side, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    opt = sys.argv[6] if len(sys.argv) > 6 else "adam"
    mixed = len(sys.argv) > 7 and sys.argv[7] == "mixed"
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        from test_gpu_train import _no_dropout, build
        from mnasnet_pytorch_amd.train_step import Trainer
        torch.manual_seed(100 + rank)                   # ranks start from DIFFERENT parameters: the rank-0 broadcast must fix that
        m = build("512", proj_gamma=0.1).train()
        _no_dropout(m)
        if rank != 0:
            with torch.no_grad():
                for p in m.parameters():
                    p.add_(0.01 * torch.randn_like(p))
        tr = Trainer(m, lr=1e-3, distributed=True, optimizer=opt)
        tr.engine.use_side_stream = side == 1
        tr.engine.use_graphs = side == 2                # 2: every launch list replayed as a hipGraph (one per backward stage: the callback runs between them)
        tr.engine.reset_programs()
        x, t = rank_batch(rank, mixed=mixed)
        x, t = x.cuda(), t.cuda()
        p0 = tr.flat_p.detach().cpu().clone()           # after the rank-0 broadcast
        logs, losses, g1, p1 = [], [], None, None
        for step in range(3):
            losses.append(float(tr.step(x, t)))
            logs.append(list(tr.schedule.log))
            if step == 0:
                torch.cuda.synchronize()
                g1 = tr.flat_g.detach().cpu().clone()   # the all-reduced (summed) gradient of step 1
                p1 = tr.flat_p.detach().cpu().clone()   # ... and the parameters after the first optimizer step
        torch.cuda.synchronize()
        torch.save({"flat_p": tr.flat_p.detach().cpu(), "flat_p0": p0, "flat_p1": p1, "flat_g1": g1, "logs": logs, "losses": losses,
                    "bounds": list(tr.buckets.bounds), "world": tr.world, "side": tr.engine.use_side_stream,
                    "rm0": m.features[0].bn.running_mean.detach().cpu(), "shape": tuple(x.shape),
                    "programs": sorted(k[1:3] for k, lst in tr.engine.programs.items() if lst)}, out)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
