"""-m gpu: the REAL data-parallel step with world_size 2 on ONE MI355X (src/train.py:197-202 is the reference's only
parallelism).  Two fresh child processes, both on cuda:0, each a real Trainer(distributed=True) over the HIP engine (stage-done
callback, early bucket, side-stream join, fused Adam with grad_scale 1/2), backend gloo on DEVICE tensors (RCCL refuses two ranks
on one device; the transport is the only thing that differs from `bench.py --gpus 2`).  Different data per rank, 3 steps, for both
values of Engine.use_side_stream.  Checked:
  * the parameters (flat_p) are BIT-equal across the two ranks after 3 steps (ranks start from different parameters: the rank-0
    broadcast, identical summed gradients and the identical fused update must make them one model);
  * they equal a single-process emulation -- two separate forward+backward passes over the two ranks' batches accumulated into
    one flat gradient buffer, Adam with grad_scale = 0.5 -- the summed gradient of step 1 to 1e-6 relative L2 and the parameters
    after 3 steps within the bound stated in the test;
  * bucket 0 (head + features.5..7) was launched from the stage-done callback BEFORE the backward of features.4 was enqueued,
    bucket 1 after features.0, in every step;
  * BatchNorm running statistics stay per rank (DataParallel semantics).
The child processes are started with subprocess (a new interpreter each: nothing of this process's GPU state is inherited)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

import cases as C  # noqa: F401  (sys.path set-up shared with the worker)
from test_gpu_train import _no_dropout, build
from world2_worker import rank_batch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_world2(tmp_path, side, opt="adam", mixed=False, world=2):
    port = _free_port()
    outs = [str(tmp_path / ("rank%d_side%d.pt" % (r, side))) for r in range(world)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "world2_worker.py"), str(r), str(world), str(port), str(side), outs[r], opt] + (["mixed"] if mixed else []),
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors="replace")[-3000:])
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, logs[r])
    return [torch.load(o) for o in outs]


def _emulate(opt="adam", mixed=False):
    """Both ranks in ONE process: rank 0's initial parameters, per step two forward+backward passes (one per rank's batch)
    accumulated into the same flat gradient buffer, optimizer with grad_scale = 1/2."""
    from mnasnet_pytorch_amd.train_step import Trainer
    m = build("512", proj_gamma=0.1).train()
    _no_dropout(m)
    tr = Trainer(m, lr=1e-3, optimizer=opt)
    tr.optimizer.grad_scale = 0.5
    batches = [tuple(v.cuda() for v in rank_batch(r, mixed=mixed)) for r in range(2)]
    p0 = tr.flat_p.detach().cpu().clone()
    g1 = p1 = None
    for step in range(3):
        tr.optimizer.zero_grad()
        for x, t in batches:
            tr.forward_backward(x, t)
        if step == 0:
            torch.cuda.synchronize()
            g1 = tr.flat_g.detach().cpu().clone()
        tr.optimizer.step()
        if step == 0:
            torch.cuda.synchronize()
            p1 = tr.flat_p.detach().cpu().clone()
    torch.cuda.synchronize()
    return tr.flat_p.detach().cpu(), g1, p0, p1


def _check_ranks(r0, r1, side):
    assert r0["world"] == r1["world"] == 2 and r0["side"] == r1["side"] == (side == 1)
    # one model on both ranks, bit for bit (rank 1 started from perturbed parameters: the rank-0 broadcast made them equal)
    assert torch.equal(r0["flat_p0"], r1["flat_p0"])
    assert torch.equal(r0["flat_p"], r1["flat_p"])
    assert torch.equal(r0["flat_g1"], r1["flat_g1"])
    assert torch.isfinite(r0["flat_p"]).all()
    # per-rank BatchNorm statistics (different data): NOT synced
    assert not torch.equal(r0["rm0"], r1["rm0"])
    # launch order of every step: ... ("stage", 5), ("launch", 0), ("stage", 4) ... ("stage", 0), ("launch", 1)
    assert r0["bounds"][1] > 0.7 * r0["bounds"][2]
    for res in (r0, r1):
        assert len(res["logs"]) == 3
        for log in res["logs"]:
            log = [tuple(e) for e in log]
            assert log.index(("launch", 0)) == log.index(("stage", 5)) + 1
            assert log.index(("launch", 0)) < log.index(("stage", 4)) < log.index(("stage", 0)) < log.index(("launch", 1))
            assert log.count(("launch", 0)) == 1 and log[-1] == ("launch", 1)


def _compare(r0, emu, what):
    """(a) EXACT part: the summed gradient of step 1 and the parameters after the first optimizer step equal the emulation's to
    fp32 rounding -- the two sums differ in association only (all-reduce: g0 + g1; emulation: the finalize kernels add their
    partial sums into the buffer that already holds g0).  (b) Steps 2-3 start from parameters that differ in the last fp32 bit,
    and this network amplifies such differences by ~1e4 per step (a 1-ulp change of a BatchNorm coefficient flips the bf16
    rounding of a few thousand activations, and 57 BatchNorm layers amplify those ~100x: DESIGN.md section 6): the two
    trajectories are compared with MEASURED bounds (MI355X, round 5: SGD update rel-L2 1.2e-3; Adam 7.8e-2 -- Adam divides by
    sqrt(v), so in its first steps EVERY element moves by ~lr and a 1 % gradient difference is a 1e-5 parameter difference)."""
    p_ref, g_ref, p0, p1_ref = emu
    assert torch.equal(p0, r0["flat_p0"])
    eg = float((r0["flat_g1"].double() - g_ref.double()).norm() / g_ref.double().norm())
    d1 = float((r0["flat_p1"] - p1_ref).abs().max())
    upd = (p_ref - p0).double()
    eu = float(((r0["flat_p"] - p0).double() - upd).norm() / upd.norm())
    d3 = float((r0["flat_p"] - p_ref).abs().max())
    print("world-2 %s: summed gradient of step 1 vs emulation rel-L2 %.3e; parameters after step 1 max |diff| %.3e; 3-step parameter "
          "UPDATE vs emulation rel-L2 %.3e, max |diff| %.3e" % (what, eg, d1, eu, d3))
    assert eg < 1e-6 and d1 < 5e-6        # (measured: Adam 1.0e-6, SGD 9e-10)
    return eu, d3


@pytest.mark.parametrize("side", [0, 1, 2])          # 0 one stream, 1 weight gradients on the second stream, 2 hipGraph replay
def test_real_engine_world2_one_gpu(tmp_path, side):
    """Adam (the reference's optimizer, train.py:219)."""
    r0, r1 = _run_world2(tmp_path, side)
    _check_ranks(r0, r1, side)
    eu, d3 = _compare(r0, _emulate(), "adam (side stream %d)" % side)
    assert eu < 0.25 and d3 <= 3 * 2 * 1e-3 * 1.01        # at most +-lr per element and step


def test_real_engine_world2_one_gpu_sgd(tmp_path):
    """Plain SGD (train.py:226-228; linear in the gradient)."""
    r0, r1 = _run_world2(tmp_path, 0, "sgd")
    _check_ranks(r0, r1, 0)
    eu, d3 = _compare(r0, _emulate("sgd"), "sgd")
    assert eu < 1e-2 and d3 < 5e-6


def test_real_engine_world2_mixed_shapes(tmp_path):
    """BASELINE config 5's straggler case on the REAL engine: the two ranks of every step run different input shapes (64x96 and
    96x64: different compiled programs, different launch lists, stage-done callbacks at different times).  The gradient layout
    does not depend on the input shape, so the buckets still go out in the same order with the same bounds, the ranks end up with
    bit-equal parameters, and step 1 equals the single-process emulation over the two differently shaped batches."""
    r0, r1 = _run_world2(tmp_path, 0, "adam", mixed=True)
    assert tuple(r0["shape"][2:]) == (64, 96) and tuple(r1["shape"][2:]) == (96, 64)
    assert r0["programs"] != r1["programs"]
    _check_ranks(r0, r1, 0)
    eu, d3 = _compare(r0, _emulate("adam", mixed=True), "adam, mixed shapes across ranks")
    assert eu < 0.25 and d3 <= 3 * 2 * 1e-3 * 1.01


def test_bench_multi_rank_path_on_one_gpu():
    """bench.py's own N > 1 code path (rank-0 broadcast inside Trainer, barrier + MAX-over-ranks timing, calibration steps on every
    rank, only rank 0 printing) as the driver launches it -- `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` --
    with --one-device (both ranks on cuda:0 over gloo): ONE JSON line, n_gpus 2, weak scaling, global batch = 2 x per-GPU batch, a
    roofline object, no cpu_baseline (rank 0 at N = 1 only), and the override recorded so that the line can never pass for a
    measurement."""
    import json
    root = os.path.dirname(HERE)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
           "--batch", "32", "--size", "64", "--min-seconds", "0", "--one-device", "--no-box"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3 and d["config"]["global_batch"] == 64
    assert d["value"] > 0 and d["config"]["parallelism"] == "dp2" and "cpu_baseline" not in d
    assert d["roofline"]["launches_per_step"] >= 1 and any("--one-device" in o for o in d["overrides"])


def _bench_line(cmd, timeout=600):
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=dict(env, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_host_bound_guard_switches_to_graphs():
    """bench.py's host-bound guard (round 6): with the threshold forced to 1 % of a step the launch lists must be replayed as
    hipGraphs for the measurement -- on one rank and on two (the decision is the MAX over ranks) -- and the line must say so; with
    the default threshold on this host (enqueue ~65 % of a step) nothing switches."""
    root = os.path.dirname(HERE)
    base = ["--steps", "3", "--warmup", "2", "--batch", "32", "--size", "64", "--min-seconds", "0", "--no-box", "--no-cpu-baseline"]
    d = _bench_line([sys.executable, os.path.join(root, "bench.py"), *base, "--auto-graphs", "0.01"])
    assert d["host_bound_guard"]["switched"] is True and d["graph_mode"]["auto"] is True and d["value"] > 0
    assert "error" not in d["host_bound_guard"]
    d = _bench_line([sys.executable, os.path.join(root, "bench.py"), *base, "--auto-graphs", "0"])
    assert "host_bound_guard" not in d and "graph_mode" not in d
    d = _bench_line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                     "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", *base[:-1], "--one-device",
                     "--auto-graphs", "0.01"])
    assert d["n_gpus"] == 2 and d["host_bound_guard"]["switched"] is True and d["graph_mode"]["auto"] is True and d["value"] > 0


def test_bench_distributed_path_on_rccl_one_rank():
    """bench.py's N > 1 branches on the REAL backend: `--gpus 1 --force-distributed` initialises the process group on "nccl" (= RCCL)
    with device_id, builds Trainer(distributed=True) (rank-0 broadcast, two gradient buckets all-reduced asynchronously on RCCL's
    stream), runs barrier + synchronize around every window, reduces the window time with a float64 MAX all-reduce on the device
    and destroys the group -- as a one-rank job, the only RCCL job one GPU allows.  What it cannot show is ranks >= 1."""
    root = os.path.dirname(HERE)
    d = _bench_line([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-distributed", "--steps", "3", "--warmup", "2",
                     "--batch", "32", "--size", "64", "--min-seconds", "0", "--no-box", "--no-cpu-baseline", "--auto-graphs", "0"])
    assert d["n_gpus"] == 1 and d["value"] > 0 and any("--force-distributed" in o for o in d["overrides"])
    assert d["config"]["parallelism"] == "dp1"


# ---- world 8 on one device (round 6): the day an 8-GPU node exists, the first real run must not fail on plumbing -----------------
def test_real_engine_world8_one_gpu(tmp_path):
    """Eight real Trainer(distributed=True) ranks on cuda:0 over gloo (ranks 2..7 have never existed before round 6): ranks start
    from eight different parameter sets, see eight different batches (even ranks 64x96, odd ranks 96x64: two compiled programs),
    and after 3 Adam steps hold ONE model bit for bit; bucket 0 goes out right after stage 5 on every rank in every step;
    BatchNorm statistics stay per rank."""
    res = _run_world2(tmp_path, 0, "adam", mixed=True, world=8)
    assert all(r["world"] == 8 for r in res)
    for r in res[1:]:
        assert torch.equal(res[0]["flat_p0"], r["flat_p0"]) and torch.equal(res[0]["flat_g1"], r["flat_g1"])
        assert torch.equal(res[0]["flat_p"], r["flat_p"])
    assert torch.isfinite(res[0]["flat_p"]).all() and not torch.equal(res[0]["flat_p"], res[0]["flat_p0"])
    assert len({r["rm0"].numpy().tobytes() for r in res}) == 8                      # per-rank running statistics
    for r in res:
        assert len(r["logs"]) == 3
        for log in r["logs"]:
            log = [tuple(e) for e in log]
            assert log.index(("launch", 0)) == log.index(("stage", 5)) + 1
            assert log.index(("launch", 0)) < log.index(("stage", 4)) < log.index(("stage", 0)) < log.index(("launch", 1))
            assert log.count(("launch", 0)) == 1 and log[-1] == ("launch", 1)


def test_bench_world8_dry_run_on_one_gpu(tmp_path):
    """The driver's multi-GPU command line, dry: plain `python3 bench.py --gpus 8 --one-device --batch 32 --steps 5 --warmup 2`
    (bench.py starts its eight ranks as a CHILD torch.distributed.run on a free port; nothing is exec'ed from a process that
    has touched the GPU).  ONE JSON line, n_gpus 8, global batch 256; every rank finished with bit-equal parameters and the same
    bucket order; only rank 0 ran the roofline calibration; per-rank BatchNorm statistics differ."""
    import json
    root = os.path.dirname(HERE)
    dump = str(tmp_path / "state")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--one-device", "--batch", "32", "--steps", "5", "--warmup", "2",
           "--min-seconds", "0", "--dump-state", dump]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=dict(env, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["steps"] == 5 and d["warmup"] == 2
    assert d["config"]["global_batch"] == 256 and d["config"]["parallelism"] == "dp8" and "cpu_baseline" not in d
    assert d["value"] > 0 and d["roofline"]["launches_per_step"] >= 1 and any("--one-device" in o for o in d["overrides"])
    assert "copy4_nt_GBps" in d["box"]["before"] and "read_GBps" in d["box"]["after"]
    st = [json.load(open(os.path.join(dump, "rank%d.json" % k))) for k in range(8)]
    assert [s["rank"] for s in st] == list(range(8)) and all(s["world"] == 8 for s in st)
    assert len({s["flat_p_sha256"] for s in st}) == 1                       # one model on all eight ranks, bit for bit
    assert len({s["running_mean0_sha256"] for s in st}) == 8                # per-rank BatchNorm statistics (different data per rank)
    assert [s["profiled"] for s in st] == [True] + [False] * 7              # roofline calibration on rank 0 only
    for s in st:
        log = [tuple(e) for e in s["schedule"]]
        assert log.index(("launch", 0)) == log.index(("stage", 5)) + 1 and log[-1] == ("launch", 1)


def test_bench_refuses_more_gpus_than_the_box_has():
    """`bench.py --gpus 8` WITHOUT --one-device on a box with fewer GPUs: a clear message and a non-zero exit code right away (it
    used to die inside torch.cuda.set_device / hang in the rendezvous), both as the plain command and as a torchrun rank."""
    if torch.cuda.device_count() >= 8:
        pytest.skip("this box has 8 GPUs")
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "needs 8 visible GPUs" in (r.stderr + r.stdout)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300,
                       env=dict(env, RANK="5", LOCAL_RANK="5", WORLD_SIZE="8", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port())))
    assert r.returncode != 0 and "needs 8 visible GPUs" in (r.stderr + r.stdout)
