"""CPU: host logic of bench.py's box calibration (round 5) -- the sysfs parser, the profiler guard, the smi snapshot never raising."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_sysfs_clocks_parses_the_starred_level(tmp_path):
    b = _bench()
    d = tmp_path / "dev"
    hw = d / "hwmon" / "hwmon3"
    hw.mkdir(parents=True)
    (d / "pp_dpm_sclk").write_text("0: 132Mhz\n1: 2100Mhz *\n2: 2400Mhz\n")
    (d / "pp_dpm_mclk").write_text("0: 2000Mhz *\n")
    (hw / "power1_cap").write_text("1400000000\n")
    (hw / "power1_input").write_text("919000000\n")
    (hw / "freq1_input").write_text("2335000000\n")
    (d / "power_dpm_force_performance_level").write_text("auto\n")
    r = b.sysfs_clocks(str(d))
    assert r["sclk"] == "2100Mhz" and r["sclk_levels"] == 3 and r["mclk"] == "2000Mhz"
    assert r["power_cap_w"] == 1400.0 and r["power_input_w"] == 919.0 and r["sclk_mhz_hwmon"] == 2335.0 and r["perf_level"] == "auto"
    assert b.sysfs_clocks(None) is None and b.sysfs_clocks(str(tmp_path / "missing")) is None


def test_smi_snapshot_never_raises():
    b = _bench()
    r = b.smi_snapshot()                       # no GPU in this container: None, an error record or whatever the tool printed
    assert r is None or isinstance(r, dict)


def test_class_key_and_work_accounting():
    """SURVEY 8(d) accounting of one launch record: a depthwise backward moves 4 tensors, a fused 1x1 backward 2M(2Co+2Ci) bytes."""
    b = _bench()
    from mnasnet_pytorch_amd import _lib as L
    assert b.class_key(L.OP_DW_BWD, (256, 14, 14, 576, 5, 1024, 0, 0, 1), L) == "k_dw_bwd"
    nb, fl = b.launch_work(L.OP_DW_BWD, (256, 14, 14, 576, 5, 1024, 0, 0, 1), L)
    assert nb == 2 * 4 * 256 * 14 * 14 * 576 and fl == 4.0 * 256 * 14 * 14 * 576 * 25
    nb, _ = b.launch_work(L.OP_PW_BWD, (50176, 576, 96, 85, 1, 0), L)
    assert nb == 2 * 50176 * (2 * 96 + 2 * 576)
    assert b.class_key(L.OP_CONV_GEMM, (1, 256, 7, 7, 320, 7, 7, 192, 3, 3), L) == "k_igemm<dgrad>"


def test_bench_refuses_more_gpus_than_visible():
    """`bench.py --gpus 8` on a box with fewer GPUs (none here): clear message, non-zero exit, no JSON line, no child processes --
    as the plain command and as one rank of a torchrun launch (round 6; tests/test_gpu_world2.py repeats it on the GPU box)."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    for extra in ({}, {"RANK": "5", "LOCAL_RANK": "5", "WORLD_SIZE": "8", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"}):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"],
                           capture_output=True, text=True, timeout=300, env=dict(env, **extra))
        assert r.returncode != 0 and "needs 8 visible GPUs" in (r.stderr + r.stdout), r.stderr[-500:]
        assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_gpu_maybe_initialised_guard(monkeypatch):
    """The opt-in rocm-smi child process is refused whenever a tool library may have initialised the GPU."""
    b = _bench()
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    assert b.gpu_maybe_initialised() and "skipped" in b.smi_snapshot()
    monkeypatch.delenv("LD_PRELOAD")
    monkeypatch.setenv("ROCPROF_OUTPUT_PATH", "/tmp/x")
    assert b.gpu_maybe_initialised()
