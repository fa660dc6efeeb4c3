#!/usr/bin/env python3
"""bench.py -- images/sec of the MNASNet train step (src/train.py:423-440 of the reference) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]/[2]): MNASNet-1.0 = Mnasnet(cut_channels_first=False) + FineTuneModelPool
head '512', num_classes 1000, bs=256 per GPU, synthetic 3x224x224 fp32 batches resident in HBM,
CrossEntropyLoss, Adam(lr=1e-3), model.train(); one step = forward + backward + gradient all-reduce (N>1) +
optimizer.  Weak scaling: global batch = 256*N.  Prints ONE JSON line on rank 0.

Extra objects in the JSON line:
  roofline     -- for the kernel class that takes the most time in the step: algorithmic bytes it must move
                  (its input and output tensors once, bf16 NHWC) / its launch time measured with
                  HIP events recorded around every launch of that class inside the timed region.
  cpu_baseline -- the CPU oracle (oracle/mnasnet_oracle.py, a port of the reference's eager path; the
                  reference sources do not travel to the GPU box) running the same step at bs=32 on the host
                  cores, rank 0, N=1 only.  A reported baseline, not the optimisation target.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
MFMA_PEAK_TFLOPS = 2500.0    # dense bf16 MFMA peak (no sparsity), MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (BASELINE: 256; --clusters: 64)")
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--hw", type=str, default="", help="rectangular input HxW (BASELINE config 5 clusters: 384x512, 512x512, "
                                                       "512x384); overrides --size")
    ap.add_argument("--k5", action="store_true", help="BASELINE config 4 (its 5x5 half): every MBConv stage with 5x5 depthwise convs "
                                                     "(the SE block of that config has no counterpart in the reference: not built)")
    ap.add_argument("--se", action="store_true", help="BASELINE config 4 in full: 5x5 depthwise convs AND a squeeze-excite block "
                                                     "(se_ratio 0.25, build-defined: the reference has none) in every MBConv_block")
    ap.add_argument("--ccf", action="store_true", help="the reference's DEFAULT topology Mnasnet(cut_channels_first=True) (mnasnet.py:176: stride-2 conv "
                                                      "first, blocks at the OUT width, 2.80 M parameters) instead of the one train.py "
                                                      "trains (classifiers.py:13: cut_channels_first=False); a variant line, never the headline")
    ap.add_argument("--clusters", action="store_true",
                    help="BASELINE config 5: rectangular-crop resolution clusters (384x512 / 512x512 / 512x384, datasets.py:331-335), "
                         "ONE cluster per step drawn by DistributedClusterSampler (cluster_random_sampler.py:31-55 made rank-aware), "
                         "the three compiled programs live at once.  Reports same-cluster-per-step img/s as `value` and, with "
                         "--gpus N > 1, the mixed-across-ranks rate (every rank of a step on a different cluster) next to it")
    ap.add_argument("--h2d", action="store_true", help="also report the PCIe-INCLUSIVE step rate (never `value`): pinned host batches, "
                                                      "double-buffered upload on a copy stream under the previous step, as fp32 "
                                                      "(what train.py:427 uploads) and as uint8 with the normalisation fused into the stem")
    ap.add_argument("--one-device", action="store_true",
                    help="FUNCTIONAL TEST ONLY (tests/test_gpu_world2.py): every rank on cuda:0, backend gloo on device tensors (RCCL "
                         "refuses two ranks on one device) -- exercises this script's multi-rank code path on a 1-GPU box; the line "
                         "says so in `overrides` and its numbers mean nothing")
    ap.add_argument("--force-distributed", action="store_true",
                    help="FUNCTIONAL TEST ONLY: with --gpus 1, run the N > 1 code path (process group on backend nccl = RCCL with "
                         "device_id, Trainer(distributed=True), barriers, the float64 MAX reduce, destroy) as a one-rank job; recorded in "
                         "`overrides`")
    ap.add_argument("--dump-state", type=str, default="",
                    help="TESTS: after the timed windows every rank writes <dir>/rank<r>.json (sha256 of its flat parameter buffer, "
                         "the bucket schedule of its last step, whether it ran the roofline calibration)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-box", action="store_true", help="skip the box calibration probes (copy GB/s, packed-FMA TFLOP/s, clocks)")
    ap.add_argument("--smi", action="store_true", help="also keep a rocm-smi / amd-smi snapshot in the box object: a CHILD process, started before this "
                                                      "process touches the GPU and never when a profiler / tool library is preloaded")
    ap.add_argument("--no-roofline", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--auto-graphs", type=float, default=0.92,
                    help="host-bound guard: if, after the warm-up, the host needs more than this share of a step's wall time to enqueue it "
                         "(MAX over ranks), the launch lists are replayed as hipGraphs (Engine.use_graphs: 0.6 ms of host time per step at "
                         "+0.5 %% GPU time) for the measurement; 0 disables the guard.  Reported as graph_mode.auto in the JSON line.")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--min-seconds", type=float, default=2.0, help="repeat the K-step timed window until this much timed work exists; "
                                                                   "the median window is reported")
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 64 if args.clusters else 256
    return args


def conv_table(engine, N, H, W):
    """Per ConvBlock application: (kind, E_in, E_out, macs) at batch N -- the algorithmic accounting of
    SURVEY 8(d) / Appendix A, derived from the module geometry."""
    rows = []
    Hc, Wc = H, W
    for op, m, stage in engine.steps:
        for cb in ([m] if op == "conv" else m):
            ci = engine.info[id(cb)]
            Ho, Wo = ci.out_hw(Hc, Wc)
            e_in = N * Hc * Wc * ci.cin
            e_out = N * Ho * Wo * ci.cout
            macs = N * Ho * Wo * ci.cout * (ci.cin // ci.groups) * ci.k * ci.k
            rows.append((ci.kind, ci, (Hc, Wc), e_in, e_out, macs))
            Hc, Wc = Ho, Wo
    return rows



def class_key(opc, ints, L):
    """Kernel-class name of one launch record (opcode + its integer fields)."""
    names = {L.OP_CONV_GEMM: "k_igemm", L.OP_CONV_WGRAD: "k_wgrad", L.OP_DW_FWD: "k_dw_conv<fwd>", L.OP_DW_BWD: "k_dw_bwd",
             L.OP_BN_BWD_REDUCE: "k_bn_bwd_reduce", L.OP_STEM_FWD: "k_igemm<stem>", L.OP_STEM_WGRAD: "k_wgrad<stem>",
             L.OP_ADD_ACT: "k_add_act", L.OP_PW_BWD: "k_pw_bwd",
             L.OP_POOL_ACT: "k_pool", L.OP_POOL_BWD: "k_pool", L.OP_DY_MAT: "k_dy_mat",
             L.OP_TCONV_DGRAD: "k_igemm<dgrad>", L.OP_SE_SCALE: "k_se", L.OP_SE_BWD_REDUCE: "k_se",
             L.OP_SE_BWD_APPLY: "k_se"}
    if opc == L.OP_CONV_GEMM:
        return "k_igemm<dgrad>" if ints[0] == 1 else "k_igemm<fwd>"
    if opc == L.OP_DW_BWD:          # i: N,H,W,C,k,nparts,phase (1 = input gradient, 2 = weight gradient launch)
        return "k_dw_conv<dgrad>" if ints[6] == 1 else ("k_dw_wgrad" if ints[6] == 2 else "k_dw_bwd")
    return names[opc]


def launch_work(opc, ints, L):
    """(algorithmic bytes, flops) of one launch: SURVEY 8(d) accounting -- each input and output tensor once, bf16."""
    if opc == L.OP_CONV_GEMM:       # i: mode,N,Hi,Wi,Ci,Ho,Wo,Co,...
        _, N_, Hi, Wi, Ci, Ho, Wo, Co = ints[:8]
        e_in, e_out = N_ * Hi * Wi * Ci, N_ * Ho * Wo * Co
        nbytes = 2 * ((2 * e_in + e_out) if ints[0] == 1 else (e_in + e_out))     # dgrad reads g AND y
        flops = 2.0 * N_ * Ho * Wo * Co * Ci * ints[8] * ints[9] if ints[0] == 0 else \
            2.0 * N_ * Hi * Wi * Ci * Co * ints[8] * ints[9]
    elif opc == L.OP_PW_BWD:        # i: M,Ci,Co,nparts: reads g, y (Co) and x (Ci), writes gin (Ci)
        M_, Ci, Co = ints[:3]
        nbytes = 2 * M_ * (2 * Co + 2 * Ci)
        flops = 2 * 2.0 * M_ * Co * Ci
    elif opc == L.OP_CONV_WGRAD:    # i: N,Hi,Wi,Ci,Ho,Wo,Co,kh,kw
        N_, Hi, Wi, Ci, Ho, Wo, Co = ints[:7]
        nbytes = 2 * (N_ * Hi * Wi * Ci + 2 * N_ * Ho * Wo * Co)
        flops = 2.0 * N_ * Ho * Wo * Co * Ci * ints[7] * ints[8]
    elif opc in (L.OP_DW_FWD, L.OP_DW_BWD):
        N_, H_, W_, C_, k_ = ints[:5]
        e = N_ * H_ * W_ * C_
        if opc == L.OP_DW_FWD:
            nbytes, flops = 2 * 2 * e, 2.0 * e * k_ * k_
        elif ints[6] == 0:
            nbytes, flops = 2 * 4 * e, 4.0 * e * k_ * k_
        else:       # each backward launch reads (g, y) and one more tensor / writes gin: 3 tensors
            nbytes, flops = 2 * 3 * e, 2.0 * e * k_ * k_
    elif opc == L.OP_TCONV_DGRAD:   # i: N,Ho,Wo,Co,Ci: stride-2 3x3 input gradient: reads g and y of the conv output, writes gin (2Ho x 2Wo)
        N_, Ho, Wo, Co, Ci = ints[:5]
        nbytes = 2 * (2 * N_ * Ho * Wo * Co + N_ * 4 * Ho * Wo * Ci)
        flops = 2.0 * N_ * Ho * Wo * Co * Ci * 9
    elif opc in (L.OP_BN_BWD_REDUCE, L.OP_ADD_ACT, L.OP_POOL_ACT, L.OP_POOL_BWD, L.OP_DY_MAT, L.OP_SE_SCALE,
                 L.OP_SE_BWD_REDUCE, L.OP_SE_BWD_APPLY):
        nbytes, flops = 0, 0.0      # pure overhead in SURVEY 8(d)'s accounting
    else:                           # stem fwd / wgrad: fp32 image + bf16 output
        N_, H_, W_, Ho, Wo, Co = ints[:6]
        nbytes = N_ * 3 * H_ * W_ * 4 + 2 * N_ * Ho * Wo * Co * (1 if opc == L.OP_STEM_FWD else 2)
        flops = 2.0 * N_ * Ho * Wo * Co * 27
    return nbytes, flops


def classify(profile, L):
    """Aggregate engine.read_profile() records per kernel class: {key: [ms, bytes, flops, launches]} + per-launch detail."""
    agg, detail = {}, []
    for (tag, opc, ints), msv in profile:
        key = class_key(opc, ints, L)
        nbytes, flops = launch_work(opc, ints, L)
        a = agg.setdefault(key, [0.0, 0.0, 0.0, 0])
        a[0] += msv; a[1] += nbytes; a[2] += flops; a[3] += 1
        detail.append((key, ints[:10], msv, nbytes))
    return {"agg": agg, "detail": detail}


def pmc_traffic(kernel_class):
    """HBM bytes per launch of a kernel class from the newest committed PMC summary (profiles/*_pmc_per_class.json, made by
    tools/profile_round.sh + tools/pmc_classes.py: FETCH_SIZE x2 + WRITE_SIZE in separate --pmc passes of this bench).
    Counters cannot be collected from inside a timed run; None if no summary is present."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_pmc_per_class.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            d = json.load(f)
        return round(d["classes"][kernel_class]["hbm_bytes_per_launch"]), "profiles/" + os.path.basename(files[-1])
    except (KeyError, ValueError, OSError):
        return None, None


def pmc_mfma_busy(kernel_class):
    """Matrix-pipe busy fraction of a kernel class from the newest committed counter summary (profiles/*_mfma_per_class.json:
    SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES), tools/pmc_sq.sh pass 3 + tools/mfma_summary.py)."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_mfma_per_class.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            d = json.load(f)
        return d["classes"][kernel_class]["mfma_busy_frac"], "profiles/" + os.path.basename(files[-1])
    except (KeyError, ValueError, OSError):
        return None, None


# ---- box calibration ------------------------------------------------------------------------------------------------------
# Boxes of this pool differ by up to 12 % in step rate at the same commit (round 4: builder 10.57 ms, driver 11.68 ms), and the
# per-class pattern (pure streaming kernels unchanged, instruction-heavy ones 12-18 % slower) says shader clock / power state.
# The "box" object records what THIS GPU sustains right before and right after the timed windows -- a float4 copy (HBM side)
# and a pure v_pk_fma_f32 loop (shader-clock side), csrc/mnas_probe.hip -- plus what sysfs / rocm-smi say about clocks and
# the power cap, so that two bench lines can be compared through their box ratios.
def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _sysfs_dir(dev):
    """/sys/bus/pci/devices/<bdf> of the torch device (None if it cannot be resolved: containers may hide sysfs)."""
    import glob
    try:
        pr = torch.cuda.get_device_properties(dev)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        d = "/sys/bus/pci/devices/" + bdf
        if os.path.isdir(d):
            return d
    except Exception:
        pass
    cards = [c for c in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")) if os.path.exists(c + "/pp_dpm_sclk")]
    return cards[0] if len(cards) == 1 else None


def sysfs_clocks(d):
    """Current sclk / mclk level (the starred line of pp_dpm_*), hwmon power and cap -- read while a probe kernel runs."""
    import glob
    if not d:
        return None
    out = {}
    for key, fn in (("sclk", "pp_dpm_sclk"), ("mclk", "pp_dpm_mclk"), ("fclk", "pp_dpm_fclk")):
        txt = _read(os.path.join(d, fn))
        if txt:
            cur = [ln for ln in txt.splitlines() if ln.rstrip().endswith("*")]
            out[key] = (cur[0] if cur else txt.splitlines()[-1]).split(":", 1)[-1].replace("*", "").strip()
            out[key + "_levels"] = len(txt.splitlines())
    for hw in glob.glob(os.path.join(d, "hwmon", "hwmon*")):
        for key, fn, scale in (("power_w", "power1_average", 1e-6), ("power_input_w", "power1_input", 1e-6),
                               ("power_cap_w", "power1_cap", 1e-6), ("power_cap_max_w", "power1_cap_max", 1e-6),
                               ("sclk_mhz_hwmon", "freq1_input", 1e-6), ("temp_c", "temp1_input", 1e-3)):
            v = _read(os.path.join(hw, fn))
            if v and v.lstrip("-").isdigit():
                out[key] = round(int(v) * scale, 1)
    v = _read(os.path.join(d, "power_dpm_force_performance_level"))
    if v:
        out["perf_level"] = v
    return out or None


def gpu_maybe_initialised():
    """True when this process may already have initialised the GPU -- torch says so, or a tool library that does it at load time
    (rocprofv3's, roctracer's, any LD_PRELOAD) is mapped.  A process in that state must not fork+exec on this pool."""
    if torch.cuda.is_initialized() or os.environ.get("LD_PRELOAD"):
        return True
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER", "ROCTX")) for k in os.environ):
        return True
    try:
        with open("/proc/self/maps") as f:
            maps = f.read().lower()
    except OSError:
        return True
    return any(t in maps for t in ("rocprofiler", "roctracer", "rocprof-sdk", "librocpd", "libroctx"))


def smi_snapshot():
    """rocm-smi / amd-smi clocks and power as a CHILD process (opt-in: --smi).  Only BEFORE this process touches the GPU (a
    process that has initialised HIP must not fork+exec on this pool) -- refused when gpu_maybe_initialised(); None if neither
    tool is present or it does not answer in time."""
    import shutil
    import subprocess
    if gpu_maybe_initialised():
        return {"skipped": "the GPU may already be initialised in this process (profiler / preload): no child process is started"}
    for tool, argv in (("rocm-smi", ["--showclocks", "--showpower", "--showmaxpower", "--showperflevel", "--json"]),
                       ("amd-smi", ["metric", "--clock", "--power", "--json"])):
        exe = shutil.which(tool) or ("/opt/rocm/bin/" + tool if os.path.exists("/opt/rocm/bin/" + tool) else None)
        if not exe:
            continue
        try:
            r = subprocess.run([exe] + argv, capture_output=True, text=True, timeout=30)
            txt = r.stdout.strip()
            try:
                return {"tool": tool, "data": json.loads(txt[txt.index("{"):]) if "{" in txt else txt[:2000]}
            except ValueError:
                return {"tool": tool, "data": txt[:2000]}
        except Exception as e:      # noqa: BLE001 -- measurement garnish, never fatal
            return {"tool": tool, "error": repr(e)[:200]}
    return None


class BoxProbe:
    """copy GB/s and packed-FMA TFLOP/s of this GPU (csrc/mnas_probe.hip through the C ABI), HIP-event timed on torch's current
    stream; sysfs clocks are read from the host WHILE a long packed-FMA launch is running (idle clocks say nothing)."""
    COPY_BYTES = 1 << 30
    VALU_BLOCKS, VALU_ITERS = 2048, 20000          # ~2 ms at 2.4 GHz: 8 waves per SIMD, 8 independent chains each

    def __init__(self, lib, dev):
        self.lib, self.dev = lib, dev
        self.sysfs = _sysfs_dir(dev)

    def _time(self, fn, reps):
        best = None
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1)
            best = ms if best is None else min(best, ms)
        return best

    def measure(self):
        """The 2 x 1 GiB of probe buffers exist only inside this call (nothing of the probe stays allocated during the timed windows)."""
        from mnasnet_pytorch_amd._lib import check, cur_stream
        lib, dev = self.lib, self.dev
        src = torch.empty(self.COPY_BYTES // 4, dtype=torch.float32, device=dev).fill_(1.0)
        dst = torch.empty_like(src)
        out = torch.zeros(self.VALU_BLOCKS * 256, dtype=torch.float32, device=dev)
        sink = torch.zeros(65536, dtype=torch.int32, device=dev)
        copy = lambda: check(lib.mnas_probe_copy(src.data_ptr(), dst.data_ptr(), self.COPY_BYTES, cur_stream()), "probe_copy")
        copy4 = lambda: check(lib.mnas_probe_copy4(src.data_ptr(), dst.data_ptr(), self.COPY_BYTES, 0, cur_stream()), "probe_copy4")
        read = lambda: check(lib.mnas_probe_read(src.data_ptr(), sink.data_ptr(), self.COPY_BYTES, 0, cur_stream()), "probe_read")
        valu = lambda it=self.VALU_ITERS: check(lib.mnas_probe_valu(out.data_ptr(), self.VALU_BLOCKS, it, cur_stream()), "probe_valu")
        copy(); copy4(); read(); valu()
        torch.cuda.synchronize()
        cms, c4ms, rms = self._time(copy, 5), self._time(copy4, 5), self._time(read, 5)
        vms = self._time(valu, 5)
        flop = self.VALU_BLOCKS * 256.0 * self.VALU_ITERS * 8 * 4
        res = {"copy_GBps": round(2 * self.COPY_BYTES / cms / 1e6, 1), "copy4_nt_GBps": round(2 * self.COPY_BYTES / c4ms / 1e6, 1),
               "read_GBps": round(self.COPY_BYTES / rms / 1e6, 1), "valu_pk_fma_TFLOPps": round(flop / vms / 1e9, 2),
               "valu_clock_GHz": round(flop / vms / 1e9 / 65.536, 3)}
        if self.sysfs:
            valu(self.VALU_ITERS * 25)              # ~50 ms of pure vector load: sample the clocks from the host meanwhile
            time.sleep(0.02)
            res["under_valu_load"] = sysfs_clocks(self.sysfs)
        torch.cuda.synchronize()
        del src, dst, out, sink
        torch.cuda.empty_cache()
        return res


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


def cpu_baseline(seconds, kernel=None, se_ratio=0.0):
    """The oracle's train step on the host cores (fp32, bs=32, head '512', Adam) -- kind 'port'."""
    from oracle import mnasnet_oracle as O
    ncpu = os.cpu_count() or 1
    net = O.OracleNet(ccf=False, head="512", num_classes=1000, seed=1, kernel=kernel, se_ratio=se_ratio).train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    bs = 32
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(bs, 3, 224, 224, generator=g)
    t = torch.randint(0, 1000, (bs,), generator=g)
    # thread count: oneDNN stops scaling on these small convs well before the host's core count and then LOSES (EPYC 9575F,
    # bs 32: 16 threads 1.42 s/step, 32: 1.54, 64: 2.57, 128: 4.9, measured in round 5): one step at each candidate,
    # the fastest runs the sample, so that the baseline is the best this host does with the port
    cands = sorted({min(ncpu, c) for c in (64, 32, 16)}, reverse=True)
    torch.set_num_threads(cands[0])
    O.train_step(net, opt, x, t)       # warm-up
    best = (None, cands[0])
    for c in cands:
        torch.set_num_threads(c)
        t0 = time.perf_counter()
        O.train_step(net, opt, x, t)
        el = time.perf_counter() - t0
        if best[0] is None or el < best[0]:
            best = (el, c)
    threads = best[1]
    torch.set_num_threads(threads)
    n, t0 = 0, time.perf_counter()
    while True:
        O.train_step(net, opt, x, t)
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds or n >= 20:
            break
    return {"value": round(bs * n / el, 2), "unit": "images/sec", "cores": threads, "cpu_model": cpu_model(),
            "host_cpus": os.cpu_count(), "kind": "port",
            "sample": "%d train steps of bs=%d fp32 (same model/head/optimizer, oracle/mnasnet_oracle.py) in %.1f s"
                      % (n, bs, el)}


def h2d_mode(trainer, model, x_dev, target, steps, dev):
    """Step rate with the batch coming from the host every step (the reference hands a host tensor to the step: train.py:427
    ``input.float().to(device)``).  Two pinned host buffers and two device buffers; the upload of batch i+1 runs on a copy
    stream under step i.  Reported next to, never as, the resident-input `value`."""
    out = {}
    main = torch.cuda.current_stream(dev)
    copy = torch.cuda.Stream(device=dev)
    for kind in ("f32", "u8"):
        if kind == "u8":
            model.normalize_on_device()          # uint8 images: normalisation fused into the stem conv's load
            host = [torch.randint(0, 256, tuple(x_dev.shape), dtype=torch.uint8).pin_memory() for _ in range(2)]
        else:
            host = [torch.randn(tuple(x_dev.shape)).pin_memory() for _ in range(2)]
        devb = [torch.empty(tuple(x_dev.shape), dtype=host[0].dtype, device=dev) for _ in range(2)]
        ready = [torch.cuda.Event() for _ in range(2)]
        done = [torch.cuda.Event() for _ in range(2)]

        def upload(i):
            with torch.cuda.stream(copy):
                copy.wait_event(done[i % 2])                     # the step that read this buffer has finished
                devb[i % 2].copy_(host[i % 2], non_blocking=True)
                ready[i % 2].record(copy)
        for e in done:
            e.record(main)
        for it in range(3):                                      # warm-up (program build for the uint8 stem, allocator)
            upload(it)
            main.wait_event(ready[it % 2])
            trainer.step(devb[it % 2], target)
            done[it % 2].record(main)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        upload(0)
        for it in range(steps):
            if it + 1 < steps:
                upload(it + 1)
            main.wait_event(ready[it % 2])
            trainer.step(devb[it % 2], target)
            done[it % 2].record(main)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        mb = host[0].numel() * host[0].element_size() / 1e6
        out[kind] = {"ms_per_step": round(dt / steps * 1e3, 3), "images_per_sec": round(x_dev.shape[0] * steps / dt, 1),
                     "host_batch_MB": round(mb, 1)}
        if kind == "u8":
            model.normalize_on_device(False)
    out["note"] = ("pinned host batch uploaded every step, double-buffered on a copy stream under the previous step; f32 = the "
                   "reference's input.float().to(device) (train.py:427), u8 = raw uint8 images with Normalize(mean, std) fused "
                   "into the stem conv's load (FineTuneModelPool.normalize_on_device)")
    return out


CLUSTER_SHAPES = ((384, 512), (512, 512), (512, 384))        # (H, W) of clusters 0, 1, 2: datasets.py:331-335


def clusters_mode(args, trainer, dev, rank, world, barrier, overrides):
    """BASELINE configs[4] / SURVEY 8(d): "per step one shape from {384x512, 512x512, 512x384}; report both
    same-cluster-per-step and mixed-across-ranks".  The step schedule comes from DistributedClusterSampler over a synthetic
    dataset with three equally sized clusters (identical on every rank); a step's batch is the resident synthetic batch of its
    cluster's shape.  same-cluster: every rank runs the schedule's cluster (what the sampler guarantees); mixed-across-ranks:
    rank r runs the cluster of schedule position s + r, so the ranks of one step disagree (the straggler case the sampler
    exists to avoid).  Timed as the main bench: exactly --steps steps between barrier + synchronize, MAX over ranks, repeated
    until --min-seconds, median window."""
    import types
    import torch
    from mnasnet_pytorch_amd.sampler import DistributedClusterSampler
    B, K = args.batch, args.steps
    per_cluster = ((K + 2) // 3 + 1) * B * world
    ds = types.SimpleNamespace(cluster_indices=[list(range(c * per_cluster, (c + 1) * per_cluster)) for c in range(3)])
    sampler = DistributedClusterSampler(ds, B, num_replicas=world, rank=rank, shuffle=True, seed=0)
    sampler.set_epoch(0)
    sched = sampler.cluster_of_step()[:K]
    assert len(sched) == K and set(sched) <= {0, 1, 2}
    g = torch.Generator(device=dev).manual_seed(4321 + rank)
    xs = [torch.randn(B, 3, h, w, device=dev, generator=g) for h, w in CLUSTER_SHAPES]
    target = torch.randint(0, 1000, (B,), device=dev, generator=g)
    for _ in range(max(1, args.warmup // 3)):                # builds the three programs; all stay alive
        for c in range(3):
            trainer.step(xs[c], target)
    barrier()
    eng = trainer.engine
    live = sorted({k[1:3] for k, lst in eng.programs.items() if lst})

    def run(order):
        windows = []
        while True:
            barrier()
            t0 = time.perf_counter()
            for c in order:
                trainer.step(xs[c], target)
            barrier()
            dt = time.perf_counter() - t0
            if world > 1:
                import torch.distributed as dist
                tt = torch.tensor([dt], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dt = float(tt)
            windows.append(dt)
            if sum(windows) >= args.min_seconds or len(windows) >= 50:
                break
        windows.sort()
        dt = windows[(len(windows) - 1) // 2]
        return {"images_per_sec": round(B * world * K / dt, 1), "ms_per_step": round(dt / K * 1e3, 3), "windows": len(windows)}

    same = run(sched)
    mixed = run([sched[(s + rank) % K] for s in range(K)]) if world > 1 else None
    per_shape = {}
    for c, (h, w) in enumerate(CLUSTER_SHAPES):             # one shape per run, for reference (what --hw measures)
        r = run([c] * K)
        per_shape["%dx%d" % (h, w)] = {"images_per_sec": r["images_per_sec"], "ms_per_step": r["ms_per_step"]}
    if rank != 0:
        return None
    return {
        "metric": "images/sec MNASNet-1.0 rectangular-crop clusters bf16 train step", "value": same["images_per_sec"],
        "unit": "images/sec", "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": same["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "overrides": overrides,
        "config": {"workload": "MNASNet-1.0 (Mnasnet(cut_channels_first=False)+head '512', 1000 classes) fwd+bwd+Adam, bs=%d/GPU, "
                               "one resolution cluster per step from {384x512, 512x512, 512x384} (DistributedClusterSampler), "
                               "per-rank BatchNorm" % B,
                   "global_batch": B * world, "parallelism": "dp%d" % world,
                   "steps_per_cluster": {"%dx%d" % CLUSTER_SHAPES[c]: sched.count(c) for c in range(3)},
                   "programs_live": ["%dx%d" % hw for hw in live]},
        "clusters": {"same_cluster_per_step": same, "mixed_across_ranks": mixed,
                     "mixed_note": None if world > 1 else "one rank: identical to same_cluster_per_step by construction",
                     "single_shape_runs": per_shape},
    }


# Diagnosis switches: MNAS_* environment variables that change how the engine compiles the step.  They exist for same-call A/B
# measurements only; every one that is honoured is listed in the JSON line ("overrides") so that a number measured with one
# can never pass for the default configuration.
def _ov_int(name):
    return int(os.environ[name])


OVERRIDES = [
    # (variable, what it does, setter(eng))
    ("MNAS_NO_SIDE", "weight-gradient kernels on the main stream (the default)", lambda e: setattr(e, "use_side_stream", False)),
    ("MNAS_SIDE", "weight-gradient kernels on a second stream (default until round 3)", lambda e: setattr(e, "use_side_stream", True)),
    ("MNAS_SIDE_MAXPX", "with MNAS_SIDE: only layers of at most this many output pixels use the second stream",
     lambda e: setattr(e, "side_stream_max_pixels", _ov_int("MNAS_SIDE_MAXPX"))),
    ("MNAS_PW_SPLIT_MAX", "pixel count below which project convs use dgrad + wgrad kernels",
     lambda e: setattr(e, "pw_split_max_pixels", _ov_int("MNAS_PW_SPLIT_MAX"))),
    ("MNAS_PW_FUSED_MIN", "pixel count from which 1x1 convs use the fused backward",
     lambda e: setattr(e, "pw_fused_min_pixels", _ov_int("MNAS_PW_FUSED_MIN"))),
    ("MNAS_NO_TCONV", "stride-2 3x3 input gradients through k_igemm's parity-class form", lambda e: setattr(e, "use_tconv", False)),
    ("MNAS_NO_MERGE", "separate finalize launches", lambda e: setattr(e, "merge_post", False)),
    ("MNAS_PWB", "large,mid,small persistent grids of the fused 1x1 backward",
     lambda e: [setattr(e, n, int(v)) for n, v in zip(("pw_bwd_parts_large", "pw_bwd_parts_mid", "pw_bwd_parts_small"),
                                                      os.environ["MNAS_PWB"].split(","))]),
    ("MNAS_SE_MLP_UNFUSED", "squeeze-excite MLP as separate head-GEMM launches", lambda e: setattr(e, "se_fused_mlp", False)),
    ("MNAS_GRAPHS", "launch lists replayed as hipGraphs", lambda e: setattr(e, "use_graphs", True)),
    ("MNAS_NO_DYMAT", "dense 3x3 backward forms dy on load", lambda e: setattr(e, "materialize_dy", False)),
    ("MNAS_WGRAD_WGS", "workgroups per k_wgrad launch", lambda e: setattr(e, "wgrad_wgs", _ov_int("MNAS_WGRAD_WGS"))),
    ("MNAS_NO_RECOMP", "expand convs' fused backward reads the stored y1 instead of recomputing it", lambda e: setattr(e, "pw_recompute_y", False)),
    ("MNAS_NO_MASKED_G", "project convs store the unmasked input gradient, the depthwise backward derives the ReLU mask per window column",
     lambda e: setattr(e, "dw_masked_g", False)),
    ("MNAS_DW5_SPLIT", "two-launch backward for the 5x5 depthwise layers", lambda e: setattr(e, "dw_fused_k", (3,))),
    ("MNAS_NO_SE_ONLOAD", "squeeze-excite through the materialised a*s tensor (k_se_scale) instead of on load", lambda e: setattr(e, "se_on_load", not _ov_int("MNAS_NO_SE_ONLOAD"))),
    ("MNAS_PWB_SEGMENTS", "fused 1x1 backward over contiguous pixel segments with at most this many workgroups", lambda e: setattr(e, "pw_bwd_segments", _ov_int("MNAS_PWB_SEGMENTS"))),
    ("MNAS_IGF_PARTS", "upper bound on the persistent workgroups of a k_igemm forward launch", lambda e: setattr(e, "igemm_fwd_parts", _ov_int("MNAS_IGF_PARTS"))),
    ("MNAS_IGD_PARTS", "upper bound on the persistent workgroups of a k_igemm input-gradient launch", lambda e: setattr(e, "igemm_dgrad_parts", _ov_int("MNAS_IGD_PARTS"))),
    ("MNAS_DWB_PARTS", "upper bound on the persistent workgroups of a depthwise backward launch", lambda e: setattr(e, "dw_bwd_parts", _ov_int("MNAS_DWB_PARTS"))),
    ("MNAS_LIB_PATH", "alternative build of libmnas_hip.so (tools/build_alt.sh)", lambda e: None),
]


def apply_overrides(eng):
    used = []
    for name, what, setter in OVERRIDES:
        if os.environ.get(name):
            setter(eng)
            used.append("%s=%s (%s)" % (name, os.environ[name], what))
    # switches the DIAGNOSIS build of the library reads itself (tools/build_alt.sh, -DMNAS_DIAG); the shipped library ignores them
    if os.environ.get("MNAS_LIB_PATH"):
        known = {n for n, _, _ in OVERRIDES}
        used += ["%s=%s (library diagnosis switch)" % (k, v) for k, v in sorted(os.environ.items())
                 if k.startswith("MNAS_") and k not in known and k != "MNAS_BENCH_DETAIL"]
    return used


def main():
    args = parse()
    # multi-process GPU work on this pool needs dmabuf IPC (the host driver has no legacy IPC): RCCL and cross-process tensor sharing
    # fail with hipIpcGetMemHandle otherwise.  The driver exports it; keep it for the ranks this script starts itself.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks as a CHILD torch.distributed.run (nothing has touched the GPU
        # in this process yet; never exec from a process that has) and hand its exit code back
        import socket
        import subprocess
        if not args.one_device and torch.cuda.device_count() < args.gpus:
            sys.exit("bench.py: --gpus %d needs %d visible GPUs, this box has %d. Nothing was measured. (--one-device runs the "
                     "multi-rank code path on cuda:0 over gloo as a functional test.)" % (args.gpus, args.gpus, torch.cuda.device_count()))
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d, or run "
                 "plain `python bench.py --gpus %d`)" % (args.gpus, world, args.gpus, args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or args.force_distributed
    if args.force_distributed and "RANK" not in os.environ:       # not under a launcher: a port of our own (an inherited MASTER_PORT
        import socket                                              # may belong to a live process group of the parent process)
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
    # opt-in child process, BEFORE anything in this process touches the GPU and never with a tool library mapped (smi_snapshot
    # refuses by itself: rocprofv3's preloaded library has initialised the GPU before main() runs)
    smi = smi_snapshot() if (rank == 0 and args.smi and not args.no_box) else None
    if not args.one_device:
        ndev = torch.cuda.device_count()             # counts devices without initialising the GPU
        if ndev < max(1, (args.gpus if distributed else 1)) or local_rank >= max(ndev, 1):
            sys.exit("bench.py: --gpus %d needs %d visible GPUs, this box has %d (rank %d, LOCAL_RANK %d). Nothing was measured. "
                     "(--one-device runs the multi-rank code path on cuda:0 over gloo as a functional test.)"
                     % (args.gpus, args.gpus, ndev, rank, local_rank))
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.one_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from mnasnet_pytorch_amd import FineTuneModelPool, load_model, _lib as L
    from mnasnet_pytorch_amd.train_step import Trainer

    torch.manual_seed(0)                       # identical init on every rank (and rank-0 broadcast in Trainer)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        base = load_model("mnasnet")
    if args.ccf:
        from mnasnet_pytorch_amd import Mnasnet
        base = Mnasnet(cut_channels_first=True)
    elif args.se:
        from mnasnet_pytorch_amd import Mnasnet
        base = Mnasnet(False, kernel_size=5, se_ratio=0.25)
    elif args.k5:
        from mnasnet_pytorch_amd.mnasnet import ConvBlock, MBConv, SepConv, _Features
        cfg = [(16, 24, 3, 3, True), (24, 40, 3, 3, True), (40, 80, 6, 3, True), (80, 96, 6, 2, False), (96, 192, 6, 4, True),
               (192, 320, 6, 1, False)]
        base.features = _Features(ConvBlock(3, 32, kernel_size=3, stride=2, padding=1), SepConv(32, 16, kernel_size=3),
                                  *[MBConv(i, o, channel_factor=t, layers=n, kernel_size=5, reduce=r, cut_channels_first=False)
                                    for i, o, t, n, r in cfg])
        base.init_params()
    model = FineTuneModelPool(base, "mnasnet", 1000, "512").to(dev).train()
    trainer = Trainer(model, lr=1e-3, distributed=distributed)
    eng = trainer.engine
    overrides = apply_overrides(eng)          # diagnosis switches (environment); every honoured one is echoed in the JSON line
    if args.one_device:
        overrides.append("--one-device (functional test of the multi-rank path: all ranks on cuda:0 over gloo; NOT a measurement)")
    if args.force_distributed:
        overrides.append("--force-distributed (functional test: the multi-rank code path as a one-rank RCCL job)")
    profile = (not args.no_roofline) and rank == 0
    ALL_OPS = {L.OP_CONV_GEMM, L.OP_CONV_WGRAD, L.OP_DW_FWD, L.OP_DW_BWD, L.OP_BN_BWD_REDUCE, L.OP_STEM_FWD, L.OP_STEM_WGRAD,
               L.OP_ADD_ACT, L.OP_PW_BWD, L.OP_POOL_ACT, L.OP_POOL_BWD, L.OP_DY_MAT, L.OP_TCONV_DGRAD,
               L.OP_SE_SCALE, L.OP_SE_BWD_REDUCE, L.OP_SE_BWD_APPLY}

    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    B, S = args.batch, args.size
    Hh, Ww = (int(v) for v in args.hw.lower().split("x")) if args.hw else (S, S)
    x = torch.randn(B, 3, Hh, Ww, device=dev, generator=g)
    target = torch.randint(0, 1000, (B,), device=dev, generator=g)

    def barrier():
        if distributed:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    if args.clusters:
        del x
        res = clusters_mode(args, trainer, dev, rank, world, barrier, overrides)
        if res is not None:
            print(json.dumps(res))
        if distributed:
            import torch.distributed as dist
            dist.destroy_process_group()
        return
    # ---- box calibration, first half: BEFORE the warm-up steps (round 6: it used to run right in front of the timed windows --
    # 75 ms of full vector load and 12 GiB of copies, on rank 0 only, changed the thermal / clock state the windows were measured
    # in); every rank runs it (the MAX-over-ranks time must not depend on which rank was heated), rank 0 reports
    box, probe = None, None
    if not args.no_box:
        probe = BoxProbe(L.load(), dev)
        box = {"before": probe.measure()}
    for _ in range(args.warmup):
        trainer.step(x, target)
    barrier()
    # ---- host-bound guard: a step is ~400 launches enqueued by one Python thread (6.4-7.4 ms against 9.9 ms of GPU time on the
    # hosts seen so far); on a slower or busier host (8 ranks on one node) the enqueue could become the step.  Measured here, on
    # every rank, decided on the MAX over ranks: above --auto-graphs the launch lists are replayed as hipGraphs instead.
    auto_graph = None
    if args.auto_graphs > 0 and not eng.use_graphs:
        t0 = time.perf_counter()
        for _ in range(8):
            trainer.step(x, target)
        h_dt = time.perf_counter() - t0
        torch.cuda.synchronize()
        share = h_dt / max(time.perf_counter() - t0, 1e-9)
        if distributed:
            import torch.distributed as dist
            tt = torch.tensor([share], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            share = float(tt)
        auto_graph = {"host_share_of_step": round(share, 3), "threshold": args.auto_graphs, "switched": False}
        if share > args.auto_graphs:
            try:
                eng.use_graphs = True
                eng.reset_programs()
                for _ in range(2):
                    trainer.step(x, target)
                auto_graph["switched"] = True
            except Exception as e:                # stay on the per-launch path rather than lose the measurement
                eng.use_graphs = False
                eng.reset_programs()
                auto_graph["error"] = repr(e)
        barrier()
    # ---- roofline calibration (untimed, rank 0): bracket EVERY conv-kernel launch with HIP events for two steps, find the
    # kernel class with the largest time, then bracket only that class inside the timed region (bracketing all ~190
    # launches costs 7 % of the step rate; one class costs < 1 %)
    calib, dom_key = None, None
    if not args.no_roofline:                 # every rank runs the same extra steps (they contain collectives)
        if profile:
            eng.profile_opcodes = ALL_OPS
            eng.reset_programs()
        for _ in range(2):
            trainer.step(x, target)
        barrier()
        if profile:
            calib = classify(eng.read_profile(), L)
            dom_key = max(calib["agg"].items(), key=lambda kv: kv[1][0])[0]
            eng.profile_filter = lambda opc, ints: class_key(opc, ints, L) == dom_key
            eng.reset_programs()
        if profile:
            eng.profile_gate.value = 0               # as in all but the last step of a window; with MNAS_GRAPHS this is where the
        for _ in range(2):                           # launch lists are captured and instantiated -- outside the timed windows
            trainer.step(x, target)
    # ---- timed region: windows of EXACTLY --steps steps, each bracketed by barrier + synchronize on both sides and reduced with
    # MAX over ranks; windows are repeated until >= --min-seconds of timed work exist (a 20-step window is 0.2 s: too short for
    # an external GPU-busy sampler to see) and the MEDIAN window is reported
    windows, host_dts = [], []
    while True:
        barrier()
        t0 = time.perf_counter()
        for i_ in range(args.steps):
            if profile:                             # the dominant class is bracketed in the window's LAST step only: an event
                eng.profile_gate.value = 1 if i_ == args.steps - 1 else 0      # record costs ~3 us of dispatch gap per side
            loss = trainer.step(x, target)
        host_dt = time.perf_counter() - t0          # host-side enqueue time (no sync inside step)
        barrier()
        dt = time.perf_counter() - t0
        if distributed:
            import torch.distributed as dist
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt)
        windows.append(dt)
        host_dts.append(host_dt)
        if sum(windows) >= args.min_seconds or len(windows) >= 200:     # same decision on every rank (dt is the all-reduced MAX)
            break
    if probe is not None:
        box["after"] = probe.measure()
        pr = torch.cuda.get_device_properties(dev)
        box.update({"device": pr.name, "gcn_arch": getattr(pr, "gcnArchName", None), "cus": pr.multi_processor_count,
                    "hbm_GiB": round(pr.total_memory / 2 ** 30, 1), "torch_clock_rate_khz": getattr(pr, "clock_rate", None),
                    "smi_before_gpu_init": smi,
                    "note": "copy_GBps = 2 x 1 GiB / plain grid-stride float4 copy (ONE 16-byte load in flight per lane); "
                            "copy4_nt_GBps = the same bytes with four independent nontemporal 16-byte loads in flight per lane and "
                            "nontemporal stores (how the step's streaming kernels are built: the rate they may be compared with); "
                            "read_GBps = 1 GiB read-only stream, four loads in flight; valu_pk_fma_TFLOPps = pure v_pk_fma_f32 loop, "
                            "valu_clock_GHz = that / 65.536 flop per clock (256 CUs x 4 SIMDs x 16 lanes x 2 x 2); best of 5, HIP "
                            "events; 'before' is taken ahead of the warm-up steps, 'after' right after the timed windows, on every "
                            "rank (rank 0 reported); no probe buffer is allocated during the windows (csrc/mnas_probe.hip)"})
    if args.dump_state:
        import hashlib
        torch.cuda.synchronize()
        os.makedirs(args.dump_state, exist_ok=True)
        with open(os.path.join(args.dump_state, "rank%d.json" % rank), "w") as f:
            json.dump({"rank": rank, "world": world, "local_rank": local_rank, "profiled": bool(profile),
                       "flat_p_sha256": hashlib.sha256(trainer.flat_p.detach().cpu().numpy().tobytes()).hexdigest(),
                       "schedule": [list(e) for e in trainer.schedule.log] if trainer.schedule is not None else None,
                       "steps_run": trainer.steps if hasattr(trainer, "steps") else None,
                       "running_mean0_sha256": hashlib.sha256(model.state_dict()[next(k for k in model.state_dict() if k.endswith("running_mean"))]
                                                              .detach().cpu().numpy().tobytes()).hexdigest()}, f)
    order = sorted(range(len(windows)), key=lambda i: windows[i])
    mid = order[(len(order) - 1) // 2]
    dt, host_dt = windows[mid], host_dts[mid]
    lossv = float(loss)
    if rank != 0:
        if distributed:
            import torch.distributed as dist
            dist.destroy_process_group()
        return

    ms = dt / args.steps * 1e3
    value = B * world * args.steps / dt
    res = {
        "metric": "images/sec MNASNet-1.0 224^2 bf16 train step",
        "value": round(value, 1), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "windows": len(windows), "timed_seconds": round(sum(windows), 3),
        "window_ms_per_step": [round(w / args.steps * 1e3, 3) for w in windows[:16]],
        "ms_per_step": round(ms, 3), "host_enqueue_ms_per_step": round(host_dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic", "overrides": overrides,
        "config": {"workload": "MNASNet-1.0 (Mnasnet(cut_channels_first=%s)+head '512', 1000 classes%s) fwd+bwd+Adam, "
                               "bs=%d/GPU, %dx%d, per-rank BatchNorm" % (bool(args.ccf), ", all depthwise convs 5x5 + squeeze-excite (build-defined)" if args.se else (", all depthwise convs 5x5" if args.k5 else ""), B, Hh, Ww),
                   "global_batch": B * world, "parallelism": "dp%d" % world, "loss": round(lossv, 4)},
    }
    # ---- roofline of the dominant kernel class: events recorded inside the timed region (last timed step) -----------
    if profile:
        timed = classify(eng.read_profile(), L)
        tms, nb, fl, cnt = timed["agg"][dom_key]
        tot = sum(a[0] for a in calib["agg"].values())
        gbs = nb / (tms * 1e-3) / 1e9 if tms > 0 else 0.0
        traffic, traffic_src = pmc_traffic(dom_key)
        mfma_busy, mfma_src = pmc_mfma_busy(dom_key)
        # the binding roofline follows from the class's arithmetic intensity against the bf16 ridge (2500 TFLOP/s / 8 TB/s = 312
        # flop/B).  Every class of this path sits left of it (AI 4-165 flop/B) -> "hbm"; the matrix-pipe busy
        # fraction from the SQ counters is reported next to it either way
        ai = fl / nb if nb > 0 else 0.0
        tfl = fl / (tms * 1e-3) / 1e12 if tms > 0 else 0.0
        mfma_bound = ai > MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
        res["roofline"] = {"kernel": dom_key, "bound": "mfma" if mfma_bound else "hbm",
                           "achieved": round(tfl, 1) if mfma_bound else round(gbs, 1),
                           "peak": MFMA_PEAK_TFLOPS if mfma_bound else HBM_PEAK_GBS, "unit": "TFLOP/s" if mfma_bound else "GB/s",
                           "frac": round(tfl / MFMA_PEAK_TFLOPS, 4) if mfma_bound else round(gbs / HBM_PEAK_GBS, 4),
                           "arithmetic_intensity_flop_per_byte": round(ai, 1),
                           "traffic": traffic, "traffic_source": traffic_src,
                           "mfma_busy": mfma_busy, "mfma_busy_source": mfma_src,
                           "mfma_achieved_TFLOPps": round(fl / (tms * 1e-3) / 1e12, 1) if tms > 0 else 0.0,
                           "mfma_peak_TFLOPps": MFMA_PEAK_TFLOPS,
                           "algorithmic_bytes_per_launch": round(nb / cnt),
                           "launches_per_step": cnt, "avg_launch_us": round(tms / cnt * 1e3, 2),
                           "ms_per_step": round(tms, 3),
                           "share_of_all_conv_kernel_time": round(calib["agg"][dom_key][0] / tot, 3),
                           "note": "HIP events around every launch of this kernel class inside the timed region (the LAST step of "
                                   "every timed window; the records are gated off in the other steps -- always on they cost 1 %% of "
                                   "the step rate)%s; algorithmic bytes per SURVEY 8(d) (inputs + outputs once, bf16)."
                                   % (". Weight-gradient kernels run on a second stream concurrently with this chain, so a launch's "
                                      "duration includes the bandwidth it shares with its neighbour" if eng.use_side_stream else
                                      ", one stream")}
        res["kernel_classes"] = {k: {"ms_per_step": round(v[0], 3), "algorithmic_GB": round(v[1] / 1e9, 3),
                                     "GBps": round(v[1] / max(v[0], 1e-9) / 1e6, 1),
                                     "hbm_frac": round(v[1] / max(v[0], 1e-9) / 1e6 / HBM_PEAK_GBS, 3),
                                     "TFLOP": round(v[2] / 1e12, 4), "TFLOPps": round(v[2] / max(v[0], 1e-9) / 1e9, 1),
                                     "mfma_frac": round(v[2] / max(v[0], 1e-9) / 1e9 / MFMA_PEAK_TFLOPS, 4), "launches": v[3]}
                                 for k, v in sorted(calib["agg"].items(), key=lambda kv: -kv[1][0])}
        res["kernel_classes_note"] = ("untimed calibration step with every conv-kernel launch bracketed (%s); "
                                      % ("two streams overlapping" if eng.use_side_stream else "one stream")
                                      + "hbm_frac = algorithmic GB/s / 8 TB/s, mfma_frac = algorithmic TFLOP/s / 2500 (dense bf16): every "
                                      "class is HBM-bound unfused (SURVEY 8(d)), mfma_frac is reported for completeness")
        res["bracketed_ms_per_step"] = round(tot, 3)
        if os.environ.get("MNAS_BENCH_DETAIL"):
            for key, ints, msv, nb_ in calib["detail"]:
                sys.stderr.write("%-18s %-48s %8.1f us %8.1f GB/s\n" % (key, ",".join(map(str, ints)), msv * 1e3,
                                                                      nb_ / max(msv, 1e-9) / 1e6))
    if auto_graph is not None:
        res["host_bound_guard"] = auto_graph
    if eng.use_graphs:
        res["graph_mode"] = {"captured_before_timed_region": True, "auto": bool(auto_graph and auto_graph.get("switched")),
                             "per_launch_steps_per_window": 1 if profile else 0,
                             "note": "launch lists replayed as hipGraphs; the graphs are captured in an untimed step before the first "
                                     "window" + ("; the LAST step of every %d-step window runs per-launch (its dominant-class "
                                                 "launches are bracketed with HIP events for the roofline object; --no-roofline "
                                                 "gives a pure replay measurement)" % args.steps if profile else "")}
    if box is not None:
        res["box"] = box
    if args.h2d and world == 1:
        res["pcie_inclusive"] = h2d_mode(trainer, model, x, target, args.steps, dev)
    if world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(args.cpu_seconds, 5 if (args.k5 or args.se) else None, 0.25 if args.se else 0.0)
    print(json.dumps(res))
    if distributed:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
