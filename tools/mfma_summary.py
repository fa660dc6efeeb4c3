#!/usr/bin/env python3
"""Per-kernel and per-class matrix-core tables from pass 3 of tools/pmc_sq.sh (+ the kernel trace for durations).
usage: mfma_summary.py pmc_<tag>_3.txt [prof_<tag>_kernel_trace.txt] [--json profiles/<tag>_mfma_per_class.json] > profiles/<tag>_mfma_busy.txt

mfma_busy% = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES): the share of the time a CU had work during which its
matrix pipes were executing (16 busy cycles per v_mfma_f32_16x16x32_bf16 per SIMD, MI355X_MICROARCH.md cycle table).
TFLOP/s = 16384 flop x SQ_INSTS_MFMA / kernel time (when a kernel trace is given).  The per-class JSON (bench.py's kernel
classes, tools/pmc_classes.py::key) is what bench.py reports as roofline.mfma_busy."""
import collections
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_classes import key  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
jpath = None
if "--json" in sys.argv:
    jpath = sys.argv[sys.argv.index("--json") + 1]
    args = [a for a in args if a != jpath]
d = collections.defaultdict(dict)
for l in open(args[0]):
    if l.startswith('#') or l.startswith('kernel'):
        continue
    m = re.match(r'(.{86}) (\S+)\s+([\d.]+)\s+([\d.]+)', l)
    if not m:
        continue
    k, c = m.group(1).strip(), m.group(2)
    d[k][c] = float(m.group(4))
    d[k]['calls'] = float(m.group(3))
dur = {}
if len(args) > 1:
    for l in open(args[1]):
        m = re.match(r'(.{86})\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)', l)
        if m:
            dur[m.group(1).strip()] = (float(m.group(2)), float(m.group(4)))       # calls per step, avg us
print("# matrix-core counters per kernel, per bench step (sums over 4 steps / 4); mfma_busy% = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES)")
print("%-60s %8s %12s %14s %14s %10s %9s" % ("kernel", "calls", "mfma_insts", "mfma_busy_cy", "cu_busy_cy", "mfma_busy%", "TFLOP/s"))
rows = sorted(((c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0), k, c) for k, c in d.items()), reverse=True)
cls = collections.defaultdict(lambda: [0.0, 0.0, 0.0])
for _, k, c in rows:
    busy, cu = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0), c.get('SQ_BUSY_CU_CYCLES', 0.0)
    n = c.get('SQ_INSTS_MFMA', 0.0)
    a = cls[key(k)]
    a[0] += busy; a[1] += cu; a[2] += n
for _, k, c in rows[:48]:
    busy, cu = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0), max(c.get('SQ_BUSY_CU_CYCLES', 0.0), 1.0)
    n = c.get('SQ_INSTS_MFMA', 0.0)
    tf = ""
    if k in dur and dur[k][1] > 0:
        tf = "%.1f" % (16384.0 * n / (dur[k][0] * dur[k][1] * 1e-6) / 1e12)
    print("%-60s %8.0f %12.0f %14.0f %14.0f %10.2f %9s" % (k[:60], c['calls'], n, busy, cu, 100.0 * busy / (4.0 * cu), tf))
print("# per bench.py kernel class")
out = {}
for k, (busy, cu, n) in sorted(cls.items(), key=lambda kv: -kv[1][0]):
    frac = busy / (4.0 * cu) if cu > 0 else 0.0
    out[k] = {"mfma_busy_frac": round(frac, 4), "mfma_insts_per_step": n, "flop_per_step": 16384.0 * n}
    print("%-60s %8s %12.0f %14.0f %14.0f %10.2f" % (k, "", n, busy, cu, 100.0 * frac))
if jpath:
    with open(jpath, "w") as f:
        json.dump({"note": "SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES) per bench.py kernel class, rocprofv3 --pmc pass of "
                           "bench.py --steps 2 --warmup 2 (tools/pmc_sq.sh pass 3)", "classes": out}, f, indent=1)
