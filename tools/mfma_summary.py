#!/usr/bin/env python3
"""Per-kernel matrix-core table from pass 3 of tools/pmc_sq.sh (+ the kernel trace for durations).
usage: mfma_summary.py pmc_<tag>_3.txt [prof_<tag>_kernel_trace.txt] > profiles/<tag>_mfma_busy.txt

mfma_busy% = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES): the fraction of the time a CU had work during which its
matrix pipes were executing (16 busy cycles per v_mfma_f32_16x16x32_bf16, MI355X_MICROARCH.md cycle table).
TFLOP/s = 16384 flop x SQ_INSTS_MFMA / kernel time (when a kernel trace is given)."""
import collections
import re
import sys

d = collections.defaultdict(dict)
for l in open(sys.argv[1]):
    if l.startswith('#') or l.startswith('kernel'):
        continue
    m = re.match(r'(.{86}) (\S+)\s+([\d.]+)\s+([\d.]+)', l)
    if not m:
        continue
    k, c = m.group(1).strip(), m.group(2)
    d[k][c] = float(m.group(4))
    d[k]['calls'] = float(m.group(3))
dur = {}
if len(sys.argv) > 2:
    for l in open(sys.argv[2]):
        m = re.match(r'(.{86})\s+(\d+)\s+([\d.]+)\s+([\d.]+)', l)
        if m:
            dur[m.group(1).strip()] = (float(m.group(2)), float(m.group(4)))       # calls per step, avg us
print("# matrix-core counters per kernel, per bench step (sums over 4 steps / 4)")
print("%-60s %6s %12s %12s %10s %10s %9s" % ("kernel", "calls", "mfma_insts", "mfma_busy_cy", "cu_busy_cy", "mfma_busy%", "TFLOP/s"))
rows = sorted(((c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0), k, c) for k, c in d.items()), reverse=True)
for _, k, c in rows[:40]:
    busy, cu = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0), max(c.get('SQ_BUSY_CU_CYCLES', 0.0), 1.0)
    n = c.get('SQ_INSTS_MFMA', 0.0)
    tf = ""
    if k in dur and dur[k][1] > 0:
        tf = "%.1f" % (16384.0 * n / (dur[k][0] * dur[k][1] * 1e-6) / 1e12)
    print("%-60s %6.0f %12.0f %12.0f %10.0f %10.2f %9s" % (k[:60], c['calls'], n, busy, cu, 100.0 * busy / (4.0 * cu), tf))
