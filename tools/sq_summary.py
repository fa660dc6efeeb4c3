#!/usr/bin/env python3
"""Per-kernel SQ counter table from the two passes of tools/pmc_sq.sh.
usage: sq_summary.py pmc_<tag>_1.txt pmc_<tag>_2.txt > profiles/<tag>_sq_counters.txt
Percentages are of SQ_WAVE_CYCLES (per-wave cycles).  VALU-issue utilisation of a SIMD ~ valu% x resident waves per SIMD."""
import collections
import re
import sys

d = collections.defaultdict(dict)
for f in sys.argv[1:3]:
    for l in open(f):
        if l.startswith('#') or l.startswith('kernel'):
            continue
        m = re.match(r'(.{86}) (\S+)\s+([\d.]+)\s+([\d.]+)', l)
        if not m:
            continue
        k, c = m.group(1).strip(), m.group(2)
        d[k][c] = float(m.group(4))
        d[k]['calls'] = float(m.group(3))
rows = sorted(((c.get('SQ_BUSY_CYCLES', 0), k, c) for k, c in d.items() if 'SQ_WAVE_CYCLES' in c), reverse=True)
print("# SQ counters per kernel (sums over 4 bench steps / 4); act = SQ_ACTIVE_INST_ANY, valu = SQ_ACTIVE_INST_VALU, lds = SQ_ACTIVE_INST_LDS,")
print("# waitI = SQ_WAIT_INST_ANY (issue stalls), waitA = SQ_WAIT_ANY (s_waitcnt / barrier), all as % of SQ_WAVE_CYCLES;")
print("# valu/wv, lds/wv = instructions per wave; bankcf = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE")
print("%-50s %6s %10s %6s %6s %6s %6s %6s %8s %7s %7s" % ("kernel", "calls", "wavecyc(M)", "act%", "valu%", "lds%", "waitI%", "waitA%",
                                                        "valu/wv", "lds/wv", "bankcf%"))
for _, k, c in rows[:40]:
    wc = c['SQ_WAVE_CYCLES']
    f = lambda x: 100.0 * c.get(x, 0) / wc
    wv = max(c.get('SQ_WAVES', 1), 1)
    print("%-50s %6.0f %10.1f %6.1f %6.1f %6.1f %6.1f %6.1f %8.0f %7.0f %7.1f" % (
        k[:50], c['calls'], wc / 1e6, f('SQ_ACTIVE_INST_ANY'), f('SQ_ACTIVE_INST_VALU'), f('SQ_ACTIVE_INST_LDS'), f('SQ_WAIT_INST_ANY'),
        f('SQ_WAIT_ANY'), c.get('SQ_INSTS_VALU', 0) / wv, c.get('SQ_INSTS_LDS', 0) / wv,
        100.0 * c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c.get('SQ_LDS_IDX_ACTIVE', 1), 1)))
