#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg / %.
usage: rocpd_stats.py results.db [steps] > profiles/<name>.txt      (steps: divide totals to get per-step)"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    rows = cur.execute("select s.display_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
                       "max(s.arch_vgpr_count), max(d.group_segment_size) from `%s` d join `%s` s on d.kernel_id=s.id "
                       "group by s.display_name order by 3 desc" % (kd, ks)).fetchall()
    t0, t1 = cur.execute("select min(start), max(end) from `%s`" % kd).fetchone()
    tot = sum(r[2] for r in rows)
    print("# rocprofv3 --kernel-trace summary of %s" % sys.argv[1])
    print("# kernels busy %.3f ms total over a %.3f ms span; divided by %g steps below" % (tot / 1e6, (t1 - t0) / 1e6, steps))
    print("%-86s %8s %12s %10s %10s %10s %6s %5s %7s" % ("kernel", "calls", "ms/step", "avg_us", "min_us", "max_us", "%", "vgpr", "lds"))
    for name, n, s, mn, mx, vg, lds in rows:
        nm = name if len(name) <= 86 else name[:83] + "..."
        print("%-86s %8.1f %12.3f %10.2f %10.2f %10.2f %6.2f %5s %7s" % (nm, n / steps, s / 1e6 / steps, s / n / 1e3, mn / 1e3, mx / 1e3,
                                                                    100.0 * s / tot, vg, lds))




def pmc_main(path, steps):
    """Per-kernel sum of every collected PMC counter: rocpd_stats.py --pmc results.db [steps]"""
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    pe = [t for t in tabs if "pmc_event" in t][0]
    pi = [t for t in tabs if "info_pmc" in t][0]
    cols_pe = [r[1] for r in cur.execute("pragma table_info(`%s`)" % pe)]
    cols_pi = [r[1] for r in cur.execute("pragma table_info(`%s`)" % pi)]
    print("# pmc_event columns:", cols_pe)
    print("# info_pmc columns:", cols_pi)
    namecol = "name" if "name" in cols_pi else cols_pi[-1]
    q = ("select s.display_name, p.%s, count(*), sum(e.value) from `%s` e join `%s` p on e.pmc_id=p.id "
         "join `%s` d on e.event_id=d.event_id join `%s` s on d.kernel_id=s.id group by s.display_name, p.%s order by 4 desc"
         % (namecol, pe, pi, kd, ks, namecol))
    print("# per-kernel PMC sums, divided by %g steps" % steps)
    print("%-86s %-14s %8s %16s" % ("kernel", "counter", "calls", "sum/step"))
    for name, cname, n, v in cur.execute(q):
        nm = name if len(name) <= 86 else name[:83] + "..."
        print("%-86s %-14s %8.1f %16.1f" % (nm, cname, n / steps, (v or 0) / steps))


def timeline_main(path, nlast):
    """Idle analysis: rocpd_stats.py --timeline results.db [N]  -- union of kernel intervals vs span over the LAST N
    dispatches (default 2000), per-queue gap totals, and the 25 largest gaps with the kernels on either side."""
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    cols = [r[1] for r in cur.execute("pragma table_info(`%s`)" % kd)]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    q = "select d.start, d.end, s.display_name, %s from `%s` d join `%s` s on d.kernel_id=s.id order by d.start" % (
        ("d." + qcol) if qcol else "0", kd, ks)
    rows = cur.execute(q).fetchall()[-nlast:]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    busy, cur_end = 0, t0
    gaps = []
    prev_name = None
    for st, en, name, qq in rows:
        if st > cur_end:
            gaps.append((st - cur_end, prev_name, name))
            busy += en - st
            cur_end = en
        elif en > cur_end:
            busy += en - cur_end
            cur_end = en
        if en >= cur_end:
            prev_name = name
    print("# last %d dispatches: span %.3f ms, some kernel running %.3f ms (%.1f %%), idle %.3f ms in %d gaps" % (
        len(rows), (t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), (t1 - t0 - busy) / 1e6, len(gaps)))
    byq = {}
    for st, en, name, qq in rows:
        byq.setdefault(qq, []).append((st, en, name))
    for qq, lst in byq.items():
        b = sum(e - s_ for s_, e, _ in lst)
        print("# queue %s: %d dispatches, kernel time %.3f ms" % (qq, len(lst), b / 1e6))
    hist = {}
    for g, a, b in gaps:
        k = "<2us" if g < 2000 else "<5us" if g < 5000 else "<10us" if g < 10000 else "<50us" if g < 50000 else ">=50us"
        h = hist.setdefault(k, [0, 0]); h[0] += 1; h[1] += g
    for k in ("<2us", "<5us", "<10us", "<50us", ">=50us"):
        if k in hist:
            print("# gaps %-6s n=%-5d total %.3f ms" % (k, hist[k][0], hist[k][1] / 1e6))
    # dispatch sequence around the step boundary (the optimizer kernel): name, queue, start offset, duration
    idx = [i for i, r in enumerate(rows) if r[2].startswith("k_adam")]
    if idx:
        i0 = idx[len(idx) // 2]
        base = rows[i0][0]
        print("# around a step boundary (us relative to k_adam start): queue start dur name")
        for st, en, name, qq in rows[max(0, i0 - 6):i0 + 40]:
            print("#   q%s %9.1f %8.1f  %s" % (qq, (st - base) / 1e3, (en - st) / 1e3, name[:70]))
    # gaps of 2..50 us grouped by the kernel that was WAITING to start (round 5: kernels that use scratch memory pay a dispatch
    # set-up of ~6 us that the others do not)
    grp = {}
    for g, a, b in gaps:
        if 2000 <= g < 50000:
            e = grp.setdefault((b or "")[:70], [0, 0]); e[0] += 1; e[1] += g
    print("# gaps of 2-50 us by the kernel that starts after them: n, total us, mean us")
    for k, (n_, tot) in sorted(grp.items(), key=lambda kv: -kv[1][1])[:30]:
        print("#   %4d %9.1f %6.2f  %s" % (n_, tot / 1e3, tot / 1e3 / n_, k))
    print("# largest gaps (us): after -> before")
    for g, a, b in sorted(gaps, key=lambda x: -x[0])[:25]:
        print("%9.1f  %s  ->  %s" % (g / 1e3, (a or "")[:60], (b or "")[:60]))


def seq_main(path):
    """One full step, dispatch by dispatch: rocpd_stats.py --seq results.db  (k_adam to the next k_adam, the last complete one)"""
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    cols = [r[1] for r in cur.execute("pragma table_info(`%s`)" % kd)]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    q = "select d.start, d.end, s.display_name, %s from `%s` d join `%s` s on d.kernel_id=s.id order by d.start" % (
        ("d." + qcol) if qcol else "0", kd, ks)
    rows = cur.execute(q).fetchall()
    idx = [i for i, r in enumerate(rows) if r[2].startswith("k_adam")]
    if len(idx) < 3:
        print("# fewer than 3 steps in the trace")
        return
    i0, i1 = idx[-3], idx[-2]
    base = rows[i0][0]
    for st, en, name, qq in rows[i0:i1 + 1]:
        print("q%s %9.1f %8.1f  %s" % (qq, (st - base) / 1e3, (en - st) / 1e3, name[:90]))



def grids_main(path):
    """Wave quantisation: rocpd_stats.py --grids results.db -- per (kernel, grid) the workgroups launched against the resident
    capacity of the chip (256 CUs x workgroups per CU from the VGPR / LDS / wave limits): a launch of 1.05 rounds pays for 2."""
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    dcols = [r[1] for r in cur.execute("pragma table_info(`%s`)" % kd)]
    scols = [r[1] for r in cur.execute("pragma table_info(`%s`)" % ks)]
    print("# dispatch columns:", dcols)
    print("# symbol columns:", scols)
    gx = [c for c in dcols if c.startswith("grid_size")]
    wx = [c for c in dcols if c.startswith("workgroup_size")]
    acc = "s.accum_vgpr_count" if "accum_vgpr_count" in scols else "0"
    q = ("select s.display_name, %s, %s, max(d.group_segment_size), max(s.arch_vgpr_count), max(%s), count(*), sum(d.end-d.start) "
         "from `%s` d join `%s` s on d.kernel_id=s.id group by s.display_name, %s, %s order by 11 desc"
         % (", ".join("d." + c for c in gx), ", ".join("d." + c for c in wx), acc, kd, ks,
            ", ".join("d." + c for c in gx), ", ".join("d." + c for c in wx)))
    print("%-70s %9s %5s %7s %5s %6s %7s %7s %8s" % ("kernel", "wgs", "thr", "lds", "vgpr", "wg/CU", "rounds", "calls", "avg_us"))
    for row in cur.execute(q):
        name = row[0]
        g = row[1:1 + len(gx)]
        w = row[1 + len(gx):1 + len(gx) + len(wx)]
        lds, vg, ac, n, tot = row[1 + len(gx) + len(wx):]
        thr = 1
        for v in w: thr *= max(1, v)
        items = 1
        for v in g: items *= max(1, v)
        wgs = items // thr if items >= thr and items % thr == 0 else items          # grid in work-items (HSA) or in workgroups
        waves = (thr + 63) // 64
        regs = ((vg or 0) + (ac or 0) + 7) // 8 * 8
        wps = min(8, 512 // regs) if regs else 8                                    # waves per SIMD
        by_waves = max(1, 4 * wps // waves)
        by_lds = (160 * 1024) // lds if lds else 32
        res = max(1, min(by_waves, by_lds, 32))
        nm = name if len(name) <= 70 else name[:67] + "..."
        print("%-70s %9d %5d %7d %5d %6d %7.2f %7d %8.1f" % (nm, wgs, thr, lds or 0, regs, res, wgs / (256.0 * res), n, tot / n / 1e3))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "--grids":
    grids_main(sys.argv[2])
    sys.exit(0)

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "--seq":
    seq_main(sys.argv[2])
    sys.exit(0)

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "--timeline":
    timeline_main(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 2000)
    sys.exit(0)

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "--pmc":
    pmc_main(sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 1.0)
    sys.exit(0)

if __name__ == "__main__":
    main()
