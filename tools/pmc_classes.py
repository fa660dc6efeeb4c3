#!/usr/bin/env python3
"""Fold the per-kernel PMC summaries (rocpd_stats.py --pmc) into bench.py's kernel classes.
usage: pmc_classes.py fetch_summary.txt write_summary.txt > profiles/<tag>_pmc_per_class.json
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled (gfx950 rocprofv3 tallies 128-byte read requests at 64 B:
MI355X_MICROARCH.md, HBM section); WRITE_SIZE is taken as reported (uncalibrated there)."""
import json, re, sys


def key(name):
    m = re.match(r"void k_igemm<(\d)", name)
    if m:
        return {"0": "k_igemm<fwd>", "1": "k_igemm<dgrad>", "2": "k_igemm<stem>"}[m.group(1)]
    m = re.match(r"void k_pws<(\d)", name)
    if m:
        return "k_igemm<dgrad>" if m.group(1) == "1" else "k_igemm<fwd>"      # bench.py classes follow the C-ABI entry point
    m = re.match(r"void k_pwf<(\d)", name)
    if m:
        return "k_igemm<dgrad>" if m.group(1) == "1" else "k_igemm<fwd>"
    m = re.match(r"void k_dimg<(\d)", name)
    if m:
        return "k_igemm<dgrad>" if m.group(1) == "1" else "k_igemm<fwd>"
    m = re.match(r"void k_c3r<(\d)", name)
    if m:
        return "k_igemm<dgrad>" if m.group(1) == "1" else "k_igemm<fwd>"
    if name.startswith("void k_pwx<") or name.startswith("void k_c3x<"): return "k_igemm<fwd>"
    if name.startswith("void k_tconv") or name.startswith("void k_tcx<") or name.startswith("void k_tcr<"): return "k_igemm<dgrad>"
    if name.startswith("k_stem_fwd"): return "k_igemm<stem>"
    if name.startswith("k_stem_wgrad"): return "k_wgrad<stem>"
    if name.startswith("k_dy_mat"): return "k_dy_mat"
    if name.startswith("k_pool"): return "k_pool"
    if name.startswith("void k_wgrad<true>"): return "k_wgrad<stem>"
    if name.startswith("void k_wgrad<false>") or name.startswith("void k_wgrad_t<"): return "k_wgrad"
    if name.startswith("void k_pw_bwd"): return "k_pw_bwd"
    if name.startswith("void k_dw_fwd"): return "k_dw_conv<fwd>"
    m = re.match(r"void k_dw_bwd<\d, (true|false), (true|false)", name)
    if m:
        dg, wg = m.group(1) == "true", m.group(2) == "true"
        return "k_dw_bwd" if dg and wg else ("k_dw_conv<dgrad>" if dg else "k_dw_wgrad")
    if name.startswith("k_add_act"): return "k_add_act"
    if name.startswith("k_gram") or name.startswith("void k_gram"): return "k_gram"
    if name.startswith("k_se_") or name.startswith("void k_se_"): return "k_se"
    return "other"


def load(path, counter):
    out = {}
    for l in open(path):
        if l.startswith("#") or counter not in l: continue
        name = l[:l.index(counter)].rstrip()
        calls, val = l[l.index(counter) + len(counter):].split()
        k = key(name)
        a = out.setdefault(k, [0.0, 0.0])
        a[0] += float(calls); a[1] += float(val)
    return out


def main():
    f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    res, tot = {}, 0.0
    for k in sorted(set(f) | set(w)):
        calls = f.get(k, w.get(k))[0]
        fetch_b = 2.0 * f.get(k, [0, 0])[1] * 1024.0
        write_b = w.get(k, [0, 0])[1] * 1024.0
        tot += fetch_b + write_b
        res[k] = {"launches_per_step": calls, "fetch_bytes_per_step": fetch_b, "write_bytes_per_step": write_b,
                  "hbm_bytes_per_launch": (fetch_b + write_b) / max(calls, 1.0)}
    print(json.dumps({"note": "FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, KiB -> bytes, separate --pmc passes of "
                              "bench.py --steps 2 --warmup 2; per training step (bs 256, 1 GPU)",
                      "total_GB_per_step": round(tot / 1e9, 2), "classes": res}, indent=1))


if __name__ == "__main__":
    main()
