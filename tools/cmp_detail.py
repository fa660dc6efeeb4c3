"""Compare two MNAS_BENCH_DETAIL stderr dumps: mean us per (kernel, shape)."""
import sys, collections
def load(p):
    d = collections.OrderedDict()
    for l in open(p):
        f = l.split()
        if len(f) < 6 or not f[0].startswith("k_"): continue
        d.setdefault((f[0], f[1]), []).append(float(f[2]))
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
pat = sys.argv[3] if len(sys.argv) > 3 else ""
ta = tb = 0
for k in a:
    if pat and pat not in k[0]: continue
    ma = sum(a[k]) / len(a[k]); mb = sum(b.get(k, [0])) / max(1, len(b.get(k, [0])))
    ta += sum(a[k]); tb += sum(b.get(k, [0]))
    print("%-18s %-46s x%d %8.1f -> %8.1f  %+5.0f%%" % (k[0], k[1], len(a[k]), ma, mb, 100 * (mb - ma) / ma))
print("total %.1f -> %.1f us" % (ta, tb))
