"""Copy what tools/collect_round.sh left in gpurun_out/ into profiles/ under the round tag:  python3 tools/install_profiles.py r03
(kernel trace, timeline, PMC passes, SQ / MFMA tables, the default bench line, the variant benches folded into ONE json, the
per-launch list of the fused-block run)."""
import json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
COPY = {"prof_%s_kernel_trace.txt": "%s_kernel_trace.txt", "prof_%s_timeline.txt": "%s_timeline.txt",
        "prof_%s_pmc_fetch_size.txt": "%s_pmc_fetch_size.txt", "prof_%s_pmc_write_size.txt": "%s_pmc_write_size.txt",
        "%s_pmc_per_class.json": "%s_pmc_per_class.json", "%s_sq_counters.txt": "%s_sq_counters.txt",
        "%s_mfma_busy.txt": "%s_mfma_busy.txt", "%s_mfma_per_class.json": "%s_mfma_per_class.json",
        "%s_bench_default.json": "%s_bench_default.json", "%s_detail.txt": "%s_per_launch.txt"}
for a, b in COPY.items():
    shutil.copyfile(os.path.join(src, a % tag), os.path.join(dst, b % tag))


def line(name):
    with open(os.path.join(src, "%s_bench_%s.json" % (tag, name))) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def entry(d, classes=True):
    e = {"value": d["value"], "ms_per_step": d["ms_per_step"], "workload": d["config"]["workload"], "global_batch": d["config"]["global_batch"]}
    if classes and "kernel_classes" in d:
        e["kernel_classes_ms"] = {k: v["ms_per_step"] for k, v in sorted(d["kernel_classes"].items(), key=lambda kv: -kv[1]["ms_per_step"])}
    return e


runs = {}
h = line("h2d")
runs["h2d"] = dict(entry(h, False), pcie_inclusive=h["pcie_inclusive"])
for name in ("k5", "se", "ccf"):
    runs[name] = entry(line(name))
for hw in ("384x512", "512x512", "512x384"):
    runs[hw] = entry(line(hw), False)
c = line("clusters")
runs["clusters"] = {"value": c["value"], "ms_per_step": c["ms_per_step"], "workload": c["config"]["workload"],
                    "steps_per_cluster": c["config"]["steps_per_cluster"], "programs_live": c["config"]["programs_live"],
                    "clusters": c["clusters"]}
note = ("one gpurun call (tools/collect_round.sh): variants of python bench.py on 1 x MI355X; clusters = bench.py --clusters (one "
        "resolution cluster per step, DistributedClusterSampler); rect shapes at --batch 64.  Box-to-box spread on this pool: the same commit has measured between 19.6k and 22.8k img/s on "
        "different boxes (default command), so only numbers from the same call are comparable.")
with open(os.path.join(dst, "%s_bench_variants.json" % tag), "w") as f:
    json.dump({"note": "round %s, " % tag[1:] + note, "runs": runs}, f, indent=1)
print("installed profiles/%s_*" % tag)
