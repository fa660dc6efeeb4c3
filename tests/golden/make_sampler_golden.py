"""Generates tests/golden/sampler.json by running the REFERENCE sampler (imported from /root/reference/src, which exists
only in the build container) on synthetic cluster index lists under fixed `random.seed`s.  Run:
    python tests/golden/make_sampler_golden.py
The fixture holds the case definitions (data) and the index streams the reference yielded (outputs) only."""
import contextlib
import io
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/src")
from utils.cluster_random_sampler import ClusterRandomSampler as Ref  # noqa: E402


class FakeDataset:
    def __init__(self, cluster_indices, oversampling=None):
        self.cluster_indices = cluster_indices
        if oversampling is not None:
            self.oversampling_indices = oversampling


def clusters(sizes, start=0):
    out, n = [], start
    for s in sizes:
        out.append(list(range(n, n + s)))
        n += s
    return out


CASES = {
    # name: (cluster sizes, batch size, shuffle, oversampling repeats per item or None, seed, epochs)
    "three_clusters_b4": ([10, 7, 13], 4, True, None, 1, 3),
    "ragged_dropped_b5": ([4, 5, 11], 5, True, None, 7, 2),
    "no_shuffle_b3": ([7, 3, 8], 3, False, None, 3, 2),
    "oversampled_b4": ([5, 6], 4, True, [[1, 2, 1, 3, 1], [2, 1, 1, 1, 2, 1]], 11, 2),
    "imnet_like_b8": ([64, 96, 40], 8, True, None, 42, 2),
}

if __name__ == "__main__":
    out = {}
    for name, (sizes, bs, shuffle, over, seed, epochs) in CASES.items():
        random.seed(seed)
        with contextlib.redirect_stdout(io.StringIO()):
            s = Ref(FakeDataset(clusters(sizes), over), bs, shuffle)
        streams = [list(iter(s)) for _ in range(epochs)]
        out[name] = {"sizes": sizes, "batch_size": bs, "shuffle": shuffle, "oversampling": over, "seed": seed,
                     "len": len(s), "epochs": streams}
    with open(os.path.join(HERE, "sampler.json"), "w") as f:
        json.dump(out, f)
    print("wrote sampler.json:", {k: v["len"] for k, v in out.items()})
