"""Aggregates two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; collected separately as the TCC slot budget requires) into
per-kernel HBM traffic per launch, with the gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
counter unit = KiB; FETCH_SIZE under-reports wide coalesced read streams by exactly 2x -> doubled; WRITE_SIZE is exact.

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [inner steps of the profiled run]
"""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha1():
    """Fingerprint of the kernel sources the counters were collected on (bench.py recomputes it and reports whether the committed
    traffic file still describes the kernels it is running)."""
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(ROOT, "mliis_amd", "csrc", "*.h*")) + [os.path.join(ROOT, "include", "mliis_hip.h")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def agg(path, counter):
    d = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            e = d[r["Kernel_Name"]]
            e[0] += float(r["Counter_Value"])
            e[1] += 1
    return d


def short(name):
    n = name.replace("void ", "").replace("mliis::", "")
    return n.split("(")[0].replace(", ", ",")


def main():
    f, w, out = sys.argv[1:4]
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    fa, wa = agg(f, "FETCH_SIZE"), agg(w, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fa) | set(wa)):
        fs, fn = fa.get(k, [0.0, 0])
        ws, wn = wa.get(k, [0.0, 0])
        n = max(fn, wn, 1)
        rd = 2.0 * fs * 1024.0 / max(fn, 1)
        wr = ws * 1024.0 / max(wn, 1)
        res[short(k)] = {"launches": n, "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
                         "raw_FETCH_SIZE_KiB_per_launch": fs / max(fn, 1), "raw_WRITE_SIZE_KiB_per_launch": ws / max(wn, 1)}
    total = sum(v["launches"] * v["hbm_bytes_per_launch"] for v in res.values())
    json.dump({"inner_steps_profiled": steps, "hbm_bytes_per_inner_step": (total / steps) if steps else None,
               "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `bench.py --steps 1 --warmup 1 --no-graph`; "
                       "read = 2 * FETCH_SIZE * 1024 (gfx950 correction), write = WRITE_SIZE * 1024; averaged over all launches of a symbol",
               "csrc_sha1": csrc_sha1(), "kernels": res}, open(out, "w"), indent=1, sort_keys=True)
    print("wrote", out, len(res), "kernels")


if __name__ == "__main__":
    main()
