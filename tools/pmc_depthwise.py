"""HBM traffic of the row-marching depthwise kernels per LAYER from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate
passes as the TCC slot budget requires) of `tools/bench_dwmarch.py --n 8` (cold operands), against the algorithmic bytes of SURVEY
8(d).  Launches are mapped to layers by kernel instantiation + grid size (the library's own block-count queries).  gfx950
corrections of /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counter unit KiB; FETCH_SIZE under-reports wide coalesced read
streams by exactly 2x -> doubled; WRITE_SIZE exact.

    python tools/pmc_depthwise.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [N]
"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mliis_amd import spec  # noqa: E402
from mliis_amd._lib import lib  # noqa: E402
from tools.pmc_traffic import csrc_sha1  # noqa: E402


def agg(path, counter):
    d = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "dwm_" in r["Kernel_Name"]:
            name = r["Kernel_Name"].replace("void ", "").replace("mliis::", "").split("(")[0]
            e = d[(name, int(r["Grid_Size"]))]
            e[0] += float(r["Counter_Value"])
            e[1] += 1
    return d


def main():
    f, w, out = sys.argv[1:4]
    N = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    fa, wa = agg(f, "FETCH_SIZE"), agg(w, "WRITE_SIZE")
    rows = []
    for b in spec.derive().blocks:
        if not b.executed or (b.h_in <= 14 and b.h_out <= 14):
            continue
        C, k, s, hi, ho = b.cexp, b.k, b.stride, b.h_in, b.h_out
        cgs = -(-C // 32)
        ein, eout, ew = N * hi * hi * C, N * ho * ho * C, k * k * C
        for direction, blocks, alg, pat in (
                ("fwd", lib.raw("mliis_dwconv_bn_fwd_blocks")(N, hi, hi, C, k, s), 4.0 * (ein + eout + ew), "dwm_conv_k<%d, %d, true, false" % (k, s)),
                ("bwd", lib.raw("mliis_dwconv_bn_bwd_blocks")(N, hi, hi, C, k, s), 4.0 * (2 * ein + eout + 2 * ew),
                 ("dwm_conv_k<%d, 1, true, true" % k) if s == 1 else ("dwm_bwd_s2_k<%d, true" % k))):
            grid = blocks * cgs * 256
            hit = [(key, fa[key]) for key in fa if key[0].startswith(pat) and key[1] == grid]
            if not hit:
                continue
            key, (fs, fn) = hit[0]
            ws_, wn = wa.get(key, [0.0, 0])
            rd = 2.0 * fs * 1024.0 / max(fn, 1)
            wr = ws_ * 1024.0 / max(wn, 1)
            rows.append({"block": b.idx, "dir": direction, "kernel": key[0], "workgroups": blocks * cgs, "launches_counted": fn,
                         "algorithmic_MB": alg / 1e6, "hbm_read_MB": rd / 1e6, "hbm_write_MB": wr / 1e6, "traffic_over_algorithmic": (rd + wr) / alg})
            print("b%-2d %s  %-34s %5d wg: read %6.2f MB + write %6.2f MB = %6.2f MB vs %6.2f MB algorithmic -> %.3fx" % (
                b.idx, direction, key[0], blocks * cgs, rd / 1e6, wr / 1e6, (rd + wr) / 1e6, alg / 1e6, (rd + wr) / alg))
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of tools/bench_dwmarch.py --n %d (cold operands: rotating "
                       "copies); read = 2 * FETCH_SIZE * 1024 (gfx950 correction), write = WRITE_SIZE * 1024; per launch" % N,
               "csrc_sha1": csrc_sha1(), "layers": rows}, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
