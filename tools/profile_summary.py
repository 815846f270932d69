"""Summarises a rocprofv3 --kernel-trace --stats run of `bench.py` (csv output) into a markdown report: per-family time per inner
step, the top kernels, and the per-layer depthwise table against the HBM roofline (algorithmic bytes of SURVEY.md 8(d)).

    python tools/profile_summary.py <dir with *_kernel_stats.csv / *_kernel_trace.csv> <inner steps profiled, 0 = count them> <out.md>
"""
import collections
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mliis_amd import spec  # noqa: E402


def family(n):
    n = n.replace("void mliis::", "").replace("mliis::", "").split("(")[0]
    if n.startswith("conv_gemm") or n.startswith("conv1x1_") or n.startswith("conv_x3_k") or n.startswith("x3_fixup") or n.startswith("x3_pack"):
        return "dense conv fwd / bwd-data (MFMA implicit GEMM; conv_x3_k + x3_fixup_k + x3_pack_k: split products on the bf16 matrix cores)"
    if n.startswith("mbconv_dw"):
        return "small-map fused MBConv depthwise half (bn0 + dw + bn1 + pool | their backward)"
    if n.startswith("conv_filter"):
        return "dense conv bwd-filter (MFMA)"
    if n.startswith("dwm_"):
        return "depthwise, row-marching (bn0 + swish on load + dw + bn1 sums | one-pass backward)"
    if n.startswith("dwconv"):
        return "depthwise " + n.split("_k")[0].replace("dwconv_", "")
    if n.startswith("bn_") or "BnBwdOp" in n or "StatsOp" in n:
        return "batch norm (+swish, drop-connect, residual, SE gate grads)"
    if n.startswith("rsd"):
        return "RSD pooled branch"
    if "SumOp" in n or n.startswith("sum_fin") or n.startswith("colsum"):
        return "per-image column sums (SE pool, bias grads)"
    if n.startswith("se_"):
        return "squeeze-excite MLP"
    if n.startswith("fold") or n.startswith("splitk") or n.startswith("sk_fixup"):
        return "slab folds (split-K, weight grads)"
    if n.startswith("at::") or n.startswith("__amd"):
        return "torch plumbing (mask RNG, arena copies)"
    return "other (stem, resize, head, loss, SGD, transposes)"


def main():
    d, steps, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    stats = list(csv.DictReader(open(glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True)[0])))
    if steps <= 0:   # one optimizer launch per inner step
        steps = sum(int(r["Calls"]) for r in stats if "sgd_k" in r["Name"] or "adam_b1zero_k" in r["Name"])
    trace = list(csv.DictReader(open(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in stats) / 1e3 / steps
    nl = sum(int(r["Calls"]) for r in stats) / steps
    fam = collections.defaultdict(lambda: [0.0, 0.0])
    for r in stats:
        f = fam[family(r["Name"])]
        f[0] += int(r["Calls"]) / steps
        f[1] += float(r["TotalDurationNs"]) / 1e3 / steps
    L = ["# rocprofv3 --kernel-trace --stats summary (`bench.py`, N = 1, HIP-graph replay)", "",
         "Kernel time per inner step (8 images fwd+bwd+BN-EMA+SGD): **%.0f us in %.0f launches** (%d inner steps profiled)." % (tot, nl, steps), "",
         "| family | launches/step | us/step | share |", "|---|---|---|---|"]
    for k, (c, t) in sorted(fam.items(), key=lambda x: -x[1][1]):
        L.append("| %s | %.0f | %.0f | %.1f %% |" % (k, c, t, 100 * t / tot))
    L += ["", "| kernel | launches/step | avg us | us/step |", "|---|---|---|---|"]
    for r in sorted(stats, key=lambda r: -float(r["TotalDurationNs"]))[:25]:
        n = r["Name"].replace("void mliis::", "").replace("mliis::", "").split("(")[0]
        L.append("| `%s` | %.1f | %.1f | %.0f |" % (n, int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3 / steps))
    # depthwise per layer (identified by kernel instantiation + grid size)
    a = spec.derive()
    N = 8
    per = collections.defaultdict(list)
    for r in trace:
        n = r["Kernel_Name"]
        if "dwconv" in n:
            per[(n.split("(")[0].replace("void mliis::", ""), int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]))].append(
                int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))

    def grid(items):
        return -(-items // 256) * 256
    L += ["", "## Depthwise layers vs the HBM roofline, IN-STEP (graph replay: operands partly resident in the 256 MiB Infinity Cache, the producer "
          "ran just before; bench.py's `depthwise_hbm` has the cold-operand figures)", "",
          "8.0 TB/s spec; algorithmic bytes (SURVEY 8(d)): fwd 4*(in+out+k*k*C); one-pass backward 4*(2*in+out+2*k*k*C) (dY and X read, dX written); "
          "for the op-by-op kernels bwd-data 4*(dY+dX+k*k*C) (+ 4*z0 where the launch also emits the expand "
          "BN's backward sums); bwd-filter 4*(X+dY+k*k*C).  Blocks 6-10 (14x14): the depthwise op lives inside the fused small-map kernels "
          "(`mbconv_dw_fwd_small_k` = bn0 apply + depthwise + bn1 statistics/apply + SE pooling; `mbconv_dw_bwd_small_k` = bn1 backward + "
          "depthwise backward-data + backward-filter + bn0 backward): their whole duration is charged to the depthwise bytes (fwd; bwd = bwd-data + bwd-filter bytes).", "",
          "| block | C | k,s | map | fwd us | fwd % of 8 TB/s | bwd-data us | % | bwd-filter us | % |", "|---|---|---|---|---|---|---|---|---|---|"]
    tf = tb = tw = bf = bb = bw = 0.0
    per_f = collections.defaultdict(list)
    for r in trace:
        n = r["Kernel_Name"]
        if "mbconv_dw" in n or "dwconv_bwd_filter" in n:
            per_f[(n.split("(")[0].replace("void mliis::", ""), int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]))].append(
                int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))

    def med(v):
        return sorted(v)[len(v) // 2] / 1e3

    def grid(items):
        return -(-items // 256) * 256
    small_seen = collections.defaultdict(int)
    from mliis_amd._lib import lib as _lib
    per_m = collections.defaultdict(list)
    for r in trace:
        n = r["Kernel_Name"]
        if "dwm_" in n:
            per_m[(n.split("(")[0].replace("void mliis::", ""), int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]))].append(
                int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for b in a.blocks:
        q = b.cexp // 4
        by = 4.0 * (N * b.h_in ** 2 * b.cexp + N * b.h_out ** 2 * b.cexp + b.k ** 2 * b.cexp)
        byb = by + (4.0 * N * b.h_in ** 2 * b.cexp if b.expand != 1 else 0.0)   # + z0 where the launch also does the BN statistics
        # grid of the fused small-map kernels (threads): V channels per workgroup (quads; pairs for 5x5 layers with C / 2 <= 256 CUs),
        # runs of R groups per XCD (mbconv_small.hip: sm_group_width / sm_grid)
        V = 2 if (b.k == 5 and b.cexp % 2 == 0 and b.cexp // 2 <= 256) else 4
        R = 11 if V == 3 else 32 // V
        sg = ((-(-(b.cexp // V) // R) + 7) // 8) * 8 * R * 512
        kfs = [v for (k, g, gy), v in per_f.items() if k.startswith("mbconv_dw_fwd_small_k<%d, %d" % (b.k, V)) and g == sg]
        kbs = [v for (k, g, gy), v in per_f.items() if k.startswith("mbconv_dw_bwd_small_k<%d, %d" % (b.k, V)) and g == sg]
        if kfs and kbs and b.stride == 1 and N * b.h_in ** 2 <= 2048 and b.expand != 1:
            mf, mb = med(kfs[0]), med(kbs[0])
            tf += mf; bf += by; tb += mb; bb += by + by
            L.append("| %d | %d | %d,%d | %d->%d | %.1f (fused) | %.0f %% | %.1f (fused: bwd-data + bwd-filter) | %.0f %% | | |" % (
                b.idx, b.cexp, b.k, b.stride, b.h_in, b.h_out, mf, by / mf / 1e3 / 80, mb, 2 * by / mb / 1e3 / 80))
            continue
        # row-marching kernels (csrc/dwmarch.hip): identified by instantiation + grid (the library's own block-count queries)
        cy_ = -(-b.cexp // 32)
        gfm = _lib.raw("mliis_dwconv_bn_fwd_blocks")(N, b.h_in, b.h_in, b.cexp, b.k, b.stride)
        gbm = _lib.raw("mliis_dwconv_bn_bwd_blocks")(N, b.h_in, b.h_in, b.cexp, b.k, b.stride)
        # template arguments: dwm_conv_k<K, S, PRE, BWD, DYBN, TA, TB>, dwm_bwd_s2_k<K, PRE, DYBN, TA, TB>
        def targs(k):
            return [x.strip() for x in k[k.index("<") + 1:k.rindex(">")].split(",")]
        mfw = [v for (k, g, gy), v in per_m.items() if k.startswith("dwm_conv_k<%d, %d, " % (b.k, b.stride)) and targs(k)[3] == "false" and g == gfm * 256 and gy == cy_]
        mbw = [v for (k, g, gy), v in per_m.items() if ((k.startswith("dwm_conv_k<%d, 1, " % b.k) and targs(k)[3] == "true") if b.stride == 1 else
                                                        k.startswith("dwm_bwd_s2_k<%d, " % b.k)) and g == gbm * 256 and gy == cy_]
        if mfw and mbw:
            mf, mb = med(mfw[0]), med(mbw[0])
            byb2 = 4.0 * (2 * N * b.h_in ** 2 * b.cexp + N * b.h_out ** 2 * b.cexp + 2 * b.k ** 2 * b.cexp)
            tf += mf; bf += by; tb += mb; bb += byb2
            L.append("| %d | %d | %d,%d | %d->%d | %.1f (march: + bn0 fold/apply, bn1 sums) | %.0f %% | %.1f (march, one pass: dx + dW slabs + bn0 sums) | %.0f %% | | |" % (
                b.idx, b.cexp, b.k, b.stride, b.h_in, b.h_out, mf, by / mf / 1e3 / 80, mb, byb2 / mb / 1e3 / 80))
            continue
        gf = grid(N * b.h_out * (-(-b.h_out // 4)) * q)
        gb = grid(N * b.h_in * (-(-b.h_in // 4)) * q)
        gs = -(-(N * b.h_out * (-(-b.h_out // 4))) // 32) * 256
        kf = [v for (k, g, gy), v in per.items() if (k.startswith("dwconv_fwd_k<%d, %d" % (b.k, b.stride)) and g == gf) or
              (k.startswith("dwconv_fwd_stats_k<%d, %d" % (b.k, b.stride)) and g == gs and gy == -(-b.cexp // 32))]
        kb = [v for (k, g, gy), v in per.items() if k.startswith("dwconv_bwd_data_k<%d, %d" % (b.k, b.stride)) and g == gb]

        def tiles(h, tow):
            return N * (-(-h // 7)) * (-(-h // tow)) * 256
        cy = -(-b.cexp // 32)
        if b.k == 5 and b.stride == 1:
            kf += [v for (k, g, gy), v in per.items() if k.startswith("dwconv_tile_k<5, 1,") and ", false, true, false, false>" in k and
                   g == tiles(b.h_out, 16) and gy == cy]
        dil = "true" if b.stride == 2 else "false"
        kb += [v for (k, g, gy), v in per.items() if k.startswith("dwconv_tile_k<%d, 1," % b.k) and (", true, false, %s, " % dil) in k and
               g == tiles(b.h_in, 16) and gy == cy]
        # filter gradient: dwconv_bwd_filter_k<K, S, 4>, identified by its grid (dw_filter_geom of csrc/dwconv.hip)
        qb = min(q, 64)
        rp, ny = 256 // qb, -(-q // qb)
        items = N * b.h_out * (-(-b.h_out // 4))
        ipb = max(-(-items // max(1, 512 // ny)), rp)
        ipb = -(-ipb // rp) * rp
        nblk = -(-items // ipb)
        kw = [v for (k, g, gy), v in per_f.items() if k.startswith("dwconv_bwd_filter_k<%d, %d" % (b.k, b.stride)) and gy == ny and g == nblk * 256]
        if not kf or not kb:
            continue
        mf, mb = med(kf[0]), med(kb[0])
        tf += mf; tb += mb; bf += by; bb += byb
        wtxt = ""
        if kw:
            mw = med(kw[0])
            tw += mw; bw += by
            wtxt = "%.1f | %.0f %%" % (mw, by / mw / 1e3 / 80)
        L.append("| %d | %d | %d,%d | %d->%d | %.1f | %.0f %% | %.1f | %.0f %% | %s |" % (
            b.idx, b.cexp, b.k, b.stride, b.h_in, b.h_out, mf, by / mf / 1e3 / 80, mb, byb / mb / 1e3 / 80, wtxt or " | "))
    if tf:
        L.append("| **all** | | | | %.1f | %.0f %% | %.1f | %.0f %% | %.1f | %.0f %% |" % (tf, bf / tf / 1e3 / 80, tb, bb / tb / 1e3 / 80, tw, (bw / tw / 1e3 / 80) if tw else 0))
    open(out, "w").write("\n".join(L) + "\n")
    print("wrote", out)


if __name__ == "__main__":
    main()
