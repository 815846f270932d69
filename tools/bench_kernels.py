"""Per-layer kernel timings on the GPU (EfficientLab-6-3 shapes, N=8): depthwise fwd/bwd vs the HBM roofline (cold operands:
rotating copies, through the Python wrappers -- add ~3 us of host time per call to a launch that is shorter than that) and the
dense convs vs the fp32-MFMA roofline.  Usage: python tools/bench_kernels.py [--n 8] [--iters 20] [--dw-only]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mliis_amd import ops, spec  # noqa: E402

HBM_PEAK = 8.0e12
MFMA_F32_PEAK = 157.3e12


def timeit(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--out", default=None)
    ap.add_argument("--dense-only", action="store_true")
    ap.add_argument("--dw-only", action="store_true")
    ap.add_argument("--old-dw", action="store_true", help="time the depthwise kernels of csrc/dwconv.hip instead of the row-marching ones")
    a = ap.parse_args()
    d = torch.device("cuda:0")
    arch = spec.derive()
    N = a.n
    res = {"depthwise": [], "dense": []}
    tot = dict(fwd_t=0, fwd_b=0, bwd_t=0, bwd_b=0)
    for b in ([] if a.dense_only else arch.blocks):
        C, k, s, hi, ho = b.cexp, b.k, b.stride, b.h_in, b.h_out
        # COLD operands: rotate over enough copies of the activation tensors that one is touched again only after > 320 MB of other
        # traffic (the 256 MiB Infinity Cache is flushed in between), as bench.py's depthwise_hbm does
        per = 4 * N * C * (hi * hi + ho * ho)
        R_ = max(1, -(-320 * 2 ** 20 // per))
        xs = [torch.randn(N, hi, hi, C, device=d) for _ in range(R_)]
        w = torch.randn(k, k, C, 1, device=d)
        ys = [torch.empty(N, ho, ho, C, device=d) for _ in range(R_)]
        dys = [torch.randn(N, ho, ho, C, device=d) for _ in range(R_)]
        dxs = [torch.empty_like(xs[0]) for _ in range(R_)]
        dw = torch.empty_like(w)
        ctr = [0]

        def rot(fn):
            def call():
                i = ctr[0] % R_
                ctr[0] += 1
                fn(i)
            return call
        if a.old_dw:   # the sliding-window / tile kernels of csrc/dwconv.hip (rounds 1-2; the skip decoder still uses them)
            tf = timeit(rot(lambda i: ops.dwconv_fwd(xs[i], w, s, out=ys[i])), a.iters)
            tbd = timeit(rot(lambda i: ops.dwconv_bwd_data(dys[i], w, s, (hi, hi), out=dxs[i])), a.iters)
            tbf = timeit(rot(lambda i: ops.dwconv_bwd_filter(xs[i], dys[i], k, s, out=dw)), a.iters)
        else:          # the row-marching kernels (csrc/dwmarch.hip): forward with the batch norm + swish applied while staging (statistics
            #            given), backward = ONE pass (dx + filter-gradient slabs + the batch norm's stage-1 sums)
            gam, bet, mean, rstd = (torch.rand(C, device=d) + 0.5 for _ in range(4))
            slabs = torch.empty(ops.dwconv_bn_bwd_blocks(N, hi, hi, C, k, s) * k * k * C, device=d)
            part = torch.empty(1 << 22, device=d)
            tf = timeit(rot(lambda i: ops.dwconv_bn_fwd(xs[i], w, s, bn=(gam, bet, mean, rstd, None, None), out=ys[i], stats_part=part)), a.iters)
            tbd = timeit(rot(lambda i: ops.dwconv_bn_bwd(dys[i], xs[i], w, s, bn=(mean, rstd, gam, bet), out=dxs[i], dw_part=slabs, bn_part=part)), a.iters)
            tbf = 0.0
        x, y = xs[0], ys[0]
        fb = 4 * (x.numel() + y.numel() + w.numel())
        bb = 4 * (2 * x.numel() + y.numel() + 2 * w.numel())
        res["depthwise"].append(dict(block=b.idx, C=C, k=k, s=s, h=hi, fwd_us=tf * 1e6, bwd_data_us=tbd * 1e6, bwd_filter_us=tbf * 1e6,
                                     fwd_frac=fb / tf / HBM_PEAK, bwd_frac=bb / (tbd + tbf) / HBM_PEAK))
        tot["fwd_t"] += tf
        tot["fwd_b"] += fb
        tot["bwd_t"] += tbd + tbf
        tot["bwd_b"] += bb
        print("dw b%-2d C=%-3d k%d s%d h=%-3d fwd %7.1f us (%4.1f%% HBM)  bwd%s %7.1f%s (%4.1f%% HBM)" % (
            b.idx, C, k, s, hi, tf * 1e6, 100 * fb / tf / HBM_PEAK, "_data" if a.old_dw else " (one pass)", tbd * 1e6,
            ("  bwd_filter %7.1f" % (tbf * 1e6)) if a.old_dw else "", 100 * bb / (tbd + tbf) / HBM_PEAK), flush=True)
    if not a.dense_only:
      res["depthwise_total"] = dict(fwd_us=tot["fwd_t"] * 1e6, bwd_us=tot["bwd_t"] * 1e6, fwd_frac=tot["fwd_b"] / tot["fwd_t"] / HBM_PEAK,
                                  bwd_frac=tot["bwd_b"] / tot["bwd_t"] / HBM_PEAK)
    if not a.dense_only:
      print("dw total fwd %.1f us (%.1f%% of 8 TB/s)  bwd %.1f us (%.1f%%)" % (tot["fwd_t"] * 1e6, 100 * res["depthwise_total"]["fwd_frac"],
                                                                          tot["bwd_t"] * 1e6, 100 * res["depthwise_total"]["bwd_frac"]), flush=True)
    dense = [("b1.exp", 1, 1, 112, 16, 96), ("b2.exp", 1, 1, 56, 24, 144), ("b4.exp", 1, 1, 28, 40, 240), ("b6.exp", 1, 1, 14, 80, 480),
             ("b9.proj", 1, 1, 14, 672, 112), ("rsd4.br1", 3, 2, 14, 224, 112), ("rsd4.fuse", 3, 1, 14, 224, 112),
             ("rsd2.br0", 1, 1, 56, 136, 112), ("rsd2.br1", 3, 2, 56, 136, 112), ("rsd2.fuse", 3, 1, 56, 224, 112)]
    for name, k, dil, h, ci, co in ([] if a.dw_only else dense):
        x = torch.randn(N, h, h, ci, device=d)
        w = torch.randn(k, k, ci, co, device=d) * 0.05
        bias = torch.zeros(co, device=d)
        y = torch.empty(N, h, h, co, device=d)
        dy = torch.randn(N, h, h, co, device=d)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        fl = 2.0 * N * h * h * k * k * ci * co
        wt = w.permute(0, 1, 3, 2).contiguous().view(-1)
        tf = timeit(lambda: ops.conv2d_fwd(x, w, bias, dil, out=y, wt=wt), a.iters)
        tbd = timeit(lambda: ops.conv2d_bwd_data(dy, w, dil, out=dx), a.iters)
        tbf = timeit(lambda: ops.conv2d_bwd_filter(x, dy, k, dil, out=dw), a.iters)
        res["dense"].append(dict(name=name, fwd_us=tf * 1e6, bwd_data_us=tbd * 1e6, bwd_filter_us=tbf * 1e6, gflop=fl / 1e9,
                                 fwd_tf=fl / tf / 1e12, bwd_data_tf=fl / tbd / 1e12, bwd_filter_tf=fl / tbf / 1e12))
        print("%-10s %7.3f GFLOP  fwd %8.1f us (%5.1f TF, %4.1f%%)  bwd_data %8.1f us (%5.1f TF)  bwd_filter %8.1f us (%5.1f TF)" % (
            name, fl / 1e9, tf * 1e6, fl / tf / 1e12, 100 * fl / tf / MFMA_F32_PEAK, tbd * 1e6, fl / tbd / 1e12, tbf * 1e6, fl / tbf / 1e12), flush=True)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
