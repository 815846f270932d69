"""Split-operand (MLIIS_PREC_F32X3) dense convs beside the native fp32 instances: error against a float64 reference and launch time.
python tools/x3_probe.py [iters [shape]]   (prints one row per shape and direction)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mliis_amd import ops  # noqa: E402

SHAPES = {"rsd2.fuse": (3, 1, 56, 224, 112), "rsd2.br1": (3, 2, 56, 136, 112), "rsd4.fuse": (3, 1, 14, 224, 112), "rsd4.br1": (3, 2, 14, 224, 112),
          "even512": (3, 1, 64, 224, 112)}


def ref_conv(x, w, dil):
    xx = x.double().permute(0, 3, 1, 2)
    ww = w.double().permute(3, 2, 0, 1)
    k = w.shape[0]
    pad = dil * (k // 2)
    return torch.nn.functional.conv2d(xx, ww, padding=pad, dilation=dil).permute(0, 2, 3, 1)


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    only = sys.argv[2] if len(sys.argv) > 2 else None    # one shape only (counter passes)
    d = torch.device("cuda:0")
    torch.manual_seed(0)
    N = 8
    for name, (k, dil, h, ci, co) in SHAPES.items():
        if only is not None and name != only:
            continue
        x = torch.randn(N, h, h, ci, device=d)
        w = torch.randn(k, k, ci, co, device=d) * 0.05
        wt = w.permute(0, 1, 3, 2).contiguous().view(-1)
        dy = torch.randn(N, h, h, co, device=d)
        ref = ref_conv(x, w, dil)
        gflop = 2.0 * N * h * h * k * k * ci * co * 1e-9
        for mode in ("fwd", "bwd_data"):
            row = [name, mode]
            for prec in ("fp32", "x3k"):
                if prec == "x3k" and mode == "fwd":
                    y = torch.empty(N, h, h, co, device=d)
                    im = ops.x3_image_of(w, "fwd")
                    fn = lambda: ops.conv2d_fwd_x3(x, im, k, co, None, dil, out=y)  # noqa: E731
                    fn()
                    err = ((y.double() - ref).abs().max() / ref.abs().max()).item()
                elif prec == "x3k":
                    dx = torch.empty_like(x)
                    im = ops.x3_image_of(w, "bwd")
                    fn = lambda: ops.conv2d_bwd_data_x3(dy, im, k, ci, dil, out=dx)  # noqa: E731
                    fn()
                    xr = x.double().clone().requires_grad_(True)
                    (ref_conv(xr, w, dil) * dy.double()).sum().backward()
                    err = ((dx.double() - xr.grad).abs().max() / xr.grad.abs().max()).item()
                elif mode == "fwd":
                    y = torch.empty(N, h, h, co, device=d)
                    fn = lambda: ops.conv2d_fwd(x, w, None, dil, out=y, wt=wt, precision=prec)  # noqa: E731
                    fn()
                    err = ((y.double() - ref).abs().max() / ref.abs().max()).item()
                else:
                    dx = torch.empty_like(x)
                    fn = lambda: ops.conv2d_bwd_data(dy, w, dil, out=dx, precision=prec)  # noqa: E731
                    fn()
                    dyy = dy.double().requires_grad_(False)
                    xr = x.double().clone().requires_grad_(True)
                    (ref_conv(xr, w, dil) * dyy).sum().backward()
                    err = ((dx.double() - xr.grad).abs().max() / xr.grad.abs().max()).item()
                us = timeit(fn, iters)
                row += ["%s err %.2e  %.1f us  %.1f TF" % (prec, err, us, gflop / us * 1e3)]
            print("  ".join(row), flush=True)


if __name__ == "__main__":
    main()
