"""Depthwise filter-gradient kernel time per EfficientLab-6-3 layer (kernel only, slabs left in the workspace), for the tuning
planner in dwconv.hip (dw_filter_geom); used to pick its block target."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mliis_amd import ops, spec  # noqa: E402
from mliis_amd._lib import lib  # noqa: E402


def main():
    d = torch.device("cuda:0")
    arch = spec.derive()
    N = 8
    out = []
    for b in arch.blocks:
        C, k, s, hi, ho = b.cexp, b.k, b.stride, b.h_in, b.h_out
        x = torch.randn(N, hi, hi, C, device=d)
        dy = torch.randn(N, ho, ho, C, device=d)
        part = torch.empty(lib.size("mliis_dwconv_bwd_filter_workspace_floats", N, hi, hi, C, k, s) + 16, device=d)
        fn = lambda: ops.dwconv_bwd_filter(x, dy, k, s, partial=part)  # noqa: E731
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        out.append("b%d(k%ds%d,%d,C%d) %.1f" % (b.idx, k, s, hi, C, e0.elapsed_time(e1) * 1e3 / 100))
    print(" | ".join(out))


if __name__ == "__main__":
    main()
