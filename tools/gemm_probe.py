"""Runs one dense conv shape repeatedly (for rocprofv3 --pmc / --kernel-trace).  python tools/gemm_probe.py [name] [iters] [mode]
names as in tools/bench_kernels.py; mode in fwd | bwd_data | bwd_filter."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mliis_amd import ops  # noqa: E402

SHAPES = {"rsd2.fuse": (3, 1, 56, 224, 112), "rsd2.br1": (3, 2, 56, 136, 112), "rsd2.br0": (1, 1, 56, 136, 112), "rsd4.fuse": (3, 1, 14, 224, 112),
          "b2.exp": (1, 1, 56, 24, 144), "b1.exp": (1, 1, 112, 16, 96), "b9.proj": (1, 1, 14, 672, 112),
          "even512": (3, 1, 64, 224, 112), "big128": (3, 1, 128, 224, 112)}   # 512 tiles of 64 rows: two whole tiles per CU, no stream-K remainder


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "rsd2.fuse"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    mode = sys.argv[3] if len(sys.argv) > 3 else "fwd"
    k, dil, h, ci, co = SHAPES[name]
    d = torch.device("cuda:0")
    N = 8
    x = torch.randn(N, h, h, ci, device=d)
    w = torch.randn(k, k, ci, co, device=d) * 0.05
    wt = w.permute(0, 1, 3, 2).contiguous().view(-1)
    y = torch.empty(N, h, h, co, device=d)
    dy = torch.randn(N, h, h, co, device=d)
    dx = torch.empty_like(x)
    dw = torch.empty_like(w)
    for _ in range(iters):
        if mode == "fwd":
            ops.conv2d_fwd(x, w, None, dil, out=y, wt=wt)
        elif mode == "bwd_data":
            ops.conv2d_bwd_data(dy, w, dil, out=dx)
        else:
            ops.conv2d_bwd_filter(x, dy, k, dil, out=dw)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
