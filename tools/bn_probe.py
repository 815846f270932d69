"""Micro-benchmark of the fused BN apply / backward kernels on one tensor shape as a function of the number of statistics partials
the consumer has to fold (the producer's row-block count).  python tools/bn_probe.py [rows] [C]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mliis_amd import ops  # noqa: E402


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters // 10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (iters // 10 * 10)


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100352
    C = int(sys.argv[2]) if len(sys.argv) > 2 else 96
    d = torch.device("cuda:0")
    x = torch.randn(8, rows // 8, 1, C, device=d)
    dy = torch.randn_like(x)
    y = torch.empty_like(x)
    dx = torch.empty_like(x)
    gamma, beta = torch.ones(C, device=d), torch.zeros(C, device=d)
    m, r = torch.empty(C, device=d), torch.empty(C, device=d)
    mb = x.numel() * 4 / 1e6
    for nblk in (8, 64, 196, 784, 1568):
        part = torch.randn(nblk * 2 * C + 16, device=d).abs()
        t_f = timeit(lambda: ops.bn_apply_fused(x, part, nblk, m, r, gamma, beta, post_swish=True, out=y))
        print("nblk %5d  apply %.1f us (%.2f TB/s of %d MB)" % (nblk, t_f, 2 * mb / t_f, 2 * mb))
    t_b = timeit(lambda: ops.bn_bwd(x, dy, m, r, gamma, beta, False, True, dx=dx))
    print("bwd (colreduce + apply) %.1f us (%.2f TB/s of %d MB)" % (t_b, 5 * mb / t_b, 5 * mb))


if __name__ == "__main__":
    main()
