"""Measures the per-kernel cost of tiny kernels on this GPU: a dependent chain replayed from a captured graph, the same chain launched
eagerly, and independent kernels on several captured streams (how much of a ~5 us 'duration' is dependent-launch latency)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mliis_amd import ops  # noqa: E402


def main():
    d = torch.device("cuda:0")
    x = torch.zeros(64, device=d)
    y = torch.zeros(64, device=d)
    n = 1000
    fn = lambda: ops.axpby(1.0, x, 0.5, y)  # noqa: E731
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print("eager chain      : %.2f us / kernel" % ((time.perf_counter() - t0) / n * 1e6))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    print("graph chain      : %.2f us / kernel" % ((time.perf_counter() - t0) / (5 * n) * 1e6))
    for ns in (2, 4, 8):
        ys = [torch.zeros(64, device=d) for _ in range(ns)]
        streams = [torch.cuda.Stream() for _ in range(ns)]
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            cur = torch.cuda.current_stream()
            ev = torch.cuda.Event()
            ev.record(cur)
            for s_, yy in zip(streams, ys):
                s_.wait_event(ev)
                with torch.cuda.stream(s_):
                    for _ in range(n // ns):
                        ops.axpby(1.0, x, 0.5, yy)
                e2 = torch.cuda.Event()
                e2.record(s_)
                cur.wait_event(e2)
        g2.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            g2.replay()
        torch.cuda.synchronize()
        print("graph, %d branches: %.2f us / kernel" % (ns, (time.perf_counter() - t0) / (5 * (n // ns) * ns) * 1e6))


if __name__ == "__main__":
    main()
