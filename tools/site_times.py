"""Per-launch-site kernel time of one inner step: records every C-ABI call of an eager step, then re-issues each call `burst` times
back to back between two HIP events on the learner's stream (the kernels' own duration, caches warm as they are inside the step).
    python tools/site_times.py [--image-size 224] [--batch 8] [--top 40] [--aspp]
(Re-issuing accumulating calls changes buffer contents; the numbers are timings only -- run it in its own process.)"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mliis_amd._lib import lib  # noqa: E402
from mliis_amd.learner import Learner  # noqa: E402
from mliis_amd.metaseg import synthetic_task  # noqa: E402

# positions of a few shape arguments worth printing, per entry point
SHAPE = {"mliis_conv2d_fwd": (8, 9, 10, 13, 14, 15, 16), "mliis_conv2d_bwd_data": (5, 6, 7, 10, 11, 12, 13),
         "mliis_conv2d_bwd_filter": (6, 7, 8, 11, 12, 13, 14), "mliis_dwconv_fwd": (3, 4, 5, 6, 7, 8), "mliis_dwconv_bwd_data": (3, 4, 5, 6, 7, 8),
         "mliis_dwconv_bwd_filter": (3, 4, 5, 6, 7, 8), "mliis_bn_apply_fused": None, "mliis_bn_bwd": None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--image-size", type=int, default=224)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--burst", type=int, default=20)
    ap.add_argument("--top", type=int, default=45)
    ap.add_argument("--aspp", action="store_true")
    a = ap.parse_args()
    L = Learner(image_size=a.image_size, use_graph=False, spatial_pyramid_pooling=a.aspp)
    x, y = synthetic_task(5, a.image_size, seed=0)
    L.load_task(x, y)
    idx = [i % 5 for i in range(a.batch)]
    for _ in range(2):
        L.inner_step(idx)
    L.synchronize()
    lib.trace = []
    L.inner_step(idx)
    L.synchronize()
    calls, lib.trace = lib.trace, None
    dll = lib.load()
    rows = []
    with torch.cuda.stream(L.stream):
        for i, (name, args) in enumerate(calls):
            if name.startswith("mliis_graph"):
                continue
            fn = getattr(dll, name)
            fn(*args)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(L.stream)
            for _ in range(a.burst):
                fn(*args)
            e1.record(L.stream)
            e1.synchronize()
            pos = SHAPE.get(name)
            ints = [v for v in args if isinstance(v, int) and not isinstance(v, bool)]
            shape = tuple(args[j] for j in pos) if pos else tuple(ints[:6])
            rows.append((e0.elapsed_time(e1) * 1e3 / a.burst, i, name.replace("mliis_", ""), shape))
    total = sum(r[0] for r in rows)
    print("inner step: %d C-ABI calls, %.0f us of kernel time (burst-timed per site)" % (len(rows), total))
    fam = collections.defaultdict(lambda: [0.0, 0])
    for us, _, name, _ in rows:
        fam[name][0] += us
        fam[name][1] += 1
    print("\n%-28s %6s %9s %7s" % ("entry point", "calls", "us/step", "share"))
    for name, (us, n) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
        print("%-28s %6d %9.1f %6.1f%%" % (name, n, us, 100 * us / total))
    print("\ntop sites:\n%5s %-26s %8s  %s" % ("#", "entry point", "us", "leading integer arguments"))
    for us, i, name, shape in sorted(rows, reverse=True)[: a.top]:
        print("%5d %-26s %8.1f  %s" % (i, name, us, shape))


if __name__ == "__main__":
    main()
