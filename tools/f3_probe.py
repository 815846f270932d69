"""The decoder's two 56x56 filter-gradient problems (3x3 224 -> 112; 3x3 dilated 128 -> 112 of the concat) as ONE batched launch: split
products with 128-channel workgroup tiles (round 5), with 256-channel tiles for the 224-channel problem (round 6), and the native fp32
instruction; microseconds per launch (hot), and the slabs of the two split-product forms compared bit for bit.
    python tools/f3_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mliis_amd import ops  # noqa: E402

d = torch.device("cuda:0")
torch.manual_seed(0)
N, H = 8, 56


def prob(Cbuf, Cin, Cout, k, dil):
    buf = torch.randn(N, H, H, Cbuf, device=d)
    dy = torch.randn(N, H, H, Cout, device=d)
    n = ops.lib.size("mliis_conv2d_bwd_filter_workspace_floats", N, H, H, Cin, Cout, k)
    return buf[..., :Cin], dy, k, dil, torch.zeros(n, device=d)


probs = (prob(224, 224, 112, 3, 1), prob(136, 128, 112, 3, 2))


def batch(wide):
    fb = ops.FilterBatch(d)
    fb.X3_WIDE = wide
    for a in probs:
        fb.add(*a)
    return fb


def timeit(fn, it=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / it


narrow, wide = batch(False), batch(True)
narrow.launch("fp32x3")
torch.cuda.synchronize()
ref = [p[4].clone() for p in probs]
for p in probs:
    p[4].zero_()
wide.launch("fp32x3")
torch.cuda.synchronize()
same = all(torch.equal(a, p[4]) for a, p in zip(ref, probs))
print("x3, 128-channel tiles %.1f us   x3, 256-channel tiles %.1f us   native fp32 %.1f us   slabs of the two x3 forms bit-identical: %s   workgroups %s / %s" % (
    timeit(lambda: narrow.launch("fp32x3")), timeit(lambda: wide.launch("fp32x3")), timeit(lambda: narrow.launch("fp32")), same,
    [t[2] for t in narrow.tables], [t[6][1] if t[6] else t[2] for t in wide.tables]))
