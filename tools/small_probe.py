"""Probe of the small-map fused MBConv kernels (mbconv_small.hip): issues the forward / backward launch of one 14x14 layer in a loop so
that `rocprofv3 --kernel-trace --stats` / `--pmc` passes see them in isolation.   python tools/small_probe.py [C] [k] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mliis_amd import ops  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 672
k = int(sys.argv[2]) if len(sys.argv) > 2 else 5
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
N, H = 8, 14
d = torch.device("cuda:0")
g = torch.Generator(device=d).manual_seed(0)
r = lambda *s: torch.randn(*s, device=d, generator=g)  # noqa: E731
z0, w = r(N, H, H, C), r(k, k, C, 1) * 0.3
part = torch.zeros(1 << 18, device=d)
nblk = ops.bn_stats_partial(z0, False, part)
vec = lambda v=0.0: torch.full((C,), v, device=d)  # noqa: E731
st = [vec() for _ in range(4)]
g0, b0, g1, b1 = vec(1.0), vec(), vec(1.0), vec()
z1, a1, s = torch.empty_like(z0), torch.empty_like(z0), torch.empty(N, C, device=d)
da2, gate, cadd = r(N, H, H, C), torch.sigmoid(r(N, C)), r(N, C) * 0.01
outs = [vec() for _ in range(2)] + [torch.zeros(k, k, C, 1, device=d)] + [vec() for _ in range(2)] + [torch.empty_like(z0)]
torch.cuda.synchronize()
for V in [0] + [2, 4]:    # 0 = the planner's choice
    for _warm in range(2):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        for _ in range(reps):
            ops.mbconv_dw_fwd_small(z0, part, nblk, (g0, b0, st[0], st[1], None, None), w, (g1, b1, st[2], st[3], None, None), z1, a1, s, group_width=V)
        e[1].record()
        for _ in range(reps):
            ops.mbconv_dw_bwd_small(da2, gate, cadd, z1, (st[2], st[3], g1, b1), w, z0, (st[0], st[1], g0, b0), outs[0], outs[1], outs[2], outs[3], outs[4],
                                    outs[5], group_width=V)
        e[2].record()
        torch.cuda.synchronize()
    print("C=%d k=%d V=%d (%s): fwd %.1f us, bwd %.1f us per launch (hot, host-issued back to back)" % (
        C, k, V or ops.mbconv_dw_small_group_width(C, k), "planner" if V == 0 else "forced", 1e3 * e[0].elapsed_time(e[1]) / reps, 1e3 * e[1].elapsed_time(e[2]) / reps))
