"""Kernel time of the stem filter gradient at the config-2 shape (used to pick its block count)."""
import os, sys, torch
sys.path.insert(0, '/root/repo')
from mliis_amd import ops
from mliis_amd._lib import lib
d = torch.device("cuda:0")
x = torch.rand(8, 224, 224, 3, device=d) * 255
dz = torch.randn(8, 112, 112, 32, device=d)
idx = torch.arange(8, dtype=torch.int32, device=d)
part = torch.empty(lib.size("mliis_stem_conv_bwd_filter_workspace_floats", 8, 224, 224, 32) + 16, device=d)
fn = lambda: ops.stem_conv_bwd_filter(x, dz, idx, partial=part)
for _ in range(3): fn()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(10): fn()
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): g.replay()
e1.record(); torch.cuda.synchronize()
print("stem_bwd_filter %.1f us" % (e0.elapsed_time(e1) * 1e3 / 50))
