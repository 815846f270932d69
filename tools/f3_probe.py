import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mliis_amd import ops
d = torch.device("cuda:0")
torch.manual_seed(0)
N, H = 8, 56
def prob(Cbuf, Cin, Cout, k, dil):
    buf = torch.randn(N, H, H, Cbuf, device=d); dy = torch.randn(N, H, H, Cout, device=d)
    n = ops.lib.size("mliis_conv2d_bwd_filter_workspace_floats", N, H, H, Cin, Cout, k)
    return buf[..., :Cin], dy, k, dil, torch.empty(n, device=d)
fb = ops.FilterBatch(d)
for a in (prob(224, 224, 112, 3, 1), prob(136, 128, 112, 3, 2)):
    fb.add(*a)
def timeit(fn, it=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / it
print("x3 us", timeit(lambda: fb.launch("fp32x3")), "native us", timeit(lambda: fb.launch("fp32")), "tables", [(t[1], t[2], t[3], t[4]) for t in fb.tables])
