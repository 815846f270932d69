import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mliis_amd import ops
d = torch.device("cuda:0")
N, hd, H, S = 8, 56, 224, 5
torch.manual_seed(0)
small = (torch.randn(N, hd, hd, 2, device=d) * 3)
lab = (torch.rand(S, H, H, 1, device=d) > 0.6).float()
labels = torch.cat([1 - lab, lab], -1).contiguous()
idx = torch.tensor([i % S for i in range(N)], dtype=torch.int32, device=d)
out1, out2 = torch.zeros(4, device=d), torch.zeros(4, device=d)
ticket = torch.zeros(1, dtype=torch.int32, device=d)
lo = torch.empty(N, H, H, 2, device=d); dl = torch.empty_like(lo); ds1 = torch.empty_like(small); ds2 = torch.empty_like(small)
def chain():
    ops.resize_bilinear_fwd(small, (H, H), out=lo)
    ops.softmax_ce(lo, labels, idx, 0.0, False, 0.0, want_grad=True, dlogits=dl, out=out1)
    ops.resize_bilinear_bwd(dl, (hd, hd), out=ds1)
def fused():
    ops.head_ce_fused(small, labels, idx, (H, H), 0.0, ds2, out2)
def timeit(fn, it=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / it
chain(); fused(); torch.cuda.synchronize()
diff = (ds1 - ds2).abs()
print("max diff", diff.max().item(), "rel", (diff.max() / ds1.abs().max()).item(), "n diff", int((diff > 0).sum()), "of", diff.numel())
nz = (diff > 0).nonzero()
print(nz[:10].tolist())
print("loss", out1.tolist(), out2.tolist())
print("chain us", timeit(chain), "fused us", timeit(fused))
