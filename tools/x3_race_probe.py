"""Reproduces the multi-stream interference found in round 5 (profiles/r05_notes.md): small kernels of one stream (the head's bilinear
resize with a freshly allocated output, the final layer's filter-gradient reduction) while the split-product convs ("c"), their
filter gradients ("f") or both ("cf") run on ANOTHER stream; every victim result is compared bit for bit with its result alone.

    python tools/x3_race_probe.py [c | f | cf | none]

Measured on MI355X (gpurun boxes, ROCm 7.2): "cf": 40-50 of 360 victim launches wrong (always the first launches behind a burst of
aggressor launches, one FP component of groups of 16 lanes); "c", "f", "none": 0.  Before conv_x3_k claimed its CU's register file, "c"
alone: 150-350 of 1200.  The concurrent-lanes variant of the meta-learner therefore runs the native fp32 kernels (reptile.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mliis_amd import ops
d = torch.device("cuda:0")
torch.manual_seed(0)
N, H = 8, 56
x = torch.randn(N, H, H, 224, device=d); w = torch.randn(3, 3, 224, 112, device=d) * 0.02
w1 = torch.randn(3, 3, 136, 112, device=d) * 0.02
dy = torch.randn(N, H, H, 112, device=d)
imf, imb, imb1 = ops.x3_image_of(w, "fwd"), ops.x3_image_of(w, "bwd"), ops.x3_image_of(w1, "bwd")
ws2 = ops.Workspace(d, 1 << 25)
dx1 = torch.zeros(N, H, H, 136, device=d)
nfl = ops.lib.size("mliis_conv2d_bwd_filter_workspace_floats", N, H, H, 224, 112, 3)
pfl = torch.zeros(nfl, device=d); fb = ops.FilterBatch(d); fb.add(x, dy, 3, 1, pfl)
ws1 = ops.Workspace(d, 1 << 22)
dec, dsmall, small = torch.randn(N, H, H, 112, device=d), torch.randn(N, H, H, 2, device=d), torch.randn(N, H, H, 2, device=d)
dwf, dbf = torch.zeros(112 * 2, device=d), torch.zeros(2, device=d)
def victims():
    ops.final_conv_bwd_filter(dec, dsmall, None, dw=dwf.view(1, 1, 112, 2), db=dbf, ws=ws1)
    return dwf.clone(), ops.resize_bilinear_fwd(small, (224, 224)).clone()
ref = victims(); torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
AGG = sys.argv[1] if len(sys.argv) > 1 else "cf"
for it in range(60):
    with torch.cuda.stream(s2):
        for _ in range(2):
            if "c" in AGG:
                ops.conv2d_fwd_x3(x, imf, 3, 112, None, 1, ws=ws2)
                ops.conv2d_bwd_data_x3(dy, imb1, 3, 136, 2, out=dx1, accumulate=True, ws=ws2)
                ops.conv2d_bwd_data_x3(dy, imb, 3, 224, 1, ws=ws2)
            if "f" in AGG:
                fb.launch("fp32x3")
    with torch.cuda.stream(s1):
        for j in range(6):
            got = victims()
            for i in range(2):
                if not torch.equal(got[i], ref[i]):
                    dd = (got[i] - ref[i]).flatten(); nz = dd.nonzero().flatten()
                    print("iter", it, "call", j, "victim", i, "mismatching", nz.numel(), "idx", nz[:4].tolist(), nz[-2:].tolist(), "got", got[i].flatten()[nz[:3]].tolist(), "ref", ref[i].flatten()[nz[:3]].tolist(), flush=True)
    torch.cuda.synchronize()
print("done", AGG)
