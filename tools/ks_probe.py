"""Phase stamps of conv1x1_ksplit_k (the long-K 1x1 convs of the 14x14 / 28x28 stage): needs a library built with -DKS_DBG
(conv_gemm.hip only) in MLIIS_HIP_LIB; prints, per shape, when the workgroups reach each phase (us from the first workgroup's start).

    MLIIS_HIP_LIB=/path/to/lib_ksdbg.so python tools/ks_probe.py"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mliis_amd import ops
d = torch.device("cuda:0")
for (N, H, K, Co) in [(8, 14, 672, 112), (8, 14, 480, 80), (8, 14, 480, 112), (8, 28, 240, 40)]:
    x = torch.randn(N, H, H, K, device=d); xsrc = torch.randn(N, H, H, K, device=d)
    w = torch.randn(1, 1, K, Co, device=d) * 0.05
    wt = w.permute(0, 1, 3, 2).contiguous().view(-1)
    g = torch.rand(N, K, device=d)
    y = torch.empty(N, H, H, Co, device=d)
    part = torch.zeros(1 << 20, device=d)
    name = ops.conv2d_kernel_name(N, H, H, K, Co, 1, True)
    st = torch.zeros(4096 * 8, dtype=torch.int64, device=d)
    os.environ["MLIIS_KS_STAMPS"] = str(st.data_ptr())
    res = []
    for it in range(6):
        x.copy_(xsrc)                      # the operand was just written by another kernel (as in the step)
        st.zero_()
        torch.cuda.synchronize()
        ops.conv2d_fwd(x, w, None, 1, out=y, wt=wt, x_scale=g, stats_part=part)
        torch.cuda.synchronize()
        s = st.cpu().numpy().reshape(-1, 8)
        if it == 0:
            from mliis_amd import _lib
            print("lib", _lib.LIB_PATH, "kernel", name, "nonzero stamps", int((s != 0).sum()), flush=True)
        s = s[s[:, 7] > 0]
        if s.shape[0] == 0:
            continue
        t0 = s[:, 0].min()
        rel = (s - t0) * 10.0 / 1000.0     # 100 MHz ticks -> us
        res.append(rel)
    if not res:
        print('no stamps'); continue
    rel = res[-1]
    print("%s  N=%d H=%d K=%d Co=%d: %d workgroups" % (name, N, H, K, Co, rel.shape[0]))
    names = ["start", "loads issued", "MFMA + LDS write", "barrier", "finish+store", "loop end", "stats", "end"]
    for k in range(8):
        col = rel[:, k]
        print("   %-18s median %6.2f  min %6.2f  max %6.2f us" % (names[k], np.median(col), col.min(), col.max()))
    print("   per-workgroup duration: median %.2f  max %.2f us;  last end %.2f us" % (np.median(rel[:, 7] - rel[:, 0]), (rel[:, 7] - rel[:, 0]).max(), rel[:, 7].max()))
