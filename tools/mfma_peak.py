"""Achievable fp32 matrix-core rate of the box: a pure v_mfma_f32_16x16x4_f32 loop (tools/mfma_peak.hip, no memory traffic) at 1, 2, 4
and 8 resident waves per SIMD.  The GEMM roofline fractions in DESIGN.md quote the 157.3 TFLOP/s specification; this is what the
silicon sustains.   python tools/mfma_peak.py   (builds tools/libmfma_peak.so with hipcc on first use)"""
import ctypes
import os
import subprocess

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libmfma_peak.so")
if not os.path.exists(SO):
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO, os.path.join(HERE, "mfma_peak.hip")], check=True)
lib = ctypes.CDLL(SO)
out = torch.zeros(4096 * 256, device="cuda")
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for blocks in (256, 512, 1024, 2048):
    iters = 20000
    for _ in range(2):
        lib.mfma_probe(ctypes.c_void_p(out.data_ptr()), blocks, iters, stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    lib.mfma_probe(ctypes.c_void_p(out.data_ptr()), blocks, iters, stream)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print("16x16x4: %d waves/SIMD: %.1f TFLOP/s" % (blocks // 256, blocks * 4 * iters * 8 * 2048.0 / ms / 1e9))
    it32 = iters // 2
    lib.mfma_probe32(ctypes.c_void_p(out.data_ptr()), blocks, it32, stream)
    torch.cuda.synchronize()
    e0.record()
    lib.mfma_probe32(ctypes.c_void_p(out.data_ptr()), blocks, it32, stream)
    e1.record()
    torch.cuda.synchronize()
    print("32x32x2: %d waves/SIMD: %.1f TFLOP/s" % (blocks // 256, blocks * 4 * it32 * 4 * 4096.0 / e0.elapsed_time(e1) / 1e9))
