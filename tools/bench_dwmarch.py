"""Depthwise layers of EfficientLab-6-3 (blocks 0-5, or all with --all) through the row-marching kernels (csrc/dwmarch.hip) and through
the sliding-window / tile kernels they replace (csrc/dwconv.hip), COLD: every launch works on the next of enough rotating copies of
its activation operands that a tensor is touched again only after > 320 MB of other traffic (the 256 MiB Infinity Cache is
flushed), straight through ctypes (no Python wrapper between two launches).  Beside each: a plain device copy of the same bytes.

    python tools/bench_dwmarch.py [--n 8] [--reps 40] [--all] [--out file.json]       (MLIIS_DWM_TARGET=... : workgroups per launch)

Algorithmic bytes: fwd 4 (in + out + k^2 C); bwd 4 (2 in + out + 2 k^2 C)."""
import argparse
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mliis_amd import spec  # noqa: E402
from mliis_amd._lib import lib  # noqa: E402

HBM = 8.0e12
FLUSH = 320e6


def burst(calls, reps, stream):
    """calls: list of zero-argument callables (one per rotating copy); returns microseconds per call."""
    for f in calls:
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for r in range(reps):
        calls[r % len(calls)]()
    e1.record(stream)
    e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8)
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--all", action="store_true")
    ap.add_argument("--out", default=None)
    ap.add_argument("--blocks", default=None, help="comma-separated block indices (default: every large-map block)")
    ap.add_argument("--march-only", action="store_true", help="skip the sliding-window kernels and the copies")
    a = ap.parse_args()
    d = torch.device("cuda:0")
    dll = lib.load()
    st = torch.cuda.current_stream()
    sp = st.cuda_stream
    arch = spec.derive()
    N = a.n
    rows = []
    tot = {}
    for b in arch.blocks:
        if not b.executed or (b.h_out <= 14 and b.h_in <= 14 and not a.all):
            continue
        if a.blocks is not None and str(b.idx) not in a.blocks.split(","):
            continue
        Cc, k, s, hi, ho = b.cexp, b.k, b.stride, b.h_in, b.h_out
        ein, eout, ew = N * hi * hi * Cc, N * ho * ho * Cc, k * k * Cc
        fb, bb = 4.0 * (ein + eout + ew), 4.0 * (2 * ein + eout + 2 * ew)
        copies = max(2, int(FLUSH / fb) + 1)
        reps = max(a.reps, 2 * copies)
        z = [torch.randn(ein, device=d) for _ in range(copies)]
        y = [torch.randn(eout, device=d) for _ in range(copies)]
        dx = [torch.empty(ein, device=d) for _ in range(copies)]
        w = torch.randn(ew, device=d)
        gamma, beta, mean, rstd = (torch.rand(Cc, device=d) + 0.5 for _ in range(4))
        part = torch.empty(4 << 20, device=d)
        slabs = torch.empty(8 << 20, device=d)
        nb = C.c_int(0)
        p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731

        # argument tuples are built ONCE: converting two dozen ctypes arguments per call costs more host time than these kernels run
        def mk(fn, argsets):
            def one(args):
                def call():
                    rc = fn(*args)
                    if rc:
                        raise RuntimeError(dll.mliis_last_error())
                return call
            return [one(tuple(x)) for x in argsets]
        nbp = C.byref(nb)
        new_f = mk(dll.mliis_dwconv_bn_fwd, [(p(z[i]), None, 0, p(gamma), p(beta), p(mean), p(rstd), None, None, 1e-3, 0.99, p(w), p(y[i]), N, hi, hi, Cc, k, s,
                                             p(part), part.numel(), nbp, 0, 0, sp) for i in range(copies)])
        new_b = mk(dll.mliis_dwconv_bn_bwd, [(p(y[i]), p(z[i]), p(mean), p(rstd), p(gamma), p(beta), p(w), p(dx[i]), None, N, hi, hi, Cc, k, s, p(slabs),
                                             slabs.numel(), p(part), part.numel(), nbp, 0, 0, sp) for i in range(copies)])
        old_f = mk(dll.mliis_dwconv_fwd, [(p(z[i]), p(w), p(y[i]), N, hi, hi, Cc, k, s, p(part), part.numel(), nbp, sp) for i in range(copies)])
        old_bd = mk(dll.mliis_dwconv_bwd_data_bn, [(p(y[i]), p(w), p(dx[i]), N, hi, hi, Cc, k, s, p(z[i]), p(mean), p(rstd), p(gamma), p(beta), p(part),
                                                   part.numel(), nbp, sp) for i in range(copies)])
        old_bf = mk(dll.mliis_dwconv_bwd_filter, [(p(z[i]), p(y[i]), None, N, hi, hi, Cc, k, s, p(slabs), slabs.numel(), sp) for i in range(copies)])
        half_f, half_b = int(fb // 8), int(bb // 8)
        src = [torch.randn(max(half_f, half_b), device=d) for _ in range(copies)]
        dst = [torch.empty(max(half_f, half_b), device=d) for _ in range(copies)]
        cp_f = [lambda i=i: dst[i][:half_f].copy_(src[i][:half_f]) for i in range(copies)]
        cp_b = [lambda i=i: dst[i][:half_b].copy_(src[i][:half_b]) for i in range(copies)]
        with torch.cuda.stream(st):
            r = dict(block=b.idx, C=Cc, k=k, s=s, h=hi, fwd_MB=fb / 1e6, bwd_MB=bb / 1e6,
                     fwd_blocks=int(dll.mliis_dwconv_bn_fwd_blocks(N, hi, hi, Cc, k, s)), bwd_blocks=int(dll.mliis_dwconv_bn_bwd_blocks(N, hi, hi, Cc, k, s)),
                     march_fwd_us=burst(new_f, reps, st), march_bwd_us=burst(new_b, reps, st))
            skip = a.march_only
            r.update(old_fwd_us=1.0 if skip else burst(old_f, reps, st), old_bwd_us=1.0 if skip else burst(old_bd, reps, st) + burst(old_bf, reps, st),
                     copy_fwd_us=1.0 if skip else burst(cp_f, reps, st), copy_bwd_us=1.0 if skip else burst(cp_b, reps, st))
        r["march_fwd_frac"] = fb / (r["march_fwd_us"] * 1e-6) / HBM
        r["march_bwd_frac"] = bb / (r["march_bwd_us"] * 1e-6) / HBM
        r["old_fwd_frac"] = fb / (r["old_fwd_us"] * 1e-6) / HBM
        r["old_bwd_frac"] = bb / (r["old_bwd_us"] * 1e-6) / HBM
        rows.append(r)
        for k_ in ("march_fwd_us", "march_bwd_us", "old_fwd_us", "old_bwd_us", "copy_fwd_us", "copy_bwd_us", "fwd_MB", "bwd_MB"):
            tot[k_] = tot.get(k_, 0.0) + r[k_]
        print("b%-2d C=%-3d k%d s%d h=%-3d | fwd %5.1f MB: march %6.1f us %4.1f%% (%4d wg)  old %6.1f us %4.1f%%  copy %6.1f us | bwd %5.1f MB: march %6.1f us %4.1f%% (%4d wg)  "
              "old %6.1f us %4.1f%%  copy %6.1f us" % (b.idx, Cc, k, s, hi, fb / 1e6, r["march_fwd_us"], 100 * r["march_fwd_frac"], r["fwd_blocks"], r["old_fwd_us"],
                                                       100 * r["old_fwd_frac"], r["copy_fwd_us"], bb / 1e6, r["march_bwd_us"], 100 * r["march_bwd_frac"], r["bwd_blocks"],
                                                       r["old_bwd_us"], 100 * r["old_bwd_frac"], r["copy_bwd_us"]), flush=True)
        del z, y, dx, src, dst
    fr = lambda mb, us: mb * 1e6 / (us * 1e-6) / HBM   # noqa: E731
    print("total fwd %.1f MB: march %.1f us (%.1f%% of 8 TB/s)  old %.1f us (%.1f%%)  copy %.1f us | bwd %.1f MB: march %.1f us (%.1f%%)  old %.1f us (%.1f%%)  copy %.1f us" % (
        tot["fwd_MB"], tot["march_fwd_us"], 100 * fr(tot["fwd_MB"], tot["march_fwd_us"]), tot["old_fwd_us"], 100 * fr(tot["fwd_MB"], tot["old_fwd_us"]), tot["copy_fwd_us"],
        tot["bwd_MB"], tot["march_bwd_us"], 100 * fr(tot["bwd_MB"], tot["march_bwd_us"]), tot["old_bwd_us"], 100 * fr(tot["bwd_MB"], tot["old_bwd_us"]), tot["copy_bwd_us"]))
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        json.dump(dict(n=N, target=os.environ.get("MLIIS_DWM_TARGET"), layers=rows, total=tot), open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
