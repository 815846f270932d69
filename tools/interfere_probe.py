"""What goes wrong in a small kernel of one stream while the split-product kernels run on another.

Self-checking victims (tools/interfere.hip: every loaded word is a hash of its address and carries a 4-bit tensor tag, every piece of
arithmetic is evaluated twice) and the library's own victims (the head's resize, the final layer's filter gradient; compared bit
for bit with their result alone) run on stream V while an aggressor runs on stream A:

    library aggressors   c    conv_x3_k + x3_fixup_k (the decoder's three 56x56 convs)
                         f    conv_filter_x3_batched_k + fold
                         cf   both
                         n    the same convs on the native fp32 instruction
    synthetic  s<mask>   1 bf16 MFMA | 2 f32->bf16 conversions + split arithmetic | 4 LDS b128 traffic + LDS-only barrier | 8 raw buffer
                         loads (scalar offset, out-of-range lanes) | 16 fp32 MFMA | 32 transposed LDS reads | 64 (with 4) __syncthreads()
                         instead of the LDS-only barrier | 128 (with 8) plain global loads; s15 = a register-light model of conv_x3_k

    python tools/interfere_probe.py [--agg c,f,cf,...] [--iters 60] [--mask K]   (MLIIS_HIP_LIB selects a probe build of the library)

--mask K: the victims' stream is restricted to the last K CU ids, the aggressor's to the first 256 - K (hipExtStreamCreateWithCUMask);
the CU census of the self-checking kernels (hardware ids of their waves) is printed so that the partition can be checked.
Prints, per aggressor: victim launches, launches whose output differs from the solo run, in-kernel fault records and their decoding."""
import argparse
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from mliis_amd import ops  # noqa: E402
from mliis_amd import _lib  # noqa: E402

SO = os.path.join(HERE, "libinterfere.so")
if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(os.path.join(HERE, "interfere.hip")):
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO, os.path.join(HERE, "interfere.hip")], check=True)
ifp = C.CDLL(SO)
ifp.ifp_tagged_bits.restype = C.c_uint
hip = C.CDLL("libamdhip64.so")
d = torch.device("cuda:0")
TAGS = {0x5: "victim map (small)", 0x6: "victim dec", 0x7: "victim dsmall", 0xA: "aggressor x", 0xB: "aggressor w", 0xC: "aggressor dy", 0xD: "synthetic source"}
LOG_BYTES = ifp.ifp_log_bytes()
REC0, RECW, MAXREC = 32, 32, 512


def P(t):
    return C.c_void_p(t.data_ptr())


def st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def tagged(shape, tag):
    t = torch.empty(shape, dtype=torch.float32, device=d)
    ifp.ifp_fill_tagged(P(t), C.c_uint(t.numel()), C.c_uint(tag), st())
    return t


def masked_stream(cus):
    words = (C.c_uint32 * 8)()
    for c in cus:
        words[c // 32] |= 1 << (c % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask failed: %d" % rc)
    return torch.cuda.ExternalStream(s.value)


class Log:
    def __init__(self):
        self.t = torch.zeros(LOG_BYTES // 4, dtype=torch.int32, device=d)

    def reset(self):
        self.t.zero_()

    def read(self):
        a = self.t.cpu().numpy().view(np.uint32)
        n = int(a[0])
        cus = sum(bin(int(w)).count("1") for w in a[1:17])
        per_xcc = [bin(int(a[1 + 2 * x])).count("1") + bin(int(a[2 + 2 * x])).count("1") for x in range(8)]
        recs = a[REC0:REC0 + RECW * min(n, MAXREC)].reshape(-1, RECW)
        return n, cus, per_xcc, recs, a[1:17].copy()


def f32(u):
    return float(np.array([u], dtype=np.uint32).view(np.float32)[0])


def origin(word, searchable):
    """where a wrong loaded word could have come from: its tag, and an exact-bits search of the probe's tensors"""
    tag = int(word) & 15
    hits = []
    for name, t in searchable.items():
        idx = (t.view(torch.int32).flatten() == int(np.array([word], dtype=np.uint32).view(np.int32)[0])).nonzero().flatten()
        if idx.numel():
            hits.append("%s[%d]%s" % (name, int(idx[0]), "" if idx.numel() == 1 else " (+%d more)" % (idx.numel() - 1)))
    return "tag 0x%x (%s) %s" % (tag, TAGS.get(tag, "no tensor of the probe"), ("found in " + ", ".join(hits)) if hits else "bits found in no tensor of the probe")


def hwstr(h, x):
    return "xcc %d se %d cu %2d simd %d wave %2d" % (x, (h >> 13) & 3, (h >> 8) & 15, (h >> 4) & 3, h & 15)


def decode(rec, searchable):
    kind = int(rec[0])
    head = "  block %5d thread %3d (lane %2d) %s elem %d: " % (rec[1], rec[2], rec[2] & 63, hwstr(int(rec[3]), int(rec[4])), rec[5])
    if kind == 1:
        f = int(rec[6])
        parts = []
        for k, nm in enumerate(("tl", "tr", "bl", "br")):
            for which, off, bit in (("load", 11, k), ("reload", 19, 4 + k)):
                if f >> bit & 1:
                    e = int(rec[7 + k])
                    exp = (ifp.ifp_tagged_bits(2 * e, 0x5), ifp.ifp_tagged_bits(2 * e + 1, 0x5))
                    got = (int(rec[off + 2 * k]), int(rec[off + 2 * k + 1]))
                    for c in range(2):
                        if got[c] != exp[c]:
                            parts.append("%s %s.%s elem %d: got %08x (%.6g) expected %08x (%.6g): %s" % (which, nm, "xy"[c], e, got[c], f32(got[c]), exp[c], f32(exp[c]), origin(got[c], searchable)))
        if f >> 8 & 1:
            parts.append("index arithmetic disagrees (e0 %d vs %d)" % (rec[7], rec[31]))
        if f >> 9 & 1:
            parts.append("outputs disagree: (%08x %08x) vs (%08x %08x)%s" % (rec[27], rec[28], rec[29], rec[30], "" if f & 0xff else " with IDENTICAL loads: arithmetic"))
        return head + "; ".join(parts)
    if kind == 2:
        names = ("wo", "ho", "n", "floor", "fma chain 0", "fma chain 1")
        return head + "; ".join("%s %08x vs %08x" % (names[k], rec[7 + k], rec[13 + k]) for k in range(6) if int(rec[6]) >> k & 1)
    if kind == 3:
        v = int(rec[7])
        return head + "%d-byte load: " % (4 * v) + "; ".join("word %d got %08x expected %08x: %s" % (k, rec[8 + k], rec[12 + k], origin(int(rec[8 + k]), searchable)) for k in range(v) if int(rec[6]) >> k & 1)
    return head + "kind %d" % kind


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--agg", default="none,cf,s15,s1,s2,s4,s8,s3,s5,s9,s6,s10,s12,s7,s11,s13,s14,s79,s143,s30,s28")
    ap.add_argument("--iters", type=int, default=60)
    ap.add_argument("--mask", type=int, default=0)
    ap.add_argument("--calls", type=int, default=6, help="victim rounds per aggressor burst")
    ap.add_argument("--matrix", action="store_true", help="also run every op_sel form of v_pk_mul / add / fma_f32 as a victim (48 launches per round)")
    ap.add_argument("--form-iters", type=int, default=48)
    ap.add_argument("--syn-iters", type=int, default=600)
    ap.add_argument("--show", type=int, default=6, help="fault records decoded per aggressor")
    a = ap.parse_args()
    print("library: %s   mask: %s" % (_lib.LIB_PATH, ("victims on the last %d CU ids" % a.mask) if a.mask else "none"), flush=True)
    N, H = 8, 56
    # ---- aggressor operands (tagged) and plans, as tools/x3_race_probe.py
    x = tagged((N, H, H, 224), 0xA)
    w = tagged((3, 3, 224, 112), 0xB) * 0.02
    w1 = tagged((3, 3, 136, 112), 0xB) * 0.02
    dy = tagged((N, H, H, 112), 0xC)
    imf, imb, imb1 = ops.x3_image_of(w, "fwd"), ops.x3_image_of(w, "bwd"), ops.x3_image_of(w1, "bwd")
    ws2 = ops.Workspace(d, 1 << 25)
    y_c = torch.empty(N, H, H, 112, device=d)
    dx1, dx2 = torch.zeros(N, H, H, 136, device=d), torch.empty(N, H, H, 224, device=d)
    nfl = ops.lib.size("mliis_conv2d_bwd_filter_workspace_floats", N, H, H, 224, 112, 3)
    pfl = torch.zeros(nfl, device=d)
    fb = ops.FilterBatch(d)
    fb.add(x, dy, 3, 1, pfl)
    wt, wt1 = ops.hwoi(w), ops.hwoi(w1)
    syn_src = tagged((1 << 22,), 0xD)
    syn_out = torch.empty(4096 * 256, device=d)
    # ---- victims
    small = tagged((N, H, H, 2), 0x5)
    dec, dsmall = tagged((N, H, H, 112), 0x6), tagged((N, H, H, 2), 0x7)
    searchable = {"x": x, "dy": dy, "small": small, "dec": dec, "dsmall": dsmall, "syn_src": syn_src}
    ws1 = ops.Workspace(d, 1 << 22)
    CALLS = a.calls
    out_r = [torch.empty(N, 224, 224, 2, device=d) for _ in range(CALLS)]
    out_a = [torch.empty(N, 224, 224, 2, device=d) for _ in range(CALLS)]
    out_c = [torch.empty(N * H * H * 2, device=d) for _ in range(3 * CALLS)]
    out_lr = [torch.empty(N, 224, 224, 2, device=d) for _ in range(CALLS)]
    out_dw = [torch.zeros(112 * 2, device=d) for _ in range(CALLS)]
    out_db = [torch.zeros(2, device=d) for _ in range(CALLS)]
    vlog, alog = Log(), Log()
    total = N * 224 * 224

    out_pr = [torch.empty(N, 224, 224, 2, device=d) for _ in range(CALLS)]
    out_pg = [torch.empty(N, 224, 224, 2, device=d) for _ in range(CALLS)]

    form_y = torch.empty(2 * 1568 * 256, device=d)
    form_bad = torch.zeros(16, dtype=torch.int32, device=d)
    FORMS = ["v_pk_fma_f32 (VGPR)", "v_pk_mul_f32 op_sel", "v_pk_mul_f32 (SGPR pair)", "v_pk_fma_f32 (SGPR pair, neg)", "v_pk_add_f32 (1.0, op_sel_hi, neg)",
             "v_pk_fma_f32 (0, op_sel_hi)", "v_pk_fma_f32 (operand from global_load_dwordx2)", "control: v_fma_f32 twice"]

    sel_bad = torch.zeros(288, dtype=torch.int32, device=d)

    def victims(j):
        if a.matrix:
            ifp.ifp_sel_matrix(P(form_y), P(sel_bad), 1568, a.form_iters, st())
        ifp.ifp_forms(P(form_y), P(form_bad), P(small), C.c_uint(small.numel() // 2), 1568, a.form_iters, st())
        ops.resize_bilinear_fwd(small, (224, 224), out=out_lr[j])
        ops.final_conv_bwd_filter(dec, dsmall, None, dw=out_dw[j].view(1, 1, 112, 2), db=out_db[j], ws=ws1)
        ifp.ifp_plain_resize(P(small), P(out_pr[j]), N, H, H, 224, 224, 1568, st())
        ifp.ifp_plain_gather(P(small), P(out_pg[j]), N, H, H, 224, 224, 1568, st())
        ifp.ifp_resize(P(small), P(out_r[j]), N, H, H, 224, 224, 0x5, P(vlog.t), st())
        ifp.ifp_alu(P(out_a[j]), C.c_longlong(total), 224, 224, P(vlog.t), st())
        for k, v in enumerate((1, 2, 4)):
            ifp.ifp_copy(P(small), P(out_c[3 * j + k]), C.c_uint(small.numel()), v, 0x5, P(vlog.t), st())

    def outputs(j):
        return [out_r[j], out_a[j], out_c[3 * j], out_c[3 * j + 1], out_c[3 * j + 2], out_lr[j], out_dw[j], out_pr[j], out_pg[j]]

    VNAMES = ["resize*", "alu*", "copy4*", "copy8*", "copy16*", "lib resize", "lib final dW", "plain resize", "plain gather"]
    victims(0)
    torch.cuda.synchronize()
    ref = [t.clone() for t in outputs(0)]
    n0 = vlog.read()[0]
    print("solo: in-kernel faults %d; self-checking resize == library resize: %s" % (n0, torch.equal(ref[0], ref[5])), flush=True)

    def aggressor(kind):
        if kind == "none":
            return
        if kind[0] == "s":
            if ifp.ifp_aggressor(int(kind[1:]), P(syn_out), P(syn_src), C.c_uint(syn_src.numel() * 4), 2048, a.syn_iters, P(alog.t), st()) != 0:
                raise RuntimeError("no synthetic aggressor instance for mask " + kind[1:])
            return
        for _ in range(2):
            if "c" in kind:
                ops.conv2d_fwd_x3(x, imf, 3, 112, None, 1, out=y_c, ws=ws2)
                ops.conv2d_bwd_data_x3(dy, imb1, 3, 136, 2, out=dx1, accumulate=True, ws=ws2)
                ops.conv2d_bwd_data_x3(dy, imb, 3, 224, 1, out=dx2, ws=ws2)
            if "f" in kind:
                fb.launch("fp32x3")
            if kind in ("n", "nb"):   # the same convs on the native instances: fp32 instruction | bf16 operands (16x16x32 bf16)
                pr = "fp32" if kind == "n" else "bf16"
                ops.conv2d_fwd(x, w, None, 1, out=y_c, ws=ws2, wt=wt, precision=pr)
                ops.conv2d_bwd_data(dy, w1, 2, out=dx1, accumulate=True, ws=ws2, precision=pr)
                ops.conv2d_bwd_data(dy, w, 1, out=dx2, ws=ws2, precision=pr)
                fb.launch(pr)

    if a.mask:
        sV, sA = masked_stream(range(256 - a.mask, 256)), masked_stream(range(256 - a.mask))
    else:
        sV, sA = torch.cuda.Stream(), torch.cuda.Stream()
    summary = []
    for kind in a.agg.split(","):
        vlog.reset(); alog.reset()
        torch.cuda.synchronize()
        # duration of one aggressor burst and of one victim round, alone
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(sA):
            aggressor(kind); e0.record(); aggressor(kind); e1.record()
        torch.cuda.synchronize()
        t_agg = e0.elapsed_time(e1) * 1e3
        with torch.cuda.stream(sV):
            victims(0); e0.record(); victims(0); e1.record()
        torch.cuda.synchronize()
        t_vic = e0.elapsed_time(e1) * 1e3
        vlog.reset(); alog.reset()
        form_bad.zero_(); sel_bad.zero_()
        wrong = [0] * len(VNAMES)
        first_wrong_call = [0] * CALLS
        examples = []
        for it in range(a.iters):
            with torch.cuda.stream(sA):
                aggressor(kind)
            with torch.cuda.stream(sV):
                for j in range(CALLS):
                    victims(j)
            torch.cuda.synchronize()
            for j in range(CALLS):
                for i, t in enumerate(outputs(j)):
                    if not torch.equal(t, ref[i]):
                        wrong[i] += 1
                        first_wrong_call[j] += 1
                        if len(examples) < a.show:
                            nz = (t.view(torch.int32).flatten() != ref[i].view(torch.int32).flatten()).nonzero().flatten()
                            runs = nz.tolist()
                            # pattern: channel parity of the wrong words, 16-lane groups (element pairs: thread = word // 2), and whether a
                            # wrong word is the CORRECT value of another position of the same tensor (a misdirected index / store)
                            par = [sum(1 for r_ in runs if r_ % 2 == c_) for c_ in (0, 1)]
                            groups = {}
                            for r_ in runs:
                                groups.setdefault((r_ // 2) // 16, []).append((r_ // 2) % 16)
                            gs = sorted(len(set(v_)) for v_ in groups.values())
                            tb, rb = t.view(torch.int32).flatten(), ref[i].view(torch.int32).flatten()
                            elsewhere = []
                            for r_ in runs[:3]:
                                hit = (rb == tb[r_]).nonzero().flatten()
                                elsewhere.append("word %d = ref[%s]" % (r_, hit[:2].tolist()) if hit.numel() else "word %d: in no position of ref" % r_)
                            examples.append("  iter %d call %d %s: %d words differ, first %s last %d; by channel parity %s; %d groups of 16 threads, distinct lanes per group min %d median %d max %d; %s; got %s ref %s" % (
                                it, j, VNAMES[i], nz.numel(), runs[:6], runs[-1], par, len(gs), gs[0], gs[len(gs) // 2], gs[-1], "; ".join(elsewhere),
                                t.flatten()[nz[:3]].tolist(), ref[i].flatten()[nz[:3]].tolist()))
        nf, vcus, vx, recs, vbm = vlog.read()
        na, acus, ax, _, abm = alog.read()
        launches = a.iters * CALLS
        print("\n== aggressor %-4s burst %7.1f us, victim round %6.1f us; %d victim rounds" % (kind, t_agg, t_vic, launches))
        print("   launches with wrong output: " + ", ".join("%s %d" % (VNAMES[i], wrong[i]) for i in range(len(VNAMES))))
        print("   by call within a burst: %s;   in-kernel fault records: %d" % (first_wrong_call, nf))
        print("   CU census: victims on %d CU slots %s%s" % (vcus, vx, (", synthetic aggressor on %d %s, shared slots %d" % (
            acus, ax, sum(bin(int(p & q)).count("1") for p, q in zip(vbm, abm)))) if kind[0] == "s" else ""))
        fbad = form_bad.cpu().tolist()
        checks = launches * 1568 * 256 * a.form_iters
        print("   instruction forms (mismatching low / high halves of %.3g checks each): " % checks + "; ".join("%s %d / %d" % (FORMS[k], fbad[2 * k], fbad[2 * k + 1]) for k in range(8)))
        if a.matrix:
            sb = sel_bad.cpu().tolist()
            print("   packed fp32 source-select matrix (op_sel:[a,b] op_sel_hi:[c,d]; mismatching low, high halves of %.3g checks per form):" % checks)
            for o_, nm in enumerate(("v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32")):
                hit = ["[%d,%d][%d,%d]: %d, %d (wrong low results that equal the operation on src1's other register %d, on src0's other %d, on both others %d, none %d)" % (
                    k & 1, (k >> 1) & 1, (k >> 2) & 1, (k >> 3) & 1, sb[(o_ * 16 + k) * 2], sb[(o_ * 16 + k) * 2 + 1], *sb[96 + (o_ * 16 + k) * 4:100 + (o_ * 16 + k) * 4])
                       for k in range(16) if sb[(o_ * 16 + k) * 2] or sb[(o_ * 16 + k) * 2 + 1]]
                print("     %s: %s" % (nm, "; ".join(hit) if hit else "no form with a mismatch"))
        for e in examples:
            print(e)
        kinds = {}
        for r in recs:
            kinds[int(r[0])] = kinds.get(int(r[0]), 0) + 1
        if nf:
            print("   records by victim (1 resize*, 2 alu*, 3 copy*): %s" % kinds)
            lanes = {}
            for r in recs:
                key = (int(r[1]), int(r[2]) >> 4)
                lanes.setdefault(key, []).append(int(r[2]) & 15)
            sizes = sorted(len(v) for v in lanes.values())
            print("   faulting lanes per (block, 16-lane group): %d groups, sizes min %d median %d max %d" % (len(sizes), sizes[0], sizes[len(sizes) // 2], sizes[-1]))
        for r in recs[:a.show]:
            print(decode(r, searchable))
        sys.stdout.flush()
        summary.append((kind, launches, wrong, nf, fbad))
    print("\nsummary (%s)" % ("mask %d" % a.mask if a.mask else "no mask"))
    print("%-5s %8s  %s  in-kernel" % ("agg", "rounds", "  ".join("%12s" % v for v in VNAMES)))
    for kind, launches, wrong, nf, fbad in summary:
        print("%-5s %8d  %s  %d" % (kind, launches, "  ".join("%12d" % v for v in wrong), nf))
    print("\ninstruction-form victims, mismatching (low, high) halves:")
    print("%-5s  %s" % ("agg", "  ".join("%22s" % f[:22] for f in FORMS)))
    for kind, launches, wrong, nf, fbad in summary:
        print("%-5s  %s" % (kind, "  ".join("%22s" % ("%d, %d" % (fbad[2 * k], fbad[2 * k + 1])) for k in range(8))))


if __name__ == "__main__":
    main()
