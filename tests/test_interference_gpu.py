"""Kernels of one stream beside the library's bf16-matrix-instruction kernels on another: the fault round 5 saw and round 6 named.

Mechanism (profiles/r06_notes.md; tools/interfere_probe.py is the long form of these tests): on MI355X, while a wave that interleaves
v_mfma_f32_16x16x32_bf16 with LDS or vector-memory instructions is resident on a CU, a packed fp32 instruction (v_pk_mul_f32 /
v_pk_add_f32 / v_pk_fma_f32) of any OTHER wave on that CU whose op_sel is [0,1] intermittently returns a wrong LOW result (not the
result of the operation on any other choice among the instruction's own source registers).  Nothing else is affected, and nothing on a CU the aggressor does not occupy.  The library is fenced on both sides:
  * conv_x3_k and conv_filter_x3_batched_k (the split-product kernels of the default fp32 path) occupy their CUs alone (whole register
    file): no foreign wave can be co-resident -> safe beside ANY kernel of another stream;
  * the library itself contains no packed fp32 instruction with that select (tests/test_build_cpu.py), so its own kernels on
    concurrent streams (task lanes) are not victims of its bf16 / fp8 instances either.
Victims here: the two library kernels that showed the fault in round 5 (head resize, final layer's filter gradient; bit-identical to
their result alone), and register-only kernels of tools/interfere.hip that execute the affected instruction form and check every
result against single instructions in the same thread."""
import ctypes as C
import os
import subprocess

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "libinterfere.so")
SRC = os.path.join(ROOT, "tools", "interfere.hip")


def _ifp():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(SRC):
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO, SRC], check=True)
    return C.CDLL(SO)


def _P(t):
    return C.c_void_p(t.data_ptr())


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _Rig:
    """Aggressors and victims of tools/interfere_probe.py at the decoder's real sizes (N = 8, 56 x 56, 224 / 136 -> 112 channels)."""

    def __init__(self):
        from mliis_amd import ops
        self.ops, self.ifp = ops, _ifp()
        d = self.d = torch.device("cuda:0")
        g = torch.Generator(device="cpu").manual_seed(3)
        N, H = 8, 56
        self.N, self.H = N, H
        r = lambda *s: torch.randn(*s, generator=g).to(d)   # noqa: E731
        self.x, self.dy = r(N, H, H, 224), r(N, H, H, 112)
        self.w, self.w1 = r(3, 3, 224, 112) * 0.02, r(3, 3, 136, 112) * 0.02
        self.imf, self.imb, self.imb1 = ops.x3_image_of(self.w, "fwd"), ops.x3_image_of(self.w, "bwd"), ops.x3_image_of(self.w1, "bwd")
        self.wt = ops.hwoi(self.w)
        self.ws2 = ops.Workspace(d, 1 << 25)
        self.y = torch.empty(N, H, H, 112, device=d)
        self.dx1, self.dx2 = torch.zeros(N, H, H, 136, device=d), torch.empty(N, H, H, 224, device=d)
        nfl = ops.lib.size("mliis_conv2d_bwd_filter_workspace_floats", N, H, H, 224, 112, 3)
        self.fb = ops.FilterBatch(d)
        self.fb.add(self.x, self.dy, 3, 1, torch.zeros(nfl, device=d))
        self.syn_src = r(1 << 20)
        self.syn_out = torch.empty(4096 * 256, device=d)
        self.syn_log = torch.zeros(self.ifp.ifp_log_bytes() // 4, dtype=torch.int32, device=d)
        # victims
        self.small, self.dec, self.dsmall = r(N, H, H, 2), r(N, H, H, 112), r(N, H, H, 2)
        self.ws1 = ops.Workspace(d, 1 << 22)
        self.out_r, self.out_dw, self.out_db = torch.empty(N, 224, 224, 2, device=d), torch.zeros(224, device=d), torch.zeros(2, device=d)
        self.form_y = torch.empty(2 * 1568 * 256, device=d)
        self.form_bad = torch.zeros(16, dtype=torch.int32, device=d)
        self.sel_bad = torch.zeros(288, dtype=torch.int32, device=d)
        self.sV, self.sA = torch.cuda.Stream(), torch.cuda.Stream()
        self.lib_victims()
        torch.cuda.synchronize()
        self.ref = (self.out_r.clone(), self.out_dw.clone())

    def lib_victims(self):
        self.ops.resize_bilinear_fwd(self.small, (224, 224), out=self.out_r)
        self.ops.final_conv_bwd_filter(self.dec, self.dsmall, None, dw=self.out_dw.view(1, 1, 112, 2), db=self.out_db, ws=self.ws1)

    def aggressor(self, kind):
        ops = self.ops
        if kind.startswith("s"):
            assert self.ifp.ifp_aggressor(int(kind[1:]), _P(self.syn_out), _P(self.syn_src), C.c_uint(self.syn_src.numel() * 4), 2048, 600, _P(self.syn_log), _st()) == 0
            return
        for _ in range(2):
            if "c" in kind:
                ops.conv2d_fwd_x3(self.x, self.imf, 3, 112, None, 1, out=self.y, ws=self.ws2)
                ops.conv2d_bwd_data_x3(self.dy, self.imb1, 3, 136, 2, out=self.dx1, accumulate=True, ws=self.ws2)
                ops.conv2d_bwd_data_x3(self.dy, self.imb, 3, 224, 1, out=self.dx2, ws=self.ws2)
            if "f" in kind:
                self.fb.launch("fp32x3")
            if kind == "nb":     # the native instances with bf16 operands (v_mfma_f32_16x16x32_bf16 out of LDS): `--precision bf16`
                ops.conv2d_fwd(self.x, self.w, None, 1, out=self.y, ws=self.ws2, wt=self.wt, precision="bf16")
                ops.conv2d_bwd_data(self.dy, self.w1, 2, out=self.dx1, accumulate=True, ws=self.ws2, precision="bf16")
                ops.conv2d_bwd_data(self.dy, self.w, 1, out=self.dx2, ws=self.ws2, precision="bf16")
                self.fb.launch("bf16")

    def run(self, kind, iters, rounds=6, matrix=False, forms=True):
        """-> (victim rounds, library-victim rounds that differ from the solo result, form counters, select-matrix counters)"""
        self.form_bad.zero_()
        self.sel_bad.zero_()
        wrong = 0
        for _ in range(iters):
            with torch.cuda.stream(self.sA):
                self.aggressor(kind)
            outs = []
            with torch.cuda.stream(self.sV):
                for _ in range(rounds):
                    if matrix:
                        self.ifp.ifp_sel_matrix(_P(self.form_y), _P(self.sel_bad), 1568, 48, _st())
                    if forms:
                        self.ifp.ifp_forms(_P(self.form_y), _P(self.form_bad), _P(self.small), C.c_uint(self.small.numel() // 2), 1568, 48, _st())
                    self.lib_victims()
                    outs.append((self.out_r.clone(), self.out_dw.clone()))
            torch.cuda.synchronize()
            wrong += sum(1 for a, b in outs if not (torch.equal(a, self.ref[0]) and torch.equal(b, self.ref[1])))
        return iters * rounds, wrong, self.form_bad.cpu().tolist(), self.sel_bad.cpu().tolist()


@pytest.fixture(scope="module")
def rig():
    return _Rig()


@pytest.mark.parametrize("kind", ["c", "f", "cf"])
def test_split_product_kernels_leave_kernels_of_another_stream_alone(rig, kind):
    """The shipped split-product kernels (convs, their filter gradients, both) on one stream; on another, the library kernels that
    showed the fault in round 5 and register-only kernels that execute the affected instruction form: 1200 victim rounds per
    aggressor, every library result bit-identical to its result alone, every packed result equal to its single-instruction value.
    (Before conv_x3_k claimed its CUs: 240 of 360 rounds wrong; before the filter-gradient kernel did: `cf` 40-50 of 360.)"""
    rounds, wrong, forms, _ = rig.run(kind, iters=200)
    assert rounds >= 1200
    assert wrong == 0, "%d of %d victim rounds differ from the solo result beside aggressor %r" % (wrong, rounds, kind)
    assert sum(forms) == 0, "packed fp32 forms disagree with single instructions beside aggressor %r: %s" % (kind, forms)


def test_bf16_instances_do_not_disturb_the_librarys_own_kernels(rig):
    """The native bf16-operand instances (`--precision bf16` / bf16-storage / fp8: v_mfma_f32_16x16x32_* out of LDS, several workgroups
    per CU, co-resident with anything) ARE aggressors for the affected form -- the library's own kernels on another stream stay exact
    because the library no longer contains that form (common.hpp: lone(); tests/test_build_cpu.py disassembles the built library)."""
    rounds, wrong, _, _ = rig.run("nb", iters=200, forms=False)
    assert rounds >= 1200
    assert wrong == 0, "%d of %d library-victim rounds differ from the solo result beside the bf16 instances" % (wrong, rounds)


def test_packed_fp32_select_fault_characterisation(rig):
    """What the hardware does, recorded in the test log: a register-light synthetic aggressor (bf16 matrix instruction + conversions + LDS
    traffic + buffer loads, tools/interfere.hip mask 15) beside every source-select form of the three packed fp32 instructions.  On the
    MI355X boxes of rounds 5-6 only op_sel:[0,1] forms fail, only in the LOW result.  The test asserts that SHAPE when the fault shows
    and reports it as an expected failure; on a part / firmware where nothing fails it passes."""
    rounds, _, _, sel = rig.run("s15", iters=40, matrix=True, forms=False)
    names = ("v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32")
    lines, total, outside = [], 0, 0
    for o in range(3):
        for k in range(16):
            lo, hi, sw = sel[(o * 16 + k) * 2], sel[(o * 16 + k) * 2 + 1], sel[96 + (o * 16 + k) * 4:100 + (o * 16 + k) * 4]
            if lo or hi:
                total += lo + hi
                a, b = k & 1, (k >> 1) & 1
                lines.append("%s op_sel:[%d,%d] op_sel_hi:[%d,%d]: low %d (equal to the operation on src1's other register / src0's other / both others / none: %s), high %d" % (names[o], a, b, (k >> 2) & 1, (k >> 3) & 1, lo, sw, hi))
                if (a, b) != (0, 1) or hi:
                    outside += lo + hi
    checks = rounds * 1568 * 256 * 48
    print("\npacked fp32 select matrix beside synthetic aggressor 15, %d victim rounds, %.3g checks per form:" % (rounds, checks))
    print("\n".join(lines) if lines else "no form with a mismatch")
    if total == 0:
        return
    assert outside == 0, "mismatches outside the op_sel:[0,1] low results:\n" + "\n".join(lines)
    pytest.xfail("hardware behaviour reproduced on this box: %d wrong low results, all in op_sel:[0,1] forms (see the captured output)" % total)
