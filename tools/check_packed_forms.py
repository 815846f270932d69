"""Which kernels of a HIP shared library contain the packed fp32 form that goes wrong beside bf16 matrix instructions on MI355X.

    python tools/check_packed_forms.py [mliis_amd/libmliis_hip.so]

The form (profiles/r06_notes.md, tools/interfere_probe.py --matrix): v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 with op_sel:[0,1,...] --
the LOW result takes src0 from the low and src1 from the HIGH register of its pair.  While a wave that interleaves
v_mfma_f32_16x16x32_bf16 (or _fp8) with LDS / vector-memory instructions is resident on the same CU, that low result is intermittently
wrong (for one 16-lane pass).  The shipped library must not contain the form (its bf16 / fp8 instances and the
split-product kernels would otherwise disturb the library's own kernels on concurrent streams): tests/test_build_cpu.py asserts it.
Prints kernel: count, exit code 1 if any."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
FORM = re.compile(r"\bv_pk_(mul|add|fma)_f32\b.*\bop_sel:\[0,1")


def affected_kernels(lib_path):
    """{kernel symbol: number of affected instructions} over every gfx950 code object bundled in the library."""
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib_path, local)
        subprocess.run([OBJDUMP, "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)   # extracts beside the input
        objs = [os.path.join(tmp, f) for f in sorted(os.listdir(tmp)) if "amdgcn" in f]
        if not objs:
            raise RuntimeError("no device code object found in " + lib_path)
        for o in objs:
            dis = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", o], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode(errors="replace")
            kernel = "?"
            for ln in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.*)>:", ln)
                if m:
                    kernel = m.group(1)
                elif FORM.search(ln):
                    out[kernel] = out.get(kernel, 0) + 1
    return out


def main():
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "mliis_amd", "libmliis_hip.so")
    hits = affected_kernels(lib)
    names = hits
    try:
        filt = subprocess.run(["c++filt"], input="\n".join(hits).encode(), stdout=subprocess.PIPE, check=True).stdout.decode().splitlines()
        names = dict(zip(filt, hits.values()))
    except Exception:   # noqa: BLE001  (demangling is cosmetic)
        pass
    for k, v in sorted(names.items(), key=lambda kv: -kv[1]):
        print("%4d  %s" % (v, k))
    print("%d kernels with v_pk_{mul,add,fma}_f32 op_sel:[0,1] in %s" % (len(hits), lib))
    return 1 if hits else 0


if __name__ == "__main__":
    sys.exit(main())
