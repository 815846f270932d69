"""Prints, per kernel of a .hip file, the order of global loads (L), stores (S), waits (Wn = s_waitcnt vmcnt(n)), barriers (|) and
branches (b) in the generated gfx950 ISA -- a quick way to spot dependent memory round trips in latency-bound kernels.
    python tools/isa_waits.py mliis_amd/csrc/se.hip [name filter]"""
import os
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/_isa_waits.s"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-I" + os.path.join(root, "mliis_amd/csrc"),
                "-I" + os.path.join(root, "include"), src, "-o", out], check=True, stderr=subprocess.DEVNULL)
L = open(out).read().split("\n")
i = 0
while i < len(L):
    m = re.match(r"^(_Z\w+):\s+; @", L[i])
    if not m:
        i += 1
        continue
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
    seq = []
    i += 1
    while i < len(L) and "s_endpgm" not in L[i]:
        t = L[i].strip()
        if t.startswith(("global_load", "buffer_load")):
            seq.append("L")
        elif t.startswith(("global_store", "buffer_store")):
            seq.append("S")
        elif t.startswith("s_waitcnt vmcnt"):
            seq.append("[W" + re.search(r"vmcnt\((\d+)\)", t).group(1) + "]")
        elif t.startswith("s_barrier"):
            seq.append("|")
        elif t.startswith(("s_cbranch", "s_branch")):
            seq.append("b")
        i += 1
    if flt in name:
        print(name.replace("mliis::", "")[:60])
        print("   " + "".join(seq)[:400])
