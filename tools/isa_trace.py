"""Condensed instruction trace of one kernel of a .hip file (run-length encoded opcode classes per basic block): shows at a glance
where the compiler put the LDS reads, FMAs, global loads / stores, spills and barriers.
usage: isa_trace.py file.hip 'mangled-name-substring' [--loop]"""
import collections
import re
import subprocess
import sys

src, pat = sys.argv[1], sys.argv[2]
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-pass-failed", "-S",
                "--cuda-device-only", "-o", "/tmp/isa_trace.s", src], check=True)
s = open("/tmp/isa_trace.s").read()
names = [m.group(1) for m in re.finditer(r"^(_Z\w+):", s, re.M) if pat in m.group(1)]
assert names, "no kernel matches"
name = names[0]
i = s.index(name + ":")
body = s[i:s.index("s_endpgm", i)].split("\n")


def cls(op):
    for pre, c in (("v_pk_fma", "FMA"), ("v_fma", "fma"), ("v_pk_mul", "MUL"), ("v_pk_add", "ADD"), ("ds_read", "LDSR"), ("ds_write", "LDSW"), ("buffer_load", "GLD"), ("flat_load", "FLAT"),
                   ("global_load", "GLD"), ("global_store", "GST"), ("buffer_store", "GST"), ("scratch_load", "SPL"), ("scratch_store", "SPS"),
                   ("s_barrier", "BAR"), ("s_waitcnt", "W"), ("s_cbranch", "BR"), ("v_exp", "EXP"), ("v_rcp", "RCP"), ("ds_bpermute", "PERM"),
                   ("v_accvgpr", "ACC")):
        if op.startswith(pre):
            return c
    return None


out, last, n = [], None, 0
cnt = collections.Counter()
for l in body:
    if re.match(r"^\.LBB", l):
        if last:
            out.append("%sx%d" % (last, n))
        out.append("\n" + l.split(":")[0] + ":")
        last, n = None, 0
        continue
    m = re.match(r"\s+([a-z_0-9]+)", l)
    if not m:
        continue
    cnt[m.group(1)] += 1
    c = cls(m.group(1))
    if c is None:
        continue
    if c == last:
        n += 1
    else:
        if last:
            out.append("%sx%d" % (last, n))
        last, n = c, 1
if last:
    out.append("%sx%d" % (last, n))
print(name, len(body), "lines")
print(" ".join(out))
print(cnt.most_common(25))
