"""Timeline of the LAST inner step of a rocprofv3 --kernel-trace run of bench.py: one row per kernel (start offset, duration, queue,
grid), gaps and overlaps between consecutive kernels, and -- for a step with a side branch -- how long the side queue's kernels ran
beside the main queue's.

    python tools/timeline.py <dir with *_kernel_trace.csv> [out.txt]
"""
import csv
import glob
import os
import sys


def short(n):
    n = n.replace("void mliis::", "").replace("mliis::", "")
    return n.split("(")[0][:58]


def main():
    d = sys.argv[1]
    rows = list(csv.DictReader(open(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    opt = [i for i, r in enumerate(rows) if "sgd_k" in r["Kernel_Name"] or "adam_b1zero_k" in r["Kernel_Name"]]
    if len(opt) < 2:
        raise SystemExit("need at least two optimizer launches in the trace")
    lo, hi = opt[-2] + 1, opt[-1] + 1
    step = rows[lo:hi]
    t0 = int(step[0]["Start_Timestamp"])
    out = []
    queues = {}
    for r in step:
        queues.setdefault(r["Queue_Id"], 0)
        queues[r["Queue_Id"]] += 1
    main_q = max(queues, key=queues.get)
    out.append("# last inner step: %d kernels, %.1f us wall, queues %s (main = %s)" % (
        len(step), (max(int(r["End_Timestamp"]) for r in step) - t0) / 1e3, dict(queues), main_q))
    prev_end = t0
    busy_main = busy_side = both = 0.0
    side_iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in step if r["Queue_Id"] != main_q]
    for r in step:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        wg = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
        is_main = r["Queue_Id"] == main_q
        ov = sum(max(0, min(e, b) - max(s, a)) for a, b in side_iv) / 1e3 if is_main else 0.0
        out.append("%9.1f %8.1f %s q%-3s wg %6d  gap %6.1f  %s%s" % ((s - t0) / 1e3, (e - s) / 1e3, "M" if is_main else "S", r["Queue_Id"], wg,
                                                                  (s - prev_end) / 1e3 if is_main else 0.0, short(r["Kernel_Name"]),
                                                                  ("   [%.1f us beside the side branch]" % ov) if ov > 0 else ""))
        if is_main:
            prev_end = e
            busy_main += (e - s) / 1e3
            both += ov
        else:
            busy_side += (e - s) / 1e3
    out.append("# main-queue kernel time %.1f us, side-queue kernel time %.1f us, main kernels' time beside side kernels %.1f us" % (busy_main, busy_side, both))
    text = "\n".join(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
