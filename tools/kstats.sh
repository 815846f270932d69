#!/bin/bash
# Per-kernel statistics of the headline bench under rocprofv3 --kernel-trace --stats: the rows whose name matches $1 (a grep pattern).
#   bash tools/kstats.sh 'conv_filter_x3|conv_x3|x3_fixup' [bench.py arguments]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/kstats_$$
PAT=$1; shift
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-native-retime "$@" > /dev/null 2>&1
F=$(find $O -name "*kernel_stats.csv" | head -1)
python - "$F" "$PAT" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows:
    if re.search(sys.argv[2], r["Name"]):
        print("%-70s calls %5s  avg %8.1f us  total %9.1f us" % (r["Name"].replace("void ", "").replace("mliis::", "")[:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
print("all kernels: %.1f us" % (tot / 1e3))
PY
rm -rf $O
