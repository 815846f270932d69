#!/bin/bash
# Variants of conv_x3_k (tools/build_variants.sh "name:-Dflag"): parity (tests/test_x3_gpu.py), launch time of the decoder's convs
# (tools/x3_probe.py, host-timed with the fix-up launch), fabric read traffic of the 224 -> 112 conv (rocprofv3 --pmc FETCH_SIZE, its own
# pass) and the step.   VARIANTS="base xcd k64" bash tools/run_x3_variants.sh     -> gpurun_out/x3var/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/x3var
mkdir -p $O
cd $R
for V in ${VARIANTS:-base xcd k64 xcdk64}; do
  if [ $V = base ]; then unset MLIIS_HIP_LIB; else export MLIIS_HIP_LIB=$R/tools/_alt/libmliis_$V.so; fi
  echo "== $V"
  python -m pytest tests/test_x3_gpu.py -q -x 2>&1 | tail -1
  python tools/x3_probe.py 30 rsd2.fuse 2>/dev/null | grep -v amdgpu
  python tools/x3_probe.py 30 rsd2.br1 2>/dev/null | grep -v amdgpu
  rm -rf $O/pmc_$V
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_$V -o f -- python3 tools/x3_probe.py 6 rsd2.fuse > $O/pmc_$V.log 2>&1
  python - $O/pmc_$V <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            k = r["Kernel_Name"].replace("void ", "").replace("mliis::", "").split("(")[0]
            if "x3" in k:
                a = acc[k]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (v, n) in sorted(acc.items()):
    print("   %-28s fabric read %7.1f MB per launch (2 x FETCH_SIZE KiB, gfx950 correction) over %d launches" % (k, 2 * v * 1024 / n / 1e6, n))
PY
  rm -rf $O/pmc_$V
  for rep in 1 2; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-native-retime 2>/dev/null | python -c "import sys,json; print('   step', json.loads(sys.stdin.read())['value'])"; done
done
