#!/bin/bash
# Same-box A/B of library variants (tools/build_variants.sh "name:-Dflag") on the headline step: the variants alternate REPS times
# (box-to-box spread is +-1 %, so only runs of one call compare).  Optional TESTS="tests/test_ops_gpu.py -k resize" first, per variant.
#   VARIANTS="base norows" REPS=3 bash tools/run_ab.sh        -> gpurun_out/ab/ab.txt
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ab
mkdir -p $O
cd $R
: > $O/ab.txt
pick() { if [ $1 = base ]; then unset MLIIS_HIP_LIB; else export MLIIS_HIP_LIB=$R/tools/_alt/libmliis_$1.so; fi; }
if [ -n "$TESTS" ]; then
  for V in ${VARIANTS:-base}; do
    pick $V
    echo "== $V: pytest $TESTS" | tee -a $O/ab.txt
    python -m pytest $TESTS -q -x 2>&1 | tail -3 | tee -a $O/ab.txt
  done
fi
for rep in $(seq 1 ${REPS:-3}); do
  for V in ${VARIANTS:-base}; do
    pick $V
    python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-roofline --no-native-retime ${BENCH_ARGS} 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-12s rep $rep  %8.1f images/s  %7.3f ms/task' % ('$V', d['value'], d['ms_per_step']))" | tee -a $O/ab.txt
  done
done
