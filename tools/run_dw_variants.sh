#!/bin/bash
# Variants of the row-marching depthwise kernels (SRC=dwmarch tools/build_variants.sh "name:-Dflag"): parity (tests/test_dwmarch_gpu.py) and the
# cold layer times at N = 8 and N = 64 of the blocks named in BLOCKS.    VARIANTS="base ts2" BLOCKS=3,4 bash tools/run_dw_variants.sh
cd $GRAFT_REPO_ROOT
for V in ${VARIANTS:-base ts2}; do
  if [ $V = base ]; then unset MLIIS_HIP_LIB; else export MLIIS_HIP_LIB=$PWD/tools/_alt/libmliis_$V.so; fi
  echo "== $V"
  python -m pytest tests/test_dwmarch_gpu.py -q -x 2>&1 | tail -1
  for T in ${TARGETS:-0}; do
    for N in 8 64; do
      echo "-- N = $N  MLIIS_DWM_TARGET=$T"
      MLIIS_DWM_TARGET=$T python tools/bench_dwmarch.py --n $N --blocks ${BLOCKS:-3,4} --reps 30 2>/dev/null | grep -v -E "amdgpu|^total"
    done
  done
done
