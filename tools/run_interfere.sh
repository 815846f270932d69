#!/bin/bash
# gpurun calls of the interference study (tools/interfere_probe.py).  Output: gpurun_out/interfere/*.txt
#   STAGE=1: shipped library, the two probe builds (conv_x3_k unclaimed; both kernels claimed), CU-partitioned runs
#   STAGE=2: which instruction classes of a synthetic aggressor it takes (bit-mask subsets), which victims are hit
mkdir -p gpurun_out/interfere
O=gpurun_out/interfere
IT=${IT:-60}
case "${STAGE:-1}" in
1)
  timeout 600 python tools/interfere_probe.py --iters $IT --agg none,c,f,cf,n,s1,s16,s2,s3,s4,s8,s32,s15 > $O/shipped.txt 2>&1
  MLIIS_HIP_LIB=$PWD/tools/_alt/libmliis_noclaim.so timeout 400 python tools/interfere_probe.py --iters $IT --agg none,c,f,cf > $O/noclaim.txt 2>&1
  MLIIS_HIP_LIB=$PWD/tools/_alt/libmliis_bothclaim.so timeout 400 python tools/interfere_probe.py --iters $IT --agg none,c,f,cf > $O/bothclaim.txt 2>&1
  MLIIS_HIP_LIB=$PWD/tools/_alt/libmliis_noclaim.so timeout 400 python tools/interfere_probe.py --iters $IT --mask 64 --agg none,c,cf,s3,s15 > $O/noclaim_mask64.txt 2>&1
  timeout 400 python tools/interfere_probe.py --iters $IT --mask 64 --agg none,c,cf > $O/shipped_mask64.txt 2>&1
  timeout 300 python tools/x3_race_probe.py cf > $O/race_cf_shipped.txt 2>&1
  tail -n 12 $O/shipped.txt $O/noclaim.txt $O/bothclaim.txt $O/noclaim_mask64.txt $O/shipped_mask64.txt; tail -n 5 $O/race_cf_shipped.txt ;;
2)
  timeout 900 python tools/interfere_probe.py --iters $IT > $O/bisect.txt 2>&1
  tail -n 30 $O/bisect.txt ;;
3)
  timeout 900 python tools/interfere_probe.py --iters $IT --agg none,s5,s9,s15,s1,s20,s28,n,nb,cf > $O/forms.txt 2>&1
  MLIIS_HIP_LIB=$PWD/tools/_alt/libmliis_noclaim.so timeout 400 python tools/interfere_probe.py --iters $IT --agg c > $O/forms_noclaim.txt 2>&1
  tail -n 28 $O/forms.txt; tail -n 8 $O/forms_noclaim.txt ;;
4)
  timeout 900 python tools/interfere_probe.py --iters $IT --matrix --show 2 --agg none,s15,s5,s9,nb,s28 > $O/matrix.txt 2>&1
  MLIIS_HIP_LIB=$PWD/tools/_alt/libmliis_noclaim.so timeout 400 python tools/interfere_probe.py --iters $IT --matrix --show 2 --agg c > $O/matrix_noclaim.txt 2>&1
  MLIIS_HIP_LIB=$PWD/tools/_alt/libmliis_bothclaim.so timeout 900 python tools/interfere_probe.py --iters 600 --show 2 --agg cf,c,f > $O/bothclaim_3600.txt 2>&1
  grep -A4 "source-select matrix" $O/matrix.txt $O/matrix_noclaim.txt; tail -n 12 $O/bothclaim_3600.txt ;;
5)
  timeout 600 python tools/interfere_probe.py --iters 30 --matrix --show 1 --agg s15,nb > $O/matrix2.txt 2>&1
  grep -A4 "source-select matrix" $O/matrix2.txt | cut -c1-1500
  B="--steps 12 --warmup 3 --no-cpu-baseline --no-roofline --no-native-retime"
  for rep in 1 2 3; do
    python bench.py $B 2>/dev/null | python -c "import sys,json; print('shipped', json.loads(sys.stdin.read())['value'])"
    MLIIS_HIP_LIB=$PWD/tools/_alt/libmliis_noslp.so python bench.py $B 2>/dev/null | python -c "import sys,json; print('noslp  ', json.loads(sys.stdin.read())['value'])"
  done
  python bench.py --tasks-per-gpu 8 --concurrent-tasks 4 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-native-retime 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('lanes4 shipped', j['value'], j['dtype'][:60])"
  MLIIS_HIP_LIB=$PWD/tools/_alt/libmliis_noslp.so python bench.py --tasks-per-gpu 8 --concurrent-tasks 4 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-native-retime 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('lanes4 noslp  ', j['value'])"
  python -m pytest tests/test_interference_gpu.py tests/test_x3_gpu.py -q -rxs 2>&1 | tail -15
  for rep in 1 2 3; do python -m pytest tests/test_step_gpu.py -q -k "concurrent_task_lanes_equal_the_sequential_meta_step" 2>&1 | tail -2; done
  python bench.py > $O/bench_full.json 2> $O/bench_full.err; tail -c 600 $O/bench_full.json; tail -5 $O/bench_full.err ;;
6)
  timeout 600 python tools/interfere_probe.py --iters 30 --matrix --show 1 --agg s15,nb > $O/matrix3.txt 2>&1
  grep -A4 "source-select matrix" $O/matrix3.txt | cut -c1-2500
  python -m pytest tests -m gpu -q -x -rxs 2>&1 | tail -15
  B="--steps 12 --warmup 3 --no-cpu-baseline --no-roofline --no-native-retime"
  for rep in 1 2; do python bench.py $B 2>/dev/null | python -c "import sys,json; print('default', json.loads(sys.stdin.read())['value'])"; done
  python bench.py --tasks-per-gpu 8 --concurrent-tasks 4 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-native-retime 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('lanes4', j['value'])"
  python bench.py --precision bf16 --tasks-per-gpu 8 --concurrent-tasks 4 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('lanes4 bf16', j['value'])" ;;
esac
