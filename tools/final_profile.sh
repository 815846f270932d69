#!/bin/bash
# Round-end evidence on the GPU box (run through gpurun): kernel trace + statistics of the headline bench, PMC traffic passes (the
# dominant kernels of a step; the depthwise layers, cold), the bench line, the depthwise tables at N = 8 / 64, the variants.
# Everything lands under gpurun_out/$1 (default r04_final); tools/collect_profiles.sh copies the summaries into profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r06_final}
O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-native-retime > $O/trace.log 2>&1
python tools/profile_summary.py $O/trace 0 $O/profile.md > /dev/null
python tools/timeline.py $O/trace $O/final_timeline.txt > /dev/null
find $O/trace -name "*kernel_trace.csv" -delete     # (tens of MB of per-dispatch rows: the statistics and the last step's timeline are kept)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-native-retime --no-graph > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-native-retime --no-graph > $O/pmc_write.log 2>&1
python tools/pmc_traffic.py $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/pmc_write -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json 24
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/dw_fetch -o f -- python3 tools/bench_dwmarch.py --n 8 --reps 4 > $O/dw_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/dw_write -o w -- python3 tools/bench_dwmarch.py --n 8 --reps 4 > $O/dw_write.log 2>&1
python tools/pmc_depthwise.py $(find $O/dw_fetch -name "*counter_collection.csv" | head -1) $(find $O/dw_write -name "*counter_collection.csv" | head -1) $O/depthwise_pmc.json 8 > $O/depthwise_pmc.txt 2>&1
python tools/bench_dwmarch.py --n 8 --out $O/depthwise_cold_n8.json 2>/dev/null | grep -v amdgpu > $O/depthwise_cold_n8.txt
python tools/bench_dwmarch.py --n 64 --reps 12 --out $O/depthwise_cold_n64.json 2>/dev/null | grep -v amdgpu > $O/depthwise_cold_n64.txt
python tools/bench_kernels.py --n 64 --iters 10 --dw-only 2>/dev/null | grep -v amdgpu > $O/bench_kernels_n64.txt
cp $O/pmc_traffic.json profiles/${TAG%%_*}_pmc_traffic.json     # (bench.py quotes the newest profiles/rNN_pmc_traffic.json and checks its source hash)
python bench.py > $O/bench_final.json 2> $O/bench_final.err
{
for V in "--precision fp32-native" "--foml" "--adam" "--aspp" "--skip-decoding" "--augment" "--precision bf16" "--precision fp8" "--inner-batch 16" "--inner-batch 64" \
         "--precision bf16-storage" "--backbone efficientnet-b3 --shots 10 --inner-iters 20" "--backbone efficientnet-b3 --shots 10 --inner-iters 20 --precision bf16" \
         "--backbone efficientnet-b3 --shots 10 --inner-iters 20 --precision bf16-storage" \
         "--image-size 384" "--image-size 384 --precision fp8" "--tasks-per-gpu 8 --concurrent-tasks 4" "--tasks-per-gpu 8 --concurrent-tasks 4 --precision bf16"; do
  python bench.py $V --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-native-retime 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-75s %8.1f images/s  %7.2f ms/step  loss %.4f' % ('$V', d['value'], d['ms_per_step'], d['config']['final_loss']))"
done
for E in "MLIIS_FUSE_BN2=1" "MLIIS_NO_FUSE_HEAD=1"; do
  env $E python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-native-retime 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-75s %8.1f images/s  %7.2f ms/step  loss %.4f' % ('(environment) $E', d['value'], d['ms_per_step'], d['config']['final_loss']))"
done
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-native-retime 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-75s %8.1f images/s  %7.2f ms/step  loss %.4f' % ('(default, again: same box, end of the list)', d['value'], d['ms_per_step'], d['config']['final_loss']))"
} > $O/variants.txt 2>&1
rm -rf $O/pmc_fetch $O/pmc_write $O/dw_fetch $O/dw_write     # (hundreds of MB of per-dispatch rows; the aggregates above are what is kept)
du -sh $O; head -8 $O/profile.md; cat $O/depthwise_pmc.txt; cat $O/depthwise_cold_n8.txt; cat $O/depthwise_cold_n64.txt; cat $O/variants.txt; tail -c 300 $O/bench_final.err
