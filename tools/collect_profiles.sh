#!/bin/bash
# copies the summaries tools/final_profile.sh left under gpurun_out/<tag> into profiles/ (tracked), named per round
TAG=${1:-r06_final}; R=${2:-r06}
S=gpurun_out/$TAG
cp $S/trace/t_kernel_stats.csv profiles/${R}_final_kernel_stats.csv
cp $S/profile.md profiles/${R}_final_profile.md
cp $S/final_timeline.txt profiles/${R}_final_timeline.txt
cp $S/pmc_traffic.json profiles/${R}_pmc_traffic.json
cp $S/depthwise_pmc.json profiles/${R}_depthwise_pmc.json
cp $S/depthwise_pmc.txt profiles/${R}_depthwise_pmc.txt
cp $S/depthwise_cold_n8.txt profiles/${R}_depthwise_cold_n8.txt
cp $S/depthwise_cold_n64.txt profiles/${R}_depthwise_cold_n64.txt
cp $S/bench_kernels_n64.txt profiles/${R}_bench_kernels_n64.txt
cp $S/variants.txt profiles/${R}_variants.txt
[ -s $S/bench_final.json ] && cp $S/bench_final.json profiles/${R}_bench_final.json
ls -la profiles/${R}_*
