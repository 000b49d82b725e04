#!/bin/bash
# Copy what profiles/collect_final.sh left under gpurun_out/ into profiles/<round>/ under the names the documents cite:
#   bash profiles/install_final.sh r06 f
ROUND=${1:-r06}; TAG=${2:-f}
P=gpurun_out/prof_${ROUND}_${TAG}; E=gpurun_out/extra_${ROUND}_${TAG}; F=gpurun_out/final_${ROUND}; D=profiles/$ROUND
mkdir -p $D
for W in c2a c2b c3 c4 c5; do
  cp $P/${W}_bench.json $D/${TAG}_bench_$W.json
  cp $P/${W}_pmc_summary.txt $D/${TAG}_pmc_summary_bench_$W.txt
  cp $P/${W}_kernel_stats.csv $D/${TAG}_rocprofv3_kernel_stats_bench_$W.csv
  cp $P/traffic_$W.json $D/traffic_$W.json
done
for W in c4 c5; do cp $P/traffic_${W}_env.json $D/traffic_${W}_env.json; done
for N in other_workloads call_latencies dense_host_calls emulated_strong_w8_c2a emulated_strong_w8_c2a_1stream emulated_strong_w8_c3 emulated_strong_w8_c3_1stream emulated_strong_w8_c5 emulated_strong_w8_c5_1stream; do
  cp $E/$N.json $D/${TAG}_$N.json
done
cp $E/bench_c2b_20000atoms.json $D/${TAG}_bench_c2b_20000atoms.json
cp $E/fuzz.log $D/${TAG}_fuzz_8000_seeds.txt
cp $F/tests_gpu.log $D/${TAG}_gpu_test_suite.txt
cp $F/default_bench_line.json $D/${TAG}_default_bench_line.json
ls $D | wc -l
