#!/bin/bash
# Collect the per-round profile artefacts on the MI355X box (run through gpurun from the repository root):
#   bash profiles/collect.sh r01 h      -> gpurun_out/prof_r01_h/{stats.csv, pmc_*.txt, bench_*.json}
# Kernel trace and the PMC passes are separate runs (never --pmc together with a trace other than --kernel-trace).
set -o pipefail
ROUND=${1:-r01}; TAG=${2:-x}
REPO=$PWD
OUT=$REPO/gpurun_out/prof_${ROUND}_${TAG}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH --steps 10 --warmup 2 > "$OUT/stats.log" 2>&1 || exit 1
cp "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)" "$OUT/rocprofv3_kernel_stats_bench_c2a.csv"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH --steps 2 --warmup 1 > "$OUT/pmc_fetch.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_write" -- $BENCH --steps 2 --warmup 1 > "$OUT/pmc_write.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/pmc_sq" -- $BENCH --steps 2 --warmup 1 > "$OUT/pmc_sq.log" 2>&1 || exit 1
for W in c4 c5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$W" -- $BENCH --workload $W --steps 3 --warmup 1 > "$OUT/stats_$W.log" 2>&1 || exit 1
  cp "$(find "$OUT/stats_$W" -name '*kernel_stats.csv' | head -1)" "$OUT/rocprofv3_kernel_stats_bench_$W.csv"
  rm -rf "$OUT/stats_$W"
done
cd "$REPO"
python3 profiles/summarize_pmc.py "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_sq" > "$OUT/pmc_summary_bench_c2a.txt" 2>&1
python3 bench.py > "$OUT/bench_c2a.log" 2>&1; tail -1 "$OUT/bench_c2a.log" > "$OUT/bench_c2a.json"
python3 bench.py --workload c5 > "$OUT/bench_c5.log" 2>&1; tail -1 "$OUT/bench_c5.log" > "$OUT/bench_c5.json"
python3 bench.py --workload c4 --steps 5 --warmup 1 > "$OUT/bench_c4.log" 2>&1; tail -1 "$OUT/bench_c4.log" > "$OUT/bench_c4.json"
rm -rf "$OUT/stats" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_sq"
ls -la "$OUT"
