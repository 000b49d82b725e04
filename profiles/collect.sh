#!/bin/bash
# Collect the per-round profile artefacts on the MI355X box (run through gpurun from the repository root):
#   bash profiles/collect.sh r03 a [workloads...]   -> gpurun_out/prof_r03_a/{<w>_kernel_stats.csv, <w>_pmc_summary.txt, <w>_bench.json}
# Kernel trace and the PMC passes are separate runs (never --pmc together with a trace other than --kernel-trace); the program
# itself follows `--` (no env / bash -c hop under the profiler).
set -o pipefail
ROUND=${1:-r06}; TAG=${2:-x}; shift 2
WLS=${@:-c2a c5 c4 c3 c2b}
REPO=$PWD
OUT=$REPO/gpurun_out/prof_${ROUND}_${TAG}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for W in $WLS; do
  STEPS=10; [ $W = c4 ] && STEPS=3
  BENCH="python3 $REPO/bench.py --no-cpu-baseline --workload $W"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$W" -- $BENCH --steps $STEPS --warmup 2 > "$OUT/${W}_stats.log" 2>&1 || exit 1
  cp "$(find "$OUT/stats_$W" -name '*kernel_stats.csv' | head -1)" "$OUT/${W}_kernel_stats.csv"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_$W" -- $BENCH --steps 2 --warmup 1 > "$OUT/${W}_pmc_fetch.log" 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_write_$W" -- $BENCH --steps 2 --warmup 1 > "$OUT/${W}_pmc_write.log" 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq_$W" -- $BENCH --steps 2 --warmup 1 > "$OUT/${W}_pmc_sq.log" 2>&1 || exit 1
  python3 $REPO/profiles/summarize_pmc.py "$OUT/pmc_fetch_$W" "$OUT/pmc_write_$W" "$OUT/pmc_sq_$W" > "$OUT/${W}_pmc_summary.txt" 2>&1
  python3 $REPO/profiles/make_traffic.py $W "$OUT/${W}_pmc_summary.txt" "$OUT/${W}_kernel_stats.csv" "$REPO/profiles/$ROUND" > "$OUT/traffic_$W.json" 2> "$OUT/${W}_traffic.err"
  if [ $W = c4 ] || [ $W = c5 ]; then python3 $REPO/profiles/make_traffic.py ${W}_env "$OUT/${W}_pmc_summary.txt" "$OUT/${W}_kernel_stats.csv" "$REPO/profiles/$ROUND" > "$OUT/traffic_${W}_env.json" 2>> "$OUT/${W}_traffic.err"; fi
  rm -rf "$OUT/stats_$W" "$OUT/pmc_fetch_$W" "$OUT/pmc_write_$W" "$OUT/pmc_sq_$W"
done
cd "$REPO"
for W in $WLS; do
  STEPS=10; [ $W = c4 ] && STEPS=5
  python3 bench.py --workload $W --steps $STEPS --warmup 3 > "$OUT/${W}_bench.log" 2>&1; tail -1 "$OUT/${W}_bench.log" > "$OUT/${W}_bench.json"
done
ls -la "$OUT"
