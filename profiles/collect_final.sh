#!/bin/bash
# Final collection of a round on one box (run through gpurun from the repository root; two calls fit gpurun's 20-minute limit):
#   bash profiles/collect_final.sh r06 a   -> GPU test suite + profile set (kernel trace, three PMC passes, bench line per workload)
#   bash profiles/collect_final.sh r06 b   -> extras (other workloads, call latencies, emulated strong scaling, fuzz), default bench line, smoke()
# then, here:  bash profiles/install_final.sh r06 f
set -o pipefail
ROUND=${1:-r06}; PART=${2:-a}
F=gpurun_out/final_$ROUND
mkdir -p $F
if [ $PART = a ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > $F/tests_gpu.log 2>&1 < /dev/null; echo "gpu tests rc=$?" | tee $F/summary_a.txt
  tail -2 $F/tests_gpu.log | tee -a $F/summary_a.txt
  bash profiles/collect.sh $ROUND f > $F/collect.log 2>&1 < /dev/null || echo "collect failed" | tee -a $F/summary_a.txt
else
  bash profiles/collect_extra.sh $ROUND f > $F/collect_extra.log 2>&1 < /dev/null || echo "collect_extra failed" | tee $F/summary_b.txt
  python3 bench.py > $F/default_bench_line.json 2> $F/default_bench_line.err < /dev/null
  tail -c 600 $F/default_bench_line.json
  python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $F/summary_b.txt
fi
