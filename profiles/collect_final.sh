#!/bin/bash
# final collection of round 5 on one box: GPU test suite, profile set, extras, default bench line
set -o pipefail
mkdir -p gpurun_out/final_r05
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/final_r05/tests_gpu.log 2>&1; echo "gpu tests rc=$?" | tee gpurun_out/final_r05/summary.txt
tail -2 gpurun_out/final_r05/tests_gpu.log | tee -a gpurun_out/final_r05/summary.txt
bash profiles/collect.sh r05 f > gpurun_out/final_r05/collect.log 2>&1 || echo "collect failed" | tee -a gpurun_out/final_r05/summary.txt
bash profiles/collect_extra.sh r05 f > gpurun_out/final_r05/collect_extra.log 2>&1 || echo "collect_extra failed" | tee -a gpurun_out/final_r05/summary.txt
python3 bench.py > gpurun_out/final_r05/default_bench_line.json 2> gpurun_out/final_r05/default_bench_line.err
tail -c 600 gpurun_out/final_r05/default_bench_line.json
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a gpurun_out/final_r05/summary.txt
