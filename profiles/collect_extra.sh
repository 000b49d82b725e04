#!/bin/bash
# Secondary measurements of a round (run through gpurun from the repository root): bash profiles/collect_extra.sh r03 a
#   -> gpurun_out/extra_r03_a/{other_workloads,call_latencies,dense_host_calls,bench_c2b_20000atoms,emulated_strong_w8_<w>[_1stream]}.json, fuzz.log
ROUND=${1:-r06}; TAG=${2:-x}
OUT=gpurun_out/extra_${ROUND}_${TAG}
mkdir -p $OUT
python3 profiles/other_workloads.py > $OUT/other_workloads.json 2> $OUT/other_workloads.err
python3 profiles/call_latencies.py > $OUT/call_latencies.json 2> $OUT/call_latencies.err
python3 profiles/dense_host_calls.py > $OUT/dense_host_calls.json 2> $OUT/dense_host_calls.err
python3 bench.py --workload c2b --dense-atoms 20000 > $OUT/bench_c2b_20000atoms.json 2> $OUT/bench_c2b_20000atoms.err
for W in c2a c5 c3; do
  python3 bench.py --workload $W --emulate-world 8 --no-cpu-baseline > $OUT/emulated_strong_w8_$W.json 2> $OUT/emulated_strong_w8_$W.err
  python3 bench.py --workload $W --emulate-world 8 --streams 1 --no-cpu-baseline > $OUT/emulated_strong_w8_${W}_1stream.json 2> $OUT/emulated_strong_w8_${W}_1stream.err
done
LCHD_FUZZ_SEEDS=${LCHD_FUZZ_SEEDS:-8000} timeout -k 10 600 python3 -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > $OUT/fuzz.log 2>&1
tail -2 $OUT/fuzz.log
ls -la $OUT
