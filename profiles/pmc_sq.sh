#!/bin/bash
# SQ counter pass of one workload (run through gpurun from the repository root): bash profiles/pmc_sq.sh <tag> <workload> [bench args]
# -> gpurun_out/<tag>_<workload>_sq.txt (per-kernel, per-dispatch averages).  The program follows `--` directly.
set -o pipefail
TAG=$1; W=$2; shift 2
REPO=$PWD
OUT=$REPO/gpurun_out/pmc_${TAG}_$W
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d "$OUT" -- python3 $REPO/bench.py --no-cpu-baseline --workload $W --steps 2 --warmup 1 "$@" > "$OUT.log" 2>&1 || exit 1
python3 $REPO/profiles/summarize_pmc.py "$OUT" > "$REPO/gpurun_out/${TAG}_${W}_sq.txt" 2>&1
rm -rf "$OUT"
