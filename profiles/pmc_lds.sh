#!/bin/bash
# LDS-side counters of the team sweeps: separate --pmc passes (never together with a trace other than --kernel-trace)
OUT=$PWD/gpurun_out/r5_lds
mkdir -p $OUT
R=$PWD
cd /tmp && export TMPDIR=/tmp
for W in ${WLS:-c5 c2a}; do
  BENCH="python3 $R/bench.py --no-cpu-baseline --workload $W --steps 2 --warmup 1"
  i=0
  for SET in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN" "SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/pmc_${W}_$i -- $BENCH > $OUT/${W}_pmc_$i.log 2>&1 || echo "set $i failed for $W (see log)"
  done
  python3 $R/profiles/summarize_pmc.py $OUT/pmc_${W}_1 $OUT/pmc_${W}_2 $OUT/pmc_${W}_3 > $OUT/${W}_lds_pmc_summary.txt 2>&1
  rm -rf $OUT/pmc_${W}_1 $OUT/pmc_${W}_2 $OUT/pmc_${W}_3
  grep "k_sweep_duo\|k_dense_fused" $OUT/${W}_lds_pmc_summary.txt | head -3 | cut -c1-900
done
