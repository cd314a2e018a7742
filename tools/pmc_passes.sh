#!/bin/bash
# usage: tools_pmc.sh <outdir-suffix> ; runs PMC passes of bench.py (3 steps) and writes CSVs under gpurun_out/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
S=$1
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $c | cut -d" " -f1)
  (timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${S}_$n -o pmc -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline) > $R/gpurun_out/pmc_${S}_$n.log 2>&1
done
