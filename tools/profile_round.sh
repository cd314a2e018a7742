#!/bin/bash
# One round's rocprofv3 evidence, on the GPU box:   tools/profile_round.sh r02_v1
#   gpurun_out/prof_<tag>/bench.json          python bench.py (full line)
#   gpurun_out/prof_<tag>/kernel_stats.csv    rocprofv3 --kernel-trace --stats of bench.py --steps 40 --warmup 5 --streams 1 --no-cpu-baseline --no-e2e --traffic none
#                                             (one step at a time: with the default two streams consecutive steps overlap and a
#                                             launch's duration includes the time it shares the device with its neighbour)
#   gpurun_out/prof_<tag>/bench_profiled.json the JSON line of THAT profiled process: its roofline (HIP events) and the
#                                             kernel_stats.csv averages come from the same launches on the same box
#   gpurun_out/prof_<tag>/pmc_<set>/...       separate --pmc passes (with --kernel-trace only) of bench.py --steps 3 --warmup 1 ...
#   gpurun_out/prof_<tag>/pmc_summary.json, traffic.json   tools/pmc_summary.py
# Copy what is to be judged into profiles/ afterwards (tools/pmc_summary.py --install does it for traffic.json).
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ -z "$SKIP_BENCH" ]; then
  (cd $R && timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err)
fi
if [ -z "$SKIP_STATS" ]; then
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o ks -- python $R/bench.py --steps 40 --warmup 5 --streams 1 --no-idle-probe --no-power --no-cpu-baseline --no-e2e --traffic none > $O/stats.log 2>&1
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
grep '^{"metric"' $O/stats.log | tail -1 > $O/bench_profiled.json
fi
for c in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS" \
         "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"; do  # (bench.py --pmc-child also launches the fused configs[3] kernel)
  n=$(echo $c | cut -d" " -f1)
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$n -o pmc -- python $R/bench.py --steps 3 --warmup 1 --streams 1 --no-power --no-cpu-baseline --no-e2e --traffic none > $O/pmc_$n.log 2>&1
done
cd $R && python tools/pmc_summary.py $O
find $O -name "*.db" -delete; rm -rf $O/stats $O/pmc_*/  # (the summaries are what is kept: gpurun_out travels back only below 64 MiB)
