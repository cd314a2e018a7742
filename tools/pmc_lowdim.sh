#!/bin/bash
# PMC passes over the lowdim bench; prints per-kernel means.  usage: tools/pmc_lowdim.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAIT_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $c | cut -d" " -f1)
  rm -rf /tmp/pmcld_$n
  (timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcld_$n -o pmc -- python $R/bench.py --workload lowdim --steps 3 --warmup 1 --no-cpu-baseline) > /tmp/pmcld_$n.log 2>&1
  python - <<PY
import csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
try:
    for r in csv.DictReader(open("/tmp/pmcld_$n/pmc_counter_collection.csv")):
        acc[r["Kernel_Name"][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
except Exception as e:
    print("no data for $n:", e)
for k, d in acc.items():
    if "mel_" in k or "synth_comp" in k or "noise_stats" in k:
        print("%-44s" % k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
done
