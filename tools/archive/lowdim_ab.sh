#!/bin/bash
# usage: tools/lowdim_ab.sh "<flags of variant 1>" "<flags of variant 2>" ...  ("" = default build)
# Rebuilds libmagphase_hip.so with each flag set and prints the per-kernel averages of the lowdim bench (rocprofv3),
# all on the box this runs on; restores the default build at the end.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for fl in "$@"; do
  (cd $R && python -m magphase_amd.build --force $fl > /dev/null 2>&1)
  rm -rf /tmp/ldab
  (timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ldab -o ld -- python $R/bench.py --workload lowdim --steps 10 --warmup 2 --no-cpu-baseline) > /tmp/ldab.log 2>&1
  echo "== variant: [$fl]"
  python - <<PY
import csv
for r in csv.DictReader(open("/tmp/ldab/ld_kernel_stats.csv")):
    if float(r["AverageNs"]) > 40000: print("  %-46s %8.1f us" % (r["Name"][:46], float(r["AverageNs"]) / 1e3))
PY
done
(cd $R && python -m magphase_amd.build --force > /dev/null 2>&1)
