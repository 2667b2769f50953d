#!/bin/bash
# GPU box: rocprofv3 evidence for k_cycle on configs 3 and 5 (kernel stats, then one PMC pass per counter)
set -e
export TMPDIR=/tmp
OUT=gpurun_out/r03_prof_cycle
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/pmc_workloads_cycle.py 40 > $OUT/stats.log 2> $OUT/stats.err
echo "stats done"
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_$ctr -- python3 tools/pmc_workloads_cycle.py 8 > $OUT/pmc_$ctr.log 2> $OUT/pmc_$ctr.err
  echo "pmc $ctr done"
done
