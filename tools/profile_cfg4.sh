#!/bin/bash
# GPU box: rocprofv3 evidence for config 4's step kernels at 2^19 chains (kernel stats, then one PMC pass per counter).
set -e
export TMPDIR=/tmp
OUT=gpurun_out/r03_prof_cfg4
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/pmc_workloads_cfg4.py 300 > $OUT/stats.log 2> $OUT/stats.err
echo "stats done"
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_$ctr -- python3 tools/pmc_workloads_cfg4.py 20 > $OUT/pmc_$ctr.log 2> $OUT/pmc_$ctr.err
  echo "pmc $ctr done"
done
