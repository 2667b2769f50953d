#!/bin/bash
# GPU box: kernel-trace stats + FETCH_SIZE / WRITE_SIZE passes for the streamed-shape kernels (tools/pmc_workloads_streamed.py).
set -e
export TMPDIR=/tmp
OUT=gpurun_out/r02_prof_streamed
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/pmc_workloads_streamed.py > $OUT/stats.out 2> $OUT/stats.err
echo stats done
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_$ctr -- python3 tools/pmc_workloads_streamed.py > /dev/null 2> $OUT/pmc_$ctr.err
  echo pmc $ctr done
done
