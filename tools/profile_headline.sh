#!/bin/bash
# GPU box: rocprofv3 evidence for the headline kernels (run from the repo root through gpurun).
#   kernel-trace stats of the float64 / float32 k_step at 2^20 and 2^22 chains, then the FETCH_SIZE and WRITE_SIZE
#   passes (one counter per run, --kernel-trace only, as MI355X_MICROARCH.md prescribes).  Output: gpurun_out/r02_prof/
set -e
export TMPDIR=/tmp
OUT=${OUT:-gpurun_out/r03_prof}
mkdir -p $OUT
COMMON="--gpus 1 --cpu-seconds 0 --fused-sweeps 0 --extras 0 --hbm-chains-log2 0"
for dt in f64 f32; do
  for lg in 20 22; do
    tag=${dt}_2p${lg}
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$tag -- python3 bench.py $COMMON --dtype $dt --chains-log2 $lg --steps 1000 --warmup 200 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
    echo "stats $tag done"
    for ctr in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_${ctr}_$tag -- python3 bench.py $COMMON --dtype $dt --chains-log2 $lg --steps 50 --warmup 20 > /dev/null 2> $OUT/pmc_${ctr}_$tag.err
      echo "pmc $ctr $tag done"
    done
  done
done
