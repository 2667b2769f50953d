#!/bin/bash
# GPU box, round 3 batch 6: k_cycle with the measure half's loads requested before the sweeps (levels 0 / 1 / 2)
export TMPDIR=/tmp
for r in 0 1; do
  for lv in 0 1 2; do METROPOLIS_HIP_LIB=tools/variants/c44_p$lv.so timeout -k 5 200 python tools/dev/time_cycle.py 4 4; done
  for lv in 0 1 2; do METROPOLIS_HIP_LIB=tools/variants/c27_p$lv.so timeout -k 5 200 python tools/dev/time_cycle.py 2 7; done
done
