#!/bin/bash
# GPU box, round 3 batch 3: issue-priority stagger (headline, dense-64 float64), streamed float64 per-chain factor, state layout probe
export TMPDIR=/tmp
echo "== state_layout_probe64"; timeout -k 5 120 tools/variants/state_layout_probe64
echo "== headline priority variants"; timeout -k 5 500 python tools/dev/time_f64_variants.py h_base h_p0 h_p2 h_p4 h_p6 h_p8 h_p10 h_p12
echo "== dense64 priority"; timeout -k 5 300 python tools/dev/time_dense64_variants.py d64_base d64_prio
echo "== per-chain factor (16,0) float64: registers vs streamed"
for r in 0 1; do for v in pc16_base pc16_stream; do METROPOLIS_HIP_LIB=tools/variants/$v.so timeout -k 5 120 python tools/dev/time_perchain16.py f64; done; done
