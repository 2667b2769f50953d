#!/bin/bash
# GPU box, round 3 batch 1: variants of the config-4 float64 kernel and the headline kernel, the float64 covariance probes, the ramp
export TMPDIR=/tmp
echo "== cov_probe_f64"; timeout -k 5 120 tools/variants/cov_probe_f64
echo "== rows_probe_f64"; timeout -k 5 120 tools/variants/rows_probe_f64
echo "== time_f64_ramp"; timeout -k 5 200 python tools/dev/time_f64_ramp.py
echo "== dense64 variants"; timeout -k 5 400 python tools/dev/time_dense64_variants.py d64_base d64_nt1 d64_nt2 d64_nt3 d64_mem d64_memnt
echo "== headline variants"; timeout -k 5 300 python tools/dev/time_f64_variants.py h_base h_w5 h_w6
