#!/bin/bash
# GPU box, round 3 batch 4: tile-major state (64 parameters), priority stagger + streamed float64 factor as defaults, new pooled stage 1
export TMPDIR=/tmp
echo "== dense64 default build"; timeout -k 5 200 python tools/dev/time_dense64.py
echo "== pooled moments"; timeout -k 5 200 python tools/dev/time_pool.py
echo "== per-chain (16,0)"; timeout -k 5 200 python tools/dev/time_perchain16.py f64; timeout -k 5 200 python tools/dev/time_perchain16.py f32
echo "== measure 64"; timeout -k 5 200 python tools/dev/time_measure64.py
timeout -k 10 280 python bench.py > gpurun_out/r03_bench_c.json 2> gpurun_out/r03_bench_c.err; echo "bench rc=$?"
