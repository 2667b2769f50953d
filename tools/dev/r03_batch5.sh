#!/bin/bash
# GPU box, round 3 batch 5: 16-chain tiles for the float64 dense-64 kernel; pooled-moment kernel times under rocprof
export TMPDIR=/tmp
echo "== parity, 32-chain tiles"; timeout -k 5 300 python -m pytest tests/test_gpu_dense_f64.py -q 2>&1 | tail -3
echo "== parity, 16-chain tiles"; METROPOLIS_DENSE64_F64_TILE=16 timeout -k 5 300 python -m pytest tests/test_gpu_dense_f64.py tests/test_gpu_f32_vs_f64.py tests/test_gpu_full_size.py -q 2>&1 | tail -3
for r in 0 1; do
echo "== timing TC=32"; METROPOLIS_DENSE64_F64_TILE=32 timeout -k 5 200 python tools/dev/time_dense64.py 2>&1 | grep f64
echo "== timing TC=16 (768 threads)"; METROPOLIS_DENSE64_F64_TILE=16 timeout -k 5 200 python tools/dev/time_dense64.py 2>&1 | grep f64
echo "== timing TC=16 (1024 threads, spills)"; METROPOLIS_HIP_LIB=tools/variants/d64_t1024.so METROPOLIS_DENSE64_F64_TILE=16 timeout -k 5 200 python tools/dev/time_dense64.py 2>&1 | grep f64
done
echo "== pooled kernels under rocprof"; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_prof_pool -- python3 tools/dev/time_pool.py > gpurun_out/r03_prof_pool.log 2>&1; grep -h "k_pool\|copy" gpurun_out/r03_prof_pool/*/*kernel_stats.csv | cut -c1-160
