#!/bin/bash
# GPU box, round 3 closing evidence: the bench line in its three launch forms and the per-kernel roofline table of the final build
export TMPDIR=/tmp
timeout -k 10 280 python bench.py > gpurun_out/r03_bench_n1.json 2> gpurun_out/r03_bench_n1.err; echo "bench n1 rc=$?"
timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 500 --warmup 100 --cpu-seconds 0 > gpurun_out/r03_bench_torchrun_n1_nccl.json 2> gpurun_out/r03_bench_torchrun_n1_nccl.err; echo "torchrun n1 rc=$?"
METROPOLIS_BENCH_REHEARSAL=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 300 --warmup 100 --cpu-seconds 0 --hbm-chains-log2 0 > gpurun_out/r03_bench_rehearsal_2ranks_gloo.json 2> gpurun_out/r03_bench_rehearsal_2ranks_gloo.err; echo "rehearsal rc=$?"
timeout -k 10 300 python tools/kernel_roofline.py f32 f64 > gpurun_out/r03_kernel_roofline.md 2> gpurun_out/r03_kernel_roofline.err; echo "kernel_roofline rc=$?"
