#!/bin/bash
# GPU box, round 3 final evidence: the bench line in its three launch forms, rocprofv3 stats + PMC passes of the final build
export TMPDIR=/tmp
bash tools/dev/r03_batch2.sh
timeout -k 10 300 bash tools/profile_cfg4.sh; echo "profile_cfg4 rc=$?"
timeout -k 10 200 python tools/kernel_roofline.py f32 f64 > gpurun_out/r03_kernel_roofline.md 2> gpurun_out/r03_kernel_roofline.err; echo "kernel_roofline rc=$?"
