#!/bin/bash
# dev batch (GPU box): SQ counters of k_step_dense64_f64 with and without its state traffic
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in d64r_p2 d64r_nomem; do
  export METROPOLIS_HIP_LIB=$R/tools/variants/$v.so
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${v}_a -- python3 $R/tools/dev/pmc_dense64.py > $R/gpurun_out/pmc_${v}_a.log 2>&1 &&
  timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${v}_b -- python3 $R/tools/dev/pmc_dense64.py > $R/gpurun_out/pmc_${v}_b.log 2>&1 &&
  timeout -k 10 200 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_IFETCH SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${v}_c -- python3 $R/tools/dev/pmc_dense64.py > $R/gpurun_out/pmc_${v}_c.log 2>&1 || exit 1
  for p in a b c; do echo "== $v pass $p"; python3 $R/tools/dev/pmc_dense64_summary.py $R/gpurun_out/pmc_${v}_$p; done
done > $R/gpurun_out/r03_pmc_d64r.txt 2>&1
cat $R/gpurun_out/r03_pmc_d64r.txt
rm -rf $R/gpurun_out/pmc_d64r_*_[abc]
