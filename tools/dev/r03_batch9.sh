#!/bin/bash
# dev batch (GPU box): config 5's blocks of bench.py plain and under torch.distributed.run (host-bound overlapped loop)
export METROPOLIS_BENCH_ONLY='config5'
for rep in 1 2; do
  python bench.py --gpus 1 > gpurun_out/b9_plain_$rep.json 2> gpurun_out/b9_plain_$rep.err
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 > gpurun_out/b9_torchrun_$rep.json 2> gpurun_out/b9_torchrun_$rep.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/b9_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as ex:
        print(f, "unreadable", ex); continue
    oc = d["other_configs"]
    print(f, {k: {kk: "%.3g" % vv for kk, vv in v.items() if kk.startswith("chain_steps_per_s")} for k, v in oc.items() if k.startswith("config5")})
PY
