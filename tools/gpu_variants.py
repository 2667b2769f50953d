"""Dev tool (GPU box): A/B k_step variants built by tools/build_variant.sh.  One subprocess per (variant, round),
interleaved; reports median/min ms per launch for BASELINE config 2 at 1 sweep per launch (and 32 fused)."""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, os, sys
sys.path.insert(0, %r)
import metropolisengine_amd as me
n = 1 << int(os.environ.get("ME_CHAINS_LOG2", "20"))
e = me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 16, None, temp=1.0, n_chains=n, seed=2026)
e.time_steps(300, 1)
k1 = [e.time_steps(300, 1) / 300 for _ in range(5)]
k32 = [e.time_steps(10, 32) / 10 for _ in range(3)]
print(json.dumps({"k1": k1, "k32": k32}))
''' % ROOT

variants = sys.argv[1:] or sorted(f[:-3] for f in os.listdir(os.path.join(ROOT, "tools", "variants")) if f.endswith(".so"))
rounds = int(os.environ.get("ROUNDS", "3"))
res = {v: {"k1": [], "k32": []} for v in variants}
for r in range(rounds):
    for v in variants:
        env = dict(os.environ, METROPOLIS_HIP_LIB=os.path.join(ROOT, "tools", "variants", v + ".so"))
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        if out.returncode != 0:
            print(v, "FAILED", out.stderr[-500:], flush=True)
            continue
        rec = json.loads(out.stdout.strip().splitlines()[-1])
        res[v]["k1"] += rec["k1"]
        res[v]["k32"] += rec["k32"]
n = 1 << int(os.environ.get("ME_CHAINS_LOG2", "20"))
for v in variants:
    if not res[v]["k1"]:
        continue
    m1, m32 = statistics.median(res[v]["k1"]), statistics.median(res[v]["k32"])
    print("%-12s K=1: median %.4f ms (min %.4f) -> %.0f GB/s   K=32: median %.4f ms -> %.3e chain-steps/s"
          % (v, m1, min(res[v]["k1"]), 144 * n / m1 / 1e6, m32, n * 32 / (m32 * 1e-3)), flush=True)
