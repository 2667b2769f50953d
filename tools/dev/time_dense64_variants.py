"""Dev A/B (GPU box): variants of the (64,0) kernel set built by tools/build_variant.sh, timed in interleaved child processes.
    python tools/dev/time_dense64_variants.py d64_u1 d64_u2 ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import metropolisengine_amd as me
m = np.random.default_rng(5).standard_normal((64, 64))
e4 = me.MetropolisEngine(me.DenseQuadratic(m @ m.T / 64 + np.identity(64)), None, [0.0] * 64, None, temp=1.0,
                         n_chains=1 << 19, seed=2026, cov_mode="fixed", dtype="f64", sampling_width=0.2)
e4.time_steps(20, 1)
ms = min(e4.time_steps(50, 1) for _ in range(3)) / 50
msf = min(e4.time_steps(5, 10) for _ in range(2)) / 50
print("%%.1f %%.1f" %% (ms * 1e3, msf * 1e3))
''' % ROOT
for rnd in range(2):
    for name in sys.argv[1:]:
        env = dict(os.environ, METROPOLIS_HIP_LIB=os.path.join(ROOT, "tools", "variants", name + ".so"))
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print(rnd, name, "one-sweep us / fused us per sweep:", out.stdout.strip() or out.stderr[-300:], flush=True)
