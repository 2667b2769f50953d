"""Dev (GPU box): k_step_dense64_f64 at 2^17 .. 2^21 chains, one sweep per launch and ten fused sweeps -- the fixed cost of a
launch (prologue, first load, last store) against the cost per tile.
    python tools/dev/time_dense64_ramp.py [variant ...]     (variants built by tools/build_variant.sh; none = the installed library)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import metropolisengine_amd as me
m = np.random.default_rng(5).standard_normal((64, 64))
out = []
for lg in (17, 18, 19, 20, 21):
    e4 = me.MetropolisEngine(me.DenseQuadratic(m @ m.T / 64 + np.identity(64)), None, [0.0] * 64, None, temp=1.0,
                             n_chains=1 << lg, seed=2026, cov_mode="fixed", dtype="f64", sampling_width=0.2)
    e4.time_steps(20, 1)
    ms = min(e4.time_steps(50, 1) for _ in range(3)) / 50
    msf = min(e4.time_steps(5, 10) for _ in range(2)) / 50
    out.append("2^%%d: %%.1f / %%.1f" %% (lg, ms * 1e3, msf * 1e3))
    del e4
print("   ".join(out))
''' % ROOT
for name in (sys.argv[1:] or [None]):
    env = dict(os.environ)
    if name:
        env["METROPOLIS_HIP_LIB"] = os.path.join(ROOT, "tools", "variants", name + ".so")
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print(name or "installed", "one-sweep us / fused us per sweep:", out.stdout.strip() or out.stderr[-400:], flush=True)
