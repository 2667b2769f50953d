"""Dev A/B (GPU box): k_step<double,16,0,EnergyDiag,identity> at 2^20 chains, variants built by tools/build_variant.sh.
    python tools/dev/time_diag16.py d16_old d16_new"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import metropolisengine_amd as me
out = []
for dtype in ("f64", "f32"):
    e = me.MetropolisEngine(me.DiagQuadratic(tuple(np.linspace(0.5, 2.0, 16))), None, [0.0] * 16, None, temp=1.0,
                            n_chains=1 << 20, seed=2026, cov_mode="fixed", dtype=dtype)
    e.time_steps(20, 1)
    ms = min(e.time_steps(100, 1) for _ in range(3)) / 100
    msf = min(e.time_steps(5, 32) for _ in range(2)) / 160
    out.append("%%s %%.1f / %%.1f" %% (dtype, ms * 1e3, msf * 1e3))
print("   ".join(out))
''' % ROOT
for rnd in range(2):
    for name in sys.argv[1:]:
        env = dict(os.environ, METROPOLIS_HIP_LIB=os.path.join(ROOT, "tools", "variants", name + ".so"))
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print(rnd, name, "one-sweep us / fused us per sweep:", out.stdout.strip() or out.stderr[-300:], flush=True)
