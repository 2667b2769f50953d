"""Dev A/B (GPU box): variants of the (16,0) kernel set, float64 headline kernel, interleaved child processes."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for rnd in range(3):
    for name in sys.argv[1:]:
        env = dict(os.environ, METROPOLIS_HIP_LIB=os.path.join(ROOT, "tools", "variants", name + ".so"))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dev", "time_f64.py")], env=env, capture_output=True, text=True)
        print(rnd, name, out.stdout.strip() or out.stderr[-300:], flush=True)
