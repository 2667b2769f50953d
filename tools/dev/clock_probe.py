"""Dev (GPU box): shader clock and package power (rocm-smi) while k_step_dense64_f64 runs back to back -- does the clock
drop when the state traffic is added to the arithmetic?
    python tools/dev/clock_probe.py variant:sweeps_per_launch ..."""
import os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, time, numpy as np
sys.path.insert(0, %r)
import metropolisengine_amd as me
sweeps = int(sys.argv[1])
m = np.random.default_rng(5).standard_normal((64, 64))
e4 = me.MetropolisEngine(me.DenseQuadratic(m @ m.T / 64 + np.identity(64)), None, [0.0] * 64, None, temp=1.0,
                         n_chains=1 << 19, seed=2026, cov_mode="fixed", dtype="f64", sampling_width=0.2)
e4.time_steps(20, sweeps)
t0 = time.time(); total = 0.0; launches = 0
while time.time() - t0 < 5.0:
    total += e4.time_steps(2000 // sweeps, sweeps); launches += 2000 // sweeps
print("us per sweep %%.1f" %% (total / launches / sweeps * 1e3))
''' % ROOT
def sample():
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    sclk = re.findall(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
    power = re.findall(r"Power \(W\): ([\d.]+)", out)
    return (sclk[:1] or ["?"])[0], (power[:1] or ["?"])[0]
print("idle", sample(), flush=True)
for spec in sys.argv[1:]:
    name, sweeps = spec.split(":")
    env = dict(os.environ, METROPOLIS_HIP_LIB=os.path.join(ROOT, "tools", "variants", name + ".so"))
    child = subprocess.Popen([sys.executable, "-c", CHILD, sweeps], env=env, stdout=subprocess.PIPE, text=True)
    time.sleep(2.5)     # import + engine construction
    seen = []
    while child.poll() is None and len(seen) < 6:
        seen.append(sample())
        time.sleep(0.3)
    out = child.communicate()[0].strip()
    print(spec, out, "sclk MHz / W:", seen, flush=True)
