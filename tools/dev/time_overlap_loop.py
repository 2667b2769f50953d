"""Dev timing (GPU box): where the host time of config 5's one-launch-cycle loop with an overlapped native all-reduce goes
(engine.cycle / pooled_moments_end / moments_to_statistics / pooled_moments_allreduce_begin), e.g. under OMP_NUM_THREADS=1."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if os.environ.get("WITH_TORCH"):
    import torch  # noqa: F401
import metropolisengine_amd as me
from metropolisengine_amd import distributed
src = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "examples", "user_energy_cylinder.h")
n = 1 << 18
e = me.MetropolisEngine(me.UserEnergy("cylinder", src, (1.0, 0.5, 1.0)), me.AbsReal0AtLeast(1.0), [0.1, 0.0], [0.05] * 7,
                        temp=0.1, n_chains=n, seed=2026)
distributed.init_native_comm(e, rank=0, world_size=1, id_file="/tmp/me_uid_%d" % os.getpid())
for _ in range(60):
    e.cycle(10)
e.pooled_moments_allreduce_begin()
t = {"cycle": 0.0, "end": 0.0, "stats": 0.0, "begin": 0.0}
cycles = 300
t0 = time.perf_counter()
for _ in range(cycles):
    a = time.perf_counter(); e.cycle(10)
    b = time.perf_counter(); m = e.pooled_moments_end()
    c = time.perf_counter(); distributed.moments_to_statistics(m, 2, 7)
    d = time.perf_counter(); e.pooled_moments_allreduce_begin()
    f = time.perf_counter()
    t["cycle"] += b - a; t["end"] += c - b; t["stats"] += d - c; t["begin"] += f - d
e.pooled_moments_end(); e.sync()
total = time.perf_counter() - t0
print("OMP_NUM_THREADS=%s: %.1f us per cycle; host us per call: %s" % (os.environ.get("OMP_NUM_THREADS"), total / cycles * 1e6,
      {k: round(v / cycles * 1e6, 1) for k, v in t.items()}), flush=True)
