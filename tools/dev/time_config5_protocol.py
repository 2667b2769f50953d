"""Dev timing (GPU box): protocols.config5 on a fresh float32 engine (what bench.py's config5 block does), to look for what makes the
one-launch overlapped form slow under OMP_NUM_THREADS=1.  argv[1]: comma list of forms to run before it (none,sync,overlap)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if os.environ.get("WITH_TORCH"):
    import torch  # noqa: F401
import metropolisengine_amd as me
from metropolisengine_amd import distributed, protocols
src = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "examples", "user_energy_cylinder.h")
n = 1 << 18
e = me.MetropolisEngine(me.UserEnergy("cylinder", src, (1.0, 0.5, 1.0)), me.AbsReal0AtLeast(1.0), [0.1, 0.0], [0.05] * 7,
                        temp=0.1, n_chains=n, seed=2026)
distributed.init_native_comm(e, rank=0, world_size=1, id_file="/tmp/me_uid_%d" % os.getpid())
protocols.cycle_protocol(e, 60, 10, "none", fused=True)
distributed.pooled_statistics(e, backend="rccl-native")
for form in [f for f in (sys.argv[1] if len(sys.argv) > 1 else "").split(",") if f]:
    dt, _ = protocols.cycle_protocol(e, 200, 10, form, backend="rccl-native")
    print("  11-launch form %-8s %.1f us per cycle" % (form, dt / 200 * 1e6))
dt, _ = protocols.cycle_protocol(e, 200, 10, "overlap", backend="rccl-native", one_launch=True)
print("OMP_NUM_THREADS=%s forms=%s: one-launch overlapped %.1f us per cycle" % (os.environ.get("OMP_NUM_THREADS"), sys.argv[1:] , dt / 200 * 1e6), flush=True)
