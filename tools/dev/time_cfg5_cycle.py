"""Dev timing (GPU box): the parts of one config-5 cycle (10 x step_all, measure, pooled moments) with a sync after each."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import metropolisengine_amd as me
src = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "examples", "user_energy_cylinder.h")
n = 1 << 18
e = me.MetropolisEngine(me.UserEnergy("cylinder", src, (1.0, 0.5, 1.0)), me.AbsReal0AtLeast(1.0), [0.1, 0.0], [0.05] * 7,
                        temp=0.1, n_chains=n, seed=2026)
for _ in range(60):
    e.step_all(10); e.measure()
e.sync()
def t(f, reps=20):
    e.sync(); t0 = time.perf_counter()
    for _ in range(reps): f()
    e.sync(); return (time.perf_counter() - t0) / reps * 1e6
def ten():
    for _ in range(10): e.step_all()
print("10 x step_all(): %.1f us" % t(ten))
print("step_all(10):    %.1f us" % t(lambda: e.step_all(10)))
print("measure():       %.1f us" % t(e.measure))
print("pooled_moments:  %.1f us" % t(e.pooled_moments))
def cycle():
    ten(); e.measure(); e.pooled_moments()
print("cycle K=1:       %.1f us" % t(cycle))
def cycle_sync():
    ten(); e.sync(); e.measure(); e.sync(); e.pooled_moments()
print("cycle K=1 + syncs: %.1f us" % t(cycle_sync))
print("accept_stats:    %.1f us" % t(e.accept_stats))
