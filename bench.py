"""Headline benchmark: MC steps/sec (chains x sweeps), BASELINE.json config 2.

    python bench.py --gpus 1 --steps 4000 --warmup 1000
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload: 16 real parameters, isotropic quadratic energy E = sum x_i^2, T = 1,
2^20 chains PER GPU (weak scaling: chains are the independent units, sharded with no data-path collective),
sampling_width 0.05, target acceptance 0.3, seed 2026, identity proposal shape (measure() is never called).
One "step" = one ``step_all()`` = ONE launch of k_step advancing every chain by one propose -> energy ->
accept/reject -> width-adaptation sweep (``--sweeps K`` fuses K sweeps per launch; reported separately as
``fused``).  State is resident in HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_REAL = 16
HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy rate
BYTES_PER_CHAIN_STEP = 8 * N_REAL + 16   # fp32 r/w of x[16], energy, width: SURVEY.md 8(d), B_step(identity) = 144 B


def pmc_traffic(chains_log2, sweeps):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/r*_pmc_traffic.json): FETCH_SIZE
    (x2, gfx950 correction) + WRITE_SIZE, separate passes.  PMC collection cannot run inside the timed bench, so
    the figure is taken from the newest committed pass of this same workload; None for any other workload."""
    if chains_log2 != 20 or sweeps != 1:
        return None
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    with open(files[-1]) as fh:
        rec = json.load(fh)
    return rec.get("traffic_bytes_per_launch")


def issue_utilisation(torch, device, chain_sweeps_per_second):
    """Fraction of the chip's vector-issue cycles the fused-sweep run occupies (fused sweeps are
    instruction-bound, so this -- not the HBM fraction -- describes them).  Static VALU count of one sweep from the
    newest profiles/r*_kernel_valu.json (tools/valu_count.py); a wave64 instruction holds a SIMD for 4 cycles."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_valu.json")))
    if not files:
        return None
    with open(files[-1]) as fh:
        valu = json.load(fh)["valu_instructions"]
    props = torch.cuda.get_device_properties(device)
    clock_hz = float(getattr(props, "clock_rate", 0)) * 1e3 or 2.4e9       # kHz; MI355X peak engine clock otherwise
    simds = props.multi_processor_count * 4
    busy = chain_sweeps_per_second / 64.0 * valu * 4.0
    return {"valu_per_wavefront_sweep": valu, "simds": simds, "clock_GHz": clock_hz / 1e9,
            "frac": busy / (simds * clock_hz)}


def cpu_baseline(seconds):
    """Python restatement of the reference loop, one chain per process on the host cores (no GPU involved)."""
    cores = max(1, min(os.cpu_count() or 1, 16))
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_baseline", "--seconds", str(seconds),
                               "--seed", str(100 + i)], cwd=ROOT, env=env, stdout=subprocess.PIPE, text=True)
             for i in range(cores)]
    rate = 0.0
    for p in procs:
        out, _ = p.communicate()
        rec = json.loads(out.strip().splitlines()[-1])
        rate += rec["steps"] / rec["seconds"]
    return {"value": rate, "unit": "chain-steps/s", "cores": cores, "kind": "port",
            "sample": "oracle.reference_chain (pure-Python restatement of the reference step_all loop with "
                      "np.random.multivariate_normal), config 2 (16 real, E=sum x^2, T=1), one chain per process "
                      "on %d cores for %.0f s each; steps/s summed" % (cores, seconds)}


def cpu_baseline_c(seconds):
    """The plain-C restatement (oracle/c/me_oracle.c, float64, OpenMP over chains) on config 2: the strong CPU
    baseline next to the Python port.  Same Philox streams and arithmetic as the float64 GPU kernels."""
    from oracle.c_oracle import COracle
    cores = max(1, min(os.cpu_count() or 1, 16))
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    n = 1 << 16
    chains = COracle(N_REAL, 0, a=[1.0] * N_REAL, n_chains=n, seed=2026, temp=1.0, initial_real_params=[0.0] * N_REAL)
    chains.step(20)
    done = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        chains.step(50)
        done += 50
    dt = time.perf_counter() - t0
    return {"value": n * done / dt, "unit": "chain-steps/s", "cores": cores, "kind": "port",
            "sample": "oracle/c/me_oracle.c (float64, OpenMP), config 2 with 2^16 chains x %d sweeps in %.1f s, "
                      "acceptance %.3f" % (done, dt, chains.accepted / chains.proposed)}


def other_configs(me, device, chains_log2):
    """Informational side measurements (not the headline): the float64 build of the same kernel and the protocols of
    BASELINE.json configs 3-5, each through the public API incl. measure() launches."""
    import numpy as np
    out = {}
    n = 1 << chains_log2
    f64 = me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * N_REAL, None, temp=1.0, n_chains=n, seed=2026,
                              dtype="f64", device=device)
    f64.time_steps(50, 1)
    ms = f64.time_steps(200, 1) / 200
    out["config2_f64"] = {"chain_steps_per_s": n / (ms * 1e-3), "ms_per_launch": ms,
                          "state_GBps": 2 * BYTES_PER_CHAIN_STEP * n / (ms * 1e-3) / 1e9}
    del f64

    def protocol(engine, n_chains, steps_per_measure, cycles, warm_cycles):
        for _ in range(warm_cycles):
            engine.step_all(steps_per_measure)
            engine.measure()
        engine.sync()
        t0 = time.perf_counter()
        for _ in range(cycles):
            for _ in range(steps_per_measure):
                engine.step_all()
            engine.measure()
        engine.sync()
        dt = time.perf_counter() - t0
        return {"chain_steps_per_s": n_chains * steps_per_measure * cycles / dt,
                "acceptance_rate": engine.acceptance_rate(), "chains": n_chains,
                "protocol": "(%d x step_all + measure) x %d" % (steps_per_measure, cycles)}

    a = b = (1.0, 2.0, 4.0, 8.0)
    out["config3"] = protocol(me.MetropolisEngine(me.DiagQuadratic(a, b), None, [0.0] * 4, [0j] * 4, temp=1.0, n_chains=n,
                                                  seed=2026, device=device), n, 10, 100, 60)
    m = np.random.default_rng(5).standard_normal((64, 64))
    e4 = me.MetropolisEngine(me.DenseQuadratic(m @ m.T / 64 + np.identity(64)), None, [0.0] * 64, None, temp=1.0,
                             n_chains=n // 2, seed=2026, cov_mode="fixed", device=device)
    e4.time_steps(50, 1)
    ms = e4.time_steps(100, 1) / 100
    out["config4"] = {"chain_steps_per_s": (n // 2) / (ms * 1e-3), "ms_per_launch": ms, "chains": n // 2,
                      "state_GBps": (8 * 64 + 16) * (n // 2) / (ms * 1e-3) / 1e9, "kernel": "k_step_dense64_bf16x3"}
    del e4
    src = os.path.join(ROOT, "examples", "user_energy_cylinder.h")
    out["config5"] = protocol(me.MetropolisEngine(me.UserEnergy("cylinder", src, (1.0, 0.5, 1.0)),
                                                  me.AbsReal0AtLeast(1.0), [0.1, 0.0], [0.05] * 7, temp=0.1,
                                                  n_chains=n // 4, seed=2026, device=device), n // 4, 10, 100, 60)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--warmup", type=int, default=1000)
    ap.add_argument("--chains-log2", type=int, default=20, help="chains per GPU = 2**this")
    ap.add_argument("--sweeps", type=int, default=1, help="sweeps fused per launch in the headline run")
    ap.add_argument("--fused-sweeps", type=int, default=32, help="extra fused-sweep measurement (0 = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline sample length (0 = skip)")
    ap.add_argument("--extras", type=int, default=1, help="1: also time the float64 build of the headline kernel and "
                    "BASELINE configs 3-5 (single GPU only; informational, a few seconds)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))

    # CPU baseline first: child processes are started before this process touches the GPU
    cpu = cpu_c = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        cpu = cpu_baseline(args.cpu_seconds)
        try:
            cpu_c = cpu_baseline_c(min(args.cpu_seconds, 6.0))
        except (OSError, subprocess.CalledProcessError) as exc:      # the optional C baseline must not sink the bench
            cpu_c = {"value": None, "unit": "chain-steps/s", "kind": "port", "sample": "unavailable: %s" % exc}

    import torch
    import torch.distributed as dist
    import metropolisengine_amd as me
    from metropolisengine_amd.distributed import pooled_statistics

    # Rehearsal on a one-GPU box (never used by the driver): METROPOLIS_BENCH_REHEARSAL=1 puts every rank on GPU 0 and
    # uses gloo for the barrier / timing reduction / pooled moments, so the multi-rank code path can be exercised there.
    rehearsal = os.environ.get("METROPOLIS_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    distributed = "RANK" in os.environ           # launched by torch.distributed.run (also at --gpus 1)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    n_local = 1 << args.chains_log2
    engine = me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * N_REAL, None, sampling_width=0.05,
                                 target_acceptance=0.3, temp=1.0, n_chains=n_local, seed=2026, dtype="f32",
                                 device=local_rank, chain_offset=rank * n_local)

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n_launches, sweeps):
        """(wall seconds, device ms from HIP events on the engine's stream) of n_launches launches."""
        fence()
        t0 = time.perf_counter()
        dev_ms = engine.time_steps(n_launches, sweeps)     # enqueues, records events, waits for the stop event
        fence()
        wall = time.perf_counter() - t0
        if distributed:
            t = torch.tensor([wall, dev_ms], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall, dev_ms = float(t[0]), float(t[1])
        return wall, dev_ms

    if args.warmup > 0:
        engine.time_steps(args.warmup, args.sweeps)
    wall, dev_ms = timed(args.steps, args.sweeps)
    total_chain_steps = float(n_local) * world * args.steps * args.sweeps
    value = total_chain_steps / wall
    kernel_ms = dev_ms / args.steps                        # average launch duration (events, same timed region)
    algorithmic = BYTES_PER_CHAIN_STEP * n_local           # bytes one launch must move (state in, state out)
    achieved = algorithmic / (kernel_ms * 1e-3) / 1e9

    fused = None
    if args.fused_sweeps > 0:
        launches = max(4, args.steps // args.fused_sweeps)
        fwall, fdev = timed(launches, args.fused_sweeps)
        fused = {"sweeps_per_launch": args.fused_sweeps, "launches": launches,
                 "value": float(n_local) * world * launches * args.fused_sweeps / fwall, "unit": "chain-steps/s",
                 "ms_per_launch": fdev / launches,
                 "effective_state_GBps": algorithmic / (fdev / launches * 1e-3) / 1e9}
        if args.chains_log2 == 20 and world == 1:
            fused["valu_issue"] = issue_utilisation(torch, local_rank,
                                                    float(n_local) * launches * args.fused_sweeps / (fdev * 1e-3))

    stats = pooled_statistics(engine)                      # the one collective: RCCL all-reduce of pooled moments
    engine.sync()

    extras = None
    if world == 1 and args.extras:
        extras = other_configs(me, local_rank, args.chains_log2)

    if rank == 0:
        line = {
            "metric": "MC steps/sec (chains x sweeps) at 2^20 chains, 16 params",
            "value": value, "unit": "chain-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "config 2: 16 real params, isotropic quadratic E=sum x^2, T=1, 2^%d chains per GPU, "
                                   "identity proposal shape, %d sweep(s) per launch, Philox4x32-10 streams"
                                   % (args.chains_log2, args.sweeps),
                       "chains_per_gpu": n_local, "global_chains": n_local * world, "sweeps_per_launch": args.sweeps,
                       "parallelism": "chains sharded over %d GPU(s), no data-path collective" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": pmc_traffic(args.chains_log2, args.sweeps),
                         "kernel": "k_step<float,16,0,EnergyIso,identity>", "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": algorithmic,
                         "note": "144 B per chain-step (fp32 r/w of x[16], energy, width) x 2^%d chains / average "
                                 "launch duration from HIP events on the engine's stream over the timed region"
                                 % args.chains_log2},
            "cpu_baseline": cpu,
            "cpu_baseline_c": cpu_c,
            "fused": fused,
            "other_configs": extras,
            "acceptance_rate": stats["acceptance_rate"],
            "pooled_variance_mean": float(sum(stats["covariance"][i][i] for i in range(N_REAL)) / N_REAL),
        }
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
