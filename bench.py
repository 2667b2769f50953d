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

The top-level ``value`` / ``dtype`` / ``roofline`` are the FLOAT64 run: float64 / complex128 is the reference's
arithmetic (metropolis_engine.py:41, :51), 288 B of state traffic per chain-step.  The float32 build of the same
kernels (144 B per chain-step, the production dtype) is the sibling block ``f32``.  ``roofline_hbm`` is the same
float64 kernel at 2^22 chains, where the state (604 MB) no longer fits the 256 MiB Infinity Cache.
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
BYTES_PER_CHAIN_STEP_F64 = 2 * BYTES_PER_CHAIN_STEP   # "fp64 state doubles every term"  = 288 B
INFINITY_CACHE_BYTES = 256 << 20
SIDE_LAUNCHES = 300                      # launches of every side measurement (independent of --steps)


def pmc_traffic(dtype, chains_log2, sweeps):
    """(HBM bytes per launch, source file) from the committed rocprofv3 --pmc passes (profiles/r*_pmc_traffic_<dtype>
    .json): FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE, separate passes.  PMC collection cannot run inside the
    timed bench, so the figure is a COMMITTED measurement of this same workload (the file is named in
    ``traffic_source``); (None, None) for any other workload."""
    if sweeps != 1:
        return None, None
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_%s_2p%d.json" % (dtype, chains_log2))))
    if not files and dtype == "f32" and chains_log2 == 20:
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))      # round-1 name
    if not files:
        return None, None
    with open(files[-1]) as fh:
        rec = json.load(fh)
    return rec.get("traffic_bytes_per_launch"), os.path.relpath(files[-1], ROOT)


def issue_utilisation(torch, device, chain_sweeps_per_second, suffix=""):
    """Fraction of the chip's vector-issue cycles the fused-sweep run occupies (fused sweeps are
    instruction-bound, so this -- not the HBM fraction -- describes them).  Static VALU count of one sweep from the
    newest profiles/r*_kernel_valu.json (tools/valu_count.py); a wave64 instruction holds a SIMD for 4 cycles."""
    import glob
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_valu%s.json" % suffix)))
    if not files:
        return None
    with open(files[-1]) as fh:
        valu = json.load(fh)["valu_instructions"]
    props = torch.cuda.get_device_properties(device)
    clock_hz = float(getattr(props, "clock_rate", 0)) * 1e3 or 2.4e9       # kHz; MI355X peak engine clock otherwise
    simds = props.multi_processor_count * 4
    busy = chain_sweeps_per_second / 64.0 * valu * 4.0
    return {"valu_per_wavefront_sweep": valu, "simds": simds, "clock_GHz": clock_hz / 1e9,
            "frac": busy / (simds * clock_hz),
            "note": "LOWER BOUND on vector-pipe occupancy, not a utilisation: static VALU count x 4 cycles per wave64 "
                    "instruction; float64, transcendental and 64-bit integer instructions take longer than 4 cycles"}


def cgroup_cpu_quota():
    """CPUs this process group may use according to its cgroup (v2 ``cpu.max`` / v1 ``cpu.cfs_quota_us``); None when
    unlimited.  A one-GPU box of the pool shows all 256 host CPUs but grants a 16-CPU share this way."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        if quota != "max":
            return max(1, int(float(quota) / float(period) + 0.5))
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
            quota, period = int(fq.read()), int(fp.read())
        if quota > 0:
            return max(1, int(quota / period + 0.5))
    except (OSError, ValueError):
        pass
    return None


def host_cores():
    """Cores the CPU baselines use: ALL the cores this process is allowed -- the smaller of its affinity mask and its
    cgroup CPU quota (more runnable processes than the quota only thrash) -- unless METROPOLIS_BENCH_CPU_CORES caps it.
    The JSON states this number next to os.cpu_count() and the quota."""
    try:
        avail = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        avail = os.cpu_count() or 1
    quota = cgroup_cpu_quota()
    if quota:
        avail = min(avail, quota)
    cap = int(os.environ.get("METROPOLIS_BENCH_CPU_CORES", "0"))
    return max(1, min(avail, cap) if cap > 0 else avail)


def cpu_baseline(seconds):
    """Python restatement of the reference loop, one chain per process on the host cores (no GPU involved)."""
    cores = host_cores()
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_baseline", "--seconds", str(seconds),
                               "--seed", str(100 + i)], cwd=ROOT, env=env, stdout=subprocess.PIPE, text=True)
             for i in range(cores)]
    rate = 0.0
    for p in procs:
        out, _ = p.communicate()
        rec = json.loads(out.strip().splitlines()[-1])
        rate += rec["steps"] / rec["seconds"]
    return {"value": rate, "unit": "chain-steps/s", "cores": cores, "host_cpu_count": os.cpu_count(),
            "cgroup_cpu_quota": cgroup_cpu_quota(), "kind": "port",
            "sample": "oracle.reference_chain (pure-Python restatement of the reference step_all loop with "
                      "np.random.multivariate_normal), config 2 (16 real, E=sum x^2, T=1), one chain per process "
                      "on %d cores for %.0f s each; steps/s summed" % (cores, seconds)}


def cpu_baseline_c(seconds):
    """The plain-C restatement (oracle/c/me_oracle.c, float64, OpenMP over chains) on config 2: the strong CPU
    baseline next to the Python port.  Same Philox streams and arithmetic as the float64 GPU kernels."""
    from oracle.c_oracle import COracle
    cores = host_cores()
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
    return {"value": n * done / dt, "unit": "chain-steps/s", "cores": cores, "host_cpu_count": os.cpu_count(),
            "cgroup_cpu_quota": cgroup_cpu_quota(), "kind": "port",
            "sample": "oracle/c/me_oracle.c (float64, OpenMP), config 2 with 2^16 chains x %d sweeps in %.1f s, "
                      "acceptance %.3f" % (done, dt, chains.accepted / chains.proposed)}


def other_configs(me, rank, world, device, chains_log2, reduce_max, backend, native_comm):
    """BASELINE.json configs 3-5 through their protocols at ANY rank count: every rank owns the same
    number of chains (weak scaling, chain_offset = rank x chains), times are the maximum over ranks, rates whole-job.
    config 4 = 64 real parameters, dense SPD form on the matrix cores, 2^19 chains PER RANK, identity shape and
    cov_mode="pooled" (one adapt_pooled_shape = one all-reduce, then timed steps), both dtypes; config 5 = user plugin +
    wall, 2^18 chains per rank, (10 x step_all + measure + pooled all-reduce) x 200 with the reduction inside every
    cycle and in its overlapped form (metropolisengine_amd/protocols.py).  ``backend`` names the all-reduce path
    (``"rccl-native"``: the engine's own RCCL communicator, ``native_comm(engine)`` creates it; ``None``:
    torch.distributed)."""
    import numpy as np
    from metropolisengine_amd import protocols
    out = {}
    n = 1 << chains_log2

    def guarded(name, fn):
        """A side measurement must not sink the headline line: an exception becomes ``{"error": ...}``.  On several ranks
        the outcome is agreed on (a MAX over the ranks' failure flags), so that every rank records the block the same
        way; failures here are deterministic across ranks (unsupported shape, out of memory), raised before the block's
        first collective."""
        only = os.environ.get("METROPOLIS_BENCH_ONLY")       # dev: run only the blocks whose name matches this regex
        if only:
            import re
            if not re.search(only, name):
                return
        failed, result = 0.0, None
        try:
            result = fn()
        except Exception as exc:
            failed, result = 1.0, {"error": repr(exc)}
        if reduce_max(failed) > 0.0 and failed == 0.0:
            result = {"error": "failed on another rank"}
        out[name] = result

    def protocol(engine, n_chains, steps_per_measure, cycles, warm_cycles):
        protocols.cycle_protocol(engine, warm_cycles, steps_per_measure, "none", fused=True)
        dt, _ = protocols.cycle_protocol(engine, cycles, steps_per_measure, "none")
        dt = reduce_max(dt)
        return {"chain_steps_per_s": float(n_chains) * world * steps_per_measure * cycles / dt,
                "acceptance_rate": engine.acceptance_rate(), "chains_per_gpu": n_chains,
                "protocol": "(%d x step_all + measure) x %d" % (steps_per_measure, cycles)}

    def cycle_form(engine, n_chains, steps_per_measure, cycles, warm_cycles):
        """The same protocol through me_cycle: the k sweeps and the measure of a cycle as ONE launch (k_cycle)."""
        for _ in range(warm_cycles):
            engine.cycle(steps_per_measure)
        engine.sync()
        fused0 = engine.fused_cycles()
        t0 = time.perf_counter()
        for _ in range(cycles):
            engine.cycle(steps_per_measure)
        engine.sync()
        dt = reduce_max(time.perf_counter() - t0)
        return {"chain_steps_per_s": float(n_chains) * world * steps_per_measure * cycles / dt,
                "acceptance_rate": engine.acceptance_rate(), "chains_per_gpu": n_chains,
                "launches_per_cycle": 1 if engine.fused_cycles() - fused0 == cycles else 2,
                "protocol": "cycle(%d) x %d  (= %d x step_all + measure per launch)" % (steps_per_measure, cycles, steps_per_measure)}

    a = b = (1.0, 2.0, 4.0, 8.0)
    for dtype in ("f32", "f64"):
        suffix = "" if dtype == "f32" else "_f64"
        make3 = lambda: me.MetropolisEngine(me.DiagQuadratic(a, b), None, [0.0] * 4, [0j] * 4, temp=1.0, n_chains=n,
                                            seed=2026, dtype=dtype, device=device, chain_offset=rank * n)
        guarded("config3" + suffix, lambda: protocol(make3(), n, 10, 100, 60))
        guarded("config3_cycle" + suffix, lambda: cycle_form(make3(), n, 10, 100, 60))
    m = np.random.default_rng(5).standard_normal((64, 64))
    amat = m @ m.T / 64 + np.identity(64)
    n4 = n // 2
    for dtype, kernel, state_bytes in (("f32", "k_step_dense64_bf16x3", 8 * 64 + 16), ("f64", "k_step_dense64_f64", 16 * 64 + 32)):
        suffix = "" if dtype == "f32" else "_f64"

        def identity_shape():
            e4 = me.MetropolisEngine(me.DenseQuadratic(amat), None, [0.0] * 64, None, temp=1.0, n_chains=n4, seed=2026,
                                     cov_mode="fixed", dtype=dtype, device=device, chain_offset=rank * n4)
            e4.time_steps(30, 1)
            ms = reduce_max(e4.time_steps(100, 1) / 100)
            return {"chain_steps_per_s": float(n4) * world / (ms * 1e-3), "ms_per_launch": ms, "chains_per_gpu": n4,
                    "state_GBps_per_gpu": state_bytes * n4 / (ms * 1e-3) / 1e9, "kernel": kernel, "dtype": dtype}

        def pooled_shape():
            e4 = me.MetropolisEngine(me.DenseQuadratic(amat), None, [0.0] * 64, None, temp=1.0, n_chains=n4, seed=2026,
                                     cov_mode="pooled", dtype=dtype, device=device, chain_offset=rank * n4)
            native_comm(e4)
            rec = protocols.config4_pooled(e4, n4, world, backend=backend, reduce_max=reduce_max)
            rec.update(kernel=kernel + "<shared factor>", dtype=dtype, allreduce_backend=backend or "torch.distributed")
            e4.close()
            return rec

        guarded("config4" + suffix, identity_shape)
        guarded("config4_pooled" + suffix, pooled_shape)
    if world == 1:
        # config 4's parameter space with the REFERENCE's semantics (every chain its own adaptive 64 x 64 shape, streamed
        # kernels): what BASELINE's "identity and pooled_shared" prescription avoids, timed so that its cost is on record
        def reference_shapes():
            e4 = me.MetropolisEngine(me.DenseQuadratic(amat), None, [0.0] * 64, None, temp=1.0, n_chains=n // 8, seed=2026,
                                     cov_mode="reference", dtype="f32", device=device)
            for _ in range(52):
                e4.step_all()
                e4.measure()
            rec = protocol(e4, n // 8, 10, 5, 0)
            rec["dtype"] = "f32"
            return rec
        guarded("config4_reference_shapes", reference_shapes)
    src = os.path.join(ROOT, "examples", "user_energy_cylinder.h")
    n5 = n // 4
    for dtype in ("f32", "f64"):
        def cylinder(use_backend=backend):
            e5 = me.MetropolisEngine(me.UserEnergy("cylinder", src, (1.0, 0.5, 1.0)), me.AbsReal0AtLeast(1.0), [0.1, 0.0],
                                     [0.05] * 7, temp=0.1, n_chains=n5, seed=2026, dtype=dtype, device=device,
                                     chain_offset=rank * n5)
            if use_backend == "rccl-native":
                native_comm(e5)
            rec = protocols.config5(e5, n5, world, cycles=200, backend=use_backend, reduce_max=reduce_max)
            rec.update(dtype=dtype, allreduce_backend=use_backend or "torch.distributed")
            e5.close()
            return rec
        guarded("config5" + ("" if dtype == "f32" else "_f64"), cylinder)
        guarded("config5_cycle" + ("" if dtype == "f32" else "_f64"), lambda: cycle_form(
            me.MetropolisEngine(me.UserEnergy("cylinder", src, (1.0, 0.5, 1.0)), me.AbsReal0AtLeast(1.0), [0.1, 0.0],
                                [0.05] * 7, temp=0.1, n_chains=n5, seed=2026, dtype=dtype, device=device,
                                chain_offset=rank * n5), n5, 10, 200, 60))
        if backend == "rccl-native" and dtype == "f32":
            guarded("config5_torch_allreduce", lambda: cylinder(None))     # the same cycle through torch.distributed
    return out


def roofline_block(dtype, chains_log2, kernel_ms, bound):
    """The roofline object of one k_step<dtype,16,0,EnergyIso,identity> measurement (one sweep per launch)."""
    per_step = BYTES_PER_CHAIN_STEP_F64 if dtype == "f64" else BYTES_PER_CHAIN_STEP
    algorithmic = per_step * (1 << chains_log2)
    achieved = algorithmic / (kernel_ms * 1e-3) / 1e9
    traffic, source = pmc_traffic(dtype, chains_log2, 1)
    word = "double" if dtype == "f64" else "float"
    return {"bound": bound, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic, "traffic_source": source, "kernel": "k_step<%s,16,0,EnergyIso,identity>" % word,
            "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": algorithmic,
            "state_bytes_resident": algorithmic // 2,
            "note": "%d B per chain-step (%s r/w of x[16], energy, width) x 2^%d chains / average launch duration from "
                    "HIP events on the engine's stream over the timed region; traffic = FETCH_SIZE x2 + WRITE_SIZE of "
                    "the committed rocprofv3 passes named in traffic_source (not measured in this run)"
                    % (per_step, "fp64" if dtype == "f64" else "fp32", chains_log2)}


def bound_label(state_bytes):
    return "hbm+infinity-cache" if state_bytes <= INFINITY_CACHE_BYTES else "hbm"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--warmup", type=int, default=1000)
    ap.add_argument("--chains-log2", type=int, default=20, help="chains per GPU = 2**this")
    ap.add_argument("--dtype", default="f64", choices=("f64", "f32"), help="arithmetic of the headline run")
    ap.add_argument("--sweeps", type=int, default=1, help="sweeps fused per launch in the headline run")
    ap.add_argument("--fused-sweeps", type=int, default=32, help="extra fused-sweep measurement (0 = skip)")
    ap.add_argument("--hbm-chains-log2", type=int, default=22, help="chains of the cache-free roofline_hbm run (0 = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline sample length (0 = skip)")
    ap.add_argument("--extras", type=int, default=1, help="1: also time the float32 build of the headline kernel and "
                    "BASELINE configs 3-5 through their protocols (a few seconds; at every --gpus N)")
    ap.add_argument("--pool-backend", default="native", choices=("native", "torch"),
                    help="all-reduce of the pooled moments: the engine's own RCCL communicator, or torch.distributed")
    args = ap.parse_args()

    # stdout carries exactly ONE line, the JSON record: RCCL prints a version banner to stdout when a communicator is
    # created, and child libraries may print too -- everything before the record goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))

    # CPU baseline first, on rank 0 (the other ranks wait at the first barrier): child processes are started before
    # this process touches the GPU
    cpu = cpu_c = None
    if rank == 0 and args.cpu_seconds > 0:
        cpu = cpu_baseline(args.cpu_seconds)
        try:
            cpu_c = cpu_baseline_c(min(args.cpu_seconds, 6.0))
        except (OSError, subprocess.CalledProcessError) as exc:      # the optional C baseline must not sink the bench
            cpu_c = {"value": None, "unit": "chain-steps/s", "kind": "port", "sample": "unavailable: %s" % exc}

    import torch
    import torch.distributed as dist
    import metropolisengine_amd as me
    from metropolisengine_amd.distributed import init_native_comm, pooled_statistics

    # Rehearsal on a one-GPU box (never used by the driver): METROPOLIS_BENCH_REHEARSAL=1 puts every rank on GPU 0 and
    # uses gloo for the barrier / timing reduction / pooled moments, so the multi-rank code path can be exercised there.
    rehearsal = os.environ.get("METROPOLIS_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    distributed = "RANK" in os.environ           # launched by torch.distributed.run (also at --gpus 1)
    backend = None
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "gloo" if rehearsal else "nccl"
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    comm_device = "cpu" if (rehearsal or not distributed) else "cuda"

    n_local = 1 << args.chains_log2
    # rehearsal ranks share GPU 0: RCCL refuses two ranks on one device, the gloo group carries the moments there
    pool = {"backend": "rccl-native" if (args.pool_backend == "native" and not rehearsal) else None, "fallback": None}

    def native_comm(eng):
        """Give ``eng`` its RCCL communicator over all ranks (unique id broadcast through the process group)."""
        if pool["backend"] == "rccl-native":
            init_native_comm(eng, world_size=world)

    def make_engine(dtype, chains_log2=args.chains_log2, **extra):
        n = 1 << chains_log2
        if chains_log2 > 20:
            extra.setdefault("cov_mode", "fixed")   # no per-chain covariance fields (4.5 GB at 2^22 x 136 doubles)
        return me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * N_REAL, None, sampling_width=0.05,
                                   target_acceptance=0.3, temp=1.0, n_chains=n, seed=2026, dtype=dtype,
                                   device=local_rank, chain_offset=rank * n, **extra)

    engine = make_engine(args.dtype)

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(eng, n_launches, sweeps):
        """(wall seconds, device ms from HIP events on the engine's stream) of n_launches launches; max over ranks."""
        fence()
        t0 = time.perf_counter()
        dev_ms = eng.time_steps(n_launches, sweeps)        # enqueues, records events, waits for the stop event
        fence()
        wall = time.perf_counter() - t0
        if distributed:
            t = torch.tensor([wall, dev_ms], dtype=torch.float64, device=comm_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall, dev_ms = float(t[0]), float(t[1])
        return wall, dev_ms

    def side_run(eng, chains, per_step_bytes, sweeps, launches):
        """A side measurement with its own launch count: value, ms per launch, state GB/s."""
        eng.time_steps(max(20, launches // 5), sweeps)
        w, d = timed(eng, launches, sweeps)
        return {"value": float(chains) * world * launches * sweeps / w, "unit": "chain-steps/s", "launches": launches,
                "sweeps_per_launch": sweeps, "ms_per_launch": d / launches,
                "effective_state_GBps": per_step_bytes * chains / (d / launches * 1e-3) / 1e9}

    if args.warmup > 0:
        engine.time_steps(args.warmup, args.sweeps)
    wall, dev_ms = timed(engine, args.steps, args.sweeps)
    total_chain_steps = float(n_local) * world * args.steps * args.sweeps
    value = total_chain_steps / wall
    kernel_ms = dev_ms / args.steps                        # average launch duration (events, same timed region)
    per_step = BYTES_PER_CHAIN_STEP_F64 if args.dtype == "f64" else BYTES_PER_CHAIN_STEP
    roofline = roofline_block(args.dtype, args.chains_log2, kernel_ms, bound_label(per_step // 2 * n_local))

    # observability of the multi-GPU layout: how many ranks the collective backend really joined, and who owns what
    ranks_seen, offsets = 1, [rank * n_local]
    if distributed:
        one = torch.ones(1, dtype=torch.float64, device=comm_device)
        dist.all_reduce(one)
        ranks_seen = int(round(float(one[0])))
        gathered = [torch.zeros(1, dtype=torch.int64, device=comm_device) for _ in range(world)]
        dist.all_gather(gathered, torch.tensor([rank * n_local], dtype=torch.int64, device=comm_device))
        offsets = [int(t[0]) for t in gathered]

    fused = None
    if args.fused_sweeps > 0:
        launches = max(40, args.steps // args.fused_sweeps)
        fused = side_run(engine, n_local, per_step, args.fused_sweeps, launches)
        fused["dtype"] = args.dtype
        if args.chains_log2 == 20 and world == 1:
            # float64 vector instructions take at least 4 cycles like the float32 ones the helper assumes; most take more
            fused["valu_issue"] = issue_utilisation(torch, local_rank, fused["value"], "_f64" if args.dtype == "f64" else "")

    # the one collective: the all-reduce of the pooled moments -- through the engine's own RCCL communicator
    # (me_comm_init_rank + me_pooled_moments_allreduce, no PyTorch in the data path) unless --pool-backend torch
    # The first communicator decides the backend for the whole run, collectively: if it cannot be created on ANY rank
    # (librccl missing, an id that does not arrive) every rank falls back to torch.distributed, and the line says so.
    if pool["backend"] == "rccl-native":
        ok, why = 1.0, None
        try:
            native_comm(engine)
        except Exception as exc:
            ok, why = 0.0, repr(exc)
        if distributed:
            flag = torch.tensor([ok], dtype=torch.float64, device=comm_device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = float(flag[0])
        if ok < 1.0:
            pool["backend"], pool["fallback"] = None, why or "another rank could not create its RCCL communicator"
            try:
                engine.comm_destroy()
            except Exception:
                pass
    pool_backend = pool["backend"]
    stats = pooled_statistics(engine, backend=pool_backend)
    rccl_version = engine.comm_info()[2] if pool_backend == "rccl-native" else None
    engine.sync()
    engine.close()
    del engine

    # the cache-free figure: same kernel, state beyond the 256 MiB Infinity Cache (identity shape, no covariance fields)
    roofline_hbm = None
    if args.hbm_chains_log2 > 0 and args.sweeps == 1:
        big = make_engine(args.dtype, args.hbm_chains_log2, cov_mode="fixed")
        run = side_run(big, 1 << args.hbm_chains_log2, per_step, 1, SIDE_LAUNCHES)
        roofline_hbm = roofline_block(args.dtype, args.hbm_chains_log2, run["ms_per_launch"],
                                      bound_label(per_step // 2 << args.hbm_chains_log2))
        roofline_hbm["value"] = run["value"]
        roofline_hbm["launches"] = run["launches"]
        del big

    # the float32 build of the same kernels (the production dtype): same workload, same protocol
    f32 = None
    if args.extras and args.dtype == "f64":
        e32 = make_engine("f32")
        run = side_run(e32, n_local, BYTES_PER_CHAIN_STEP, 1, max(args.steps, SIDE_LAUNCHES))
        f32 = {"value": run["value"], "unit": "chain-steps/s", "ms_per_step": run["ms_per_launch"], "dtype": "f32",
               "launches": run["launches"],
               "roofline": roofline_block("f32", args.chains_log2, run["ms_per_launch"],
                                          bound_label(BYTES_PER_CHAIN_STEP // 2 * n_local))}
        if args.fused_sweeps > 0:
            f32["fused"] = side_run(e32, n_local, BYTES_PER_CHAIN_STEP, args.fused_sweeps, 40)
            if args.chains_log2 == 20 and world == 1:
                f32["fused"]["valu_issue"] = issue_utilisation(
                    torch, local_rank, f32["fused"]["value"])
        if args.hbm_chains_log2 > 0:
            del e32
            e32 = make_engine("f32", args.hbm_chains_log2, cov_mode="fixed")
            run = side_run(e32, 1 << args.hbm_chains_log2, BYTES_PER_CHAIN_STEP, 1, SIDE_LAUNCHES)
            f32["roofline_hbm"] = roofline_block("f32", args.hbm_chains_log2, run["ms_per_launch"],
                                                 bound_label(BYTES_PER_CHAIN_STEP // 2 << args.hbm_chains_log2))
        del e32

    extras = None
    if args.extras:
        def reduce_max(value):
            if not distributed:
                return value
            t = torch.tensor([value], dtype=torch.float64, device=comm_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t[0])
        extras = other_configs(me, rank, world, local_rank, args.chains_log2, reduce_max, pool_backend, native_comm)

    if rank == 0:
        line = {
            "metric": "MC steps/sec (chains x sweeps) at 2^20 chains, 16 params",
            "value": value, "unit": "chain-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "config 2: 16 real params, isotropic quadratic E=sum x^2, T=1, 2^%d chains per GPU, "
                                   "identity proposal shape, %d sweep(s) per launch, Philox4x32-10 streams, %s state "
                                   "and arithmetic%s"
                                   % (args.chains_log2, args.sweeps, "float64" if args.dtype == "f64" else "float32",
                                      " (the reference's dtype)" if args.dtype == "f64" else ""),
                       "chains_per_gpu": n_local, "global_chains": n_local * world, "sweeps_per_launch": args.sweeps,
                       "parallelism": "chains sharded over %d GPU(s), no data-path collective" % world},
            "roofline": roofline,
            "roofline_hbm": roofline_hbm,
            "cpu_baseline": cpu,
            "cpu_baseline_c": cpu_c,
            "f32": f32,
            "fused": fused,
            "other_configs": extras,
            "multi_gpu": {"backend": backend, "ranks_seen_by_allreduce": ranks_seen, "chain_offsets": offsets,
                          "pooled_chains": stats["n_chains"],
                          "pooled_moments_backend": pool_backend or "torch.distributed", "rccl_version": rccl_version,
                          "native_comm_fallback": pool["fallback"],
                          "ranks_seen_by_pooled_allreduce": int(round(stats["n_chains"] / float(n_local)))},
            "acceptance_rate": stats["acceptance_rate"],
            "pooled_variance_mean": float(sum(stats["covariance"][i][i] for i in range(N_REAL)) / N_REAL),
        }
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
