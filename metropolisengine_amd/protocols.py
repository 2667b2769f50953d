"""The measurement protocols of BASELINE.json's configurations 4 and 5, written once and driven by
``bench.py`` (any ``--gpus N``), ``tools/run_configs.py`` and the world-size-2 ``gloo`` test.

They are the reference's driver loop (``README.md:41-44``: k x ``step_all()`` then ``measure()``) over the many-chain
engine, plus the one collective of the multi-GPU layer:

* config 5 -- ``(10 x step_all + measure + pooled all-reduce) x cycles``: the pooled moments of ALL ranks' chains every
  cycle, either collected inside the cycle (``overlap=False``) or one cycle later (``overlap=True``: ``_begin`` after
  the measure, ``_end`` after the NEXT cycle's launches are queued, so copy, all-reduce and host arithmetic hide behind
  the sampler);
* config 4 -- warm-up with the identity shape, ONE ``adapt_pooled_shape`` (= one all-reduce: pooled covariance ->
  Cholesky factor -> ``me_set_shared_factor``), then the timed steps with the shared shape.

Everything goes through the engine's public methods, so any object with that surface can be driven (the CPU test uses
an oracle-backed stand-in).  ``backend`` is handed to :mod:`metropolisengine_amd.distributed`: ``None`` =
``torch.distributed`` (RCCL via ``nccl``, or ``gloo``), ``"rccl-native"`` = the engine's own communicator.
"""
import time

from . import distributed


def cycle_protocol(engine, cycles, steps_per_measure=10, pooled="none", backend=None, group=None, fused=False, on_stats=None,
                   one_launch=False):
    """Run ``cycles`` x (``steps_per_measure`` x ``step_all`` + ``measure`` [+ pooled statistics]) and return the wall
    seconds of the loop (the engine is synchronised before and after).

    ``pooled``: ``"none"``, ``"sync"`` (collect inside the cycle) or ``"overlap"`` (collect one cycle later).
    ``fused``: one ``step_all(steps_per_measure)`` launch instead of ``steps_per_measure`` launches.
    ``one_launch``: the sweeps AND the measure of a cycle as one launch (``engine.cycle``, ``me_cycle``).
    ``on_stats(cycle, stats)`` receives every cycle's pooled statistics (in the overlapped form: of the cycle before).
    """
    if pooled not in ("none", "sync", "overlap"):
        raise ValueError("pooled must be 'none', 'sync' or 'overlap'")
    engine.sync()
    pending = None
    last = None
    t0 = time.perf_counter()
    for cycle in range(cycles):
        if one_launch:
            engine.cycle(steps_per_measure)
        else:
            if fused:
                engine.step_all(steps_per_measure)
            else:
                for _ in range(steps_per_measure):
                    engine.step_all()
            engine.measure()
        if pooled == "sync":
            last = distributed.pooled_statistics(engine, group, backend=backend)
            if on_stats:
                on_stats(cycle, last)
        elif pooled == "overlap":
            if pending is not None:          # the reduction of the PREVIOUS cycle: its launches are long done
                last = distributed.pooled_statistics_end(engine, group, backend=backend)
                if on_stats:
                    on_stats(pending, last)
            distributed.pooled_statistics_begin(engine, backend=backend)
            pending = cycle
    if pending is not None:
        last = distributed.pooled_statistics_end(engine, group, backend=backend)
        if on_stats:
            on_stats(pending, last)
    engine.sync()
    return time.perf_counter() - t0, last


def config5(engine, n_local, world, cycles=200, steps_per_measure=10, warm_cycles=60, backend=None, group=None, reduce_max=None):
    """BASELINE config 5's protocol on one rank's engine: the cycle without the reduction, with it, and with it
    overlapped.  ``reduce_max(seconds)`` returns the maximum over ranks (identity on one rank).  Returns the block
    ``bench.py`` prints: whole-job chain-steps/s of each form, the ranks the all-reduce really summed over, and what the
    reduction costs per cycle."""
    reduce_max = reduce_max or (lambda v: v)
    cycle_protocol(engine, warm_cycles, steps_per_measure, "none", fused=True)
    distributed.pooled_statistics(engine, group, backend=backend)          # first call: imports, communicator warm-up
    out = {}
    seconds = {}
    stats = None
    for form in ("none", "sync", "overlap"):
        dt, last = cycle_protocol(engine, cycles, steps_per_measure, form, backend=backend, group=group)
        seconds[form] = reduce_max(dt)
        stats = last or stats
    total = float(n_local) * world * steps_per_measure * cycles
    if hasattr(engine, "cycle"):
        # the same protocol with each cycle's sweeps + measure as ONE launch (me_cycle), reduction overlapped
        dt, _ = cycle_protocol(engine, cycles, steps_per_measure, "overlap", backend=backend, group=group, one_launch=True)
        out["chain_steps_per_s_one_launch_cycles_overlapped"] = total / reduce_max(dt)
    out["chain_steps_per_s"] = total / seconds["sync"]
    out["chain_steps_per_s_overlapped"] = total / seconds["overlap"]
    out["chain_steps_per_s_without_allreduce"] = total / seconds["none"]
    out["allreduce_us_per_cycle"] = (seconds["sync"] - seconds["none"]) / cycles * 1e6
    out["allreduce_us_per_cycle_overlapped"] = (seconds["overlap"] - seconds["none"]) / cycles * 1e6
    out["ranks_seen_by_allreduce"] = int(round(stats["n_chains"] / float(n_local)))
    out["pooled_chains"] = stats["n_chains"]
    out["chains_per_gpu"] = n_local
    out["acceptance_rate"] = stats["acceptance_rate"]
    out["protocol"] = "(%d x step_all + measure + pooled all-reduce) x %d" % (steps_per_measure, cycles)
    return out


def config4_pooled(engine, n_local, world, warm_steps=200, steps=100, backend=None, group=None, reduce_max=None, jitter=1e-9):
    """BASELINE config 4 with ``cov_mode="pooled"``: warm up with the identity shape, one ``adapt_pooled_shape`` (one
    all-reduce over all ranks), then ``steps`` timed one-sweep launches with the shared shape.  Returns the bench block."""
    reduce_max = reduce_max or (lambda v: v)
    engine.step_all(warm_steps)
    engine.sync()
    t0 = time.perf_counter()
    stats = distributed.adapt_pooled_shape(engine, group, jitter=jitter, backend=backend)
    engine.sync()
    adapt_s = reduce_max(time.perf_counter() - t0)
    engine.time_steps(max(10, steps // 5), 1)
    engine.sync()
    t0 = time.perf_counter()
    dev_ms = engine.time_steps(steps, 1)
    engine.sync()
    dt = reduce_max(time.perf_counter() - t0)
    return {"chain_steps_per_s": float(n_local) * world * steps / dt, "ms_per_launch": reduce_max(dev_ms / steps),
            "adapt_pooled_shape_ms": adapt_s * 1e3, "ranks_seen_by_allreduce": int(round(stats["n_chains"] / float(n_local))),
            "pooled_chains": stats["n_chains"], "chains_per_gpu": n_local,
            "protocol": "%d warm-up sweeps (identity shape), one adapt_pooled_shape (one all-reduce), %d timed step_all"
                        % (warm_steps, steps)}
