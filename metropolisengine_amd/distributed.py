"""Multi-GPU layer: chains shard with no data-path exchange; one small all-reduce pools the ensemble moments.

One process per GPU (``torch.distributed``, backend ``nccl`` = RCCL over xGMI on ROCm; ``gloo`` in the CPU tests).
Chains are the independent units: rank ``r`` of ``W`` owns the global chain ids ``[offset_r, offset_r + count_r)``
and the Philox streams are addressed by global id, so results do not depend on ``W``.  The only collective is a
sum all-reduce of the fp64 moment vector (1.5 KB at D=16, 18 KB at D=64): latency-bound, issued once per
``pooled_statistics`` call and never per step.  The reference has no counterpart (it runs one chain in one
process); this is the "RCCL all-reduce only for the pooled covariance/observables" of BASELINE.json.
"""
import numpy as np


def shard_chains(n_total, rank, world_size):
    """Contiguous block partition of ``n_total`` global chain ids: returns ``(offset, count)`` for ``rank``."""
    if not 0 <= rank < world_size:
        raise ValueError("rank out of range")
    base, extra = divmod(int(n_total), int(world_size))
    count = base + (1 if rank < extra else 0)
    offset = rank * base + min(rank, extra)
    return offset, count


def moments_size(n_real, n_complex):
    d = n_real + 2 * n_complex
    return 1 + d + d * (d + 1) // 2 + 2 * n_real + n_complex + 2


def allreduce_moments(moments, group=None):
    """Sum a moment vector over all ranks, in place.

    ``moments`` is a ``torch.Tensor`` (float64; CUDA for RCCL, CPU for gloo) or a numpy array (reduced through a CPU
    tensor).  With no initialised process group this is the identity (single-GPU runs).
    """
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return moments
    if isinstance(moments, np.ndarray):
        t = torch.from_numpy(moments)
        if dist.get_backend(group) == "nccl":
            t = t.cuda()
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            moments[...] = t.cpu().numpy()
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return moments
    dist.all_reduce(moments, op=dist.ReduceOp.SUM, group=group)
    return moments


_SYMMETRIC_INDEX = {}      # d -> [d, d] index into the packed lower triangle (built once per size: this runs every cycle)


def _symmetric_index(d):
    idx = _SYMMETRIC_INDEX.get(d)
    if idx is None:
        idx = np.zeros((d, d), dtype=np.intp)
        il = np.tril_indices(d)
        idx[il] = np.arange(il[0].size)
        idx = np.maximum(idx, idx.T)
        _SYMMETRIC_INDEX[d] = idx
    return idx


def moments_to_statistics(moments, n_real, n_complex):
    """Convert raw summed moments ``(n, sum x, sum x x^T, sum obs, accepted, proposed)`` to ensemble statistics."""
    m = np.asarray(moments, dtype=np.float64)
    d = n_real + 2 * n_complex
    n_pair = d * (d + 1) // 2
    n_obs = 2 * n_real + n_complex
    if m.shape[0] != moments_size(n_real, n_complex):
        raise ValueError("moment vector has the wrong length")
    n = m[0]
    mean = m[1:1 + d] / n
    second = m[1 + d:1 + d + n_pair][_symmetric_index(d)]       # the full symmetric matrix from its packed lower triangle
    cov = second / n - np.outer(mean, mean)
    obs = m[1 + d + n_pair:1 + d + n_pair + n_obs] / n
    accepted, proposed = m[-2], m[-1]
    return {"n_chains": int(round(n)), "mean": mean, "covariance": cov, "observables_mean": obs,
            "acceptance_rate": accepted / proposed if proposed else float("nan")}


def init_native_comm(engine, rank=None, world_size=None, group=None, id_file=None):
    """Give ``engine`` its own RCCL communicator (``me_comm_init_rank``; backend ``"rccl-native"``): rank 0 draws the
    unique id (``me_comm_unique_id``) and hands it to the other ranks -- through the initialised ``torch.distributed``
    process group (any backend: a 128-byte broadcast, the only thing PyTorch is used for), or, with ``id_file``, through a
    file on a file system all ranks see (rank 0 writes ``id_file`` atomically, the others poll for it; no PyTorch at
    all).  Collective over all ranks; afterwards ``pooled_statistics(engine, backend="rccl-native")`` and the
    ``_begin`` / ``_end`` pair run the all-reduce on the engine's own HIP streams."""
    import os
    import time
    if id_file is not None:
        if rank is None or world_size is None:
            raise ValueError("id_file needs explicit rank and world_size")
        if rank == 0:
            uid = engine.comm_unique_id()
            tmp = "%s.tmp.%d" % (id_file, os.getpid())
            with open(tmp, "wb") as fh:
                fh.write(uid)
            os.replace(tmp, id_file)
        else:
            deadline = time.time() + 300.0
            while not os.path.exists(id_file):
                if time.time() > deadline:
                    raise TimeoutError("no unique id appeared at %s" % id_file)
                time.sleep(0.01)
            with open(id_file, "rb") as fh:
                uid = fh.read()
    else:
        import torch
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            if (world_size or 1) != 1:
                raise RuntimeError("init_native_comm: no process group to broadcast the unique id through (pass id_file)")
            rank, world_size, uid = 0, 1, engine.comm_unique_id()
        else:
            rank, world_size = dist.get_rank(group), dist.get_world_size(group)
            uid = engine.comm_unique_id() if rank == 0 else bytes(128)
            dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
            t = torch.tensor(list(uid), dtype=torch.uint8, device=dev)
            dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            uid = bytes(t.cpu().tolist())
    engine.comm_init(uid, rank, world_size)
    return rank, world_size


def pooled_statistics(engine, group=None, backend=None):
    """Ensemble mean / covariance / observables over ALL ranks' chains of ``engine`` (one all-reduce).

    ``backend="rccl-native"``: the engine's own communicator (:func:`init_native_comm`); default: ``torch.distributed``
    (RCCL through the ``nccl`` backend, or ``gloo`` on CPU), the identity without a process group."""
    nr, nc = engine.num_real_params, engine.num_complex_params
    size = moments_size(nr, nc)
    if backend == "rccl-native":
        return moments_to_statistics(engine.pooled_moments_allreduce(), nr, nc)
    import torch
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_backend(group) == "nccl":
        # The buffer must live on the ENGINE's GPU (the engine's own stream writes it) and that must be the device the
        # process group communicates from: a mismatch would be a cross-device write, not an error message.
        dev = torch.device("cuda", engine.device)
        if torch.cuda.current_device() != engine.device:
            raise RuntimeError("pooled_statistics: torch's current device is cuda:%d but the engine lives on cuda:%d; "
                               "call torch.cuda.set_device(engine.device) (one process per GPU)"
                               % (torch.cuda.current_device(), engine.device))
        buf = torch.empty(size, dtype=torch.float64, device=dev)
        # the caching allocator may hand back a block an earlier torch kernel is still using on torch's stream; the
        # engine writes from its own stream, so order the two explicitly (me_pooled_moments_device synchronises the
        # engine's stream before it returns, which orders the all-reduce behind the write)
        torch.cuda.current_stream(dev).synchronize()
        engine.pooled_moments_into(buf.data_ptr(), size)          # k_pool_reduce straight into the RCCL buffer
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        total = buf.cpu().numpy()
    else:
        total = allreduce_moments(engine.pooled_moments(), group)
    return moments_to_statistics(total, nr, nc)


def pooled_statistics_begin(engine, backend=None):
    """Start the pooled reduction of the engine's current state without stalling its stream (see
    ``me_pooled_moments_begin``); keep enqueuing steps, then call :func:`pooled_statistics_end` with the same backend.
    With ``backend="rccl-native"`` the all-reduce itself is enqueued too (``me_pooled_moments_allreduce_begin``): nothing
    between ``_begin`` and ``_end`` touches the host."""
    if backend == "rccl-native":
        engine.pooled_moments_allreduce_begin()
    else:
        engine.pooled_moments_begin()


def pooled_statistics_end(engine, group=None, backend=None):
    """Collect the reduction started by :func:`pooled_statistics_begin`, all-reduce it over the ranks (already done on
    the device with ``backend="rccl-native"``) and convert it.  The GPU keeps executing whatever was enqueued in
    between, so the copy, the all-reduce and the host arithmetic are off the sampler's critical path."""
    total = engine.pooled_moments_end()
    if backend != "rccl-native":
        total = allreduce_moments(total, group)
    return moments_to_statistics(total, engine.num_real_params, engine.num_complex_params)


def pooled_factor(covariance, n_real, n_complex, jitter=0.0):
    """Packed proposal factor (ME_FIELD_FACTOR layout) from a pooled real-representation covariance [D, D].

    The real block is ``chol(C_rr)``; the complex block is built from the circular part of the pooled covariance,
    ``K = (C_aa + C_bb) + i (C_ba - C_ab)``, and factored as ``chol(conj(K))`` like the per-chain path
    (metropolis_engine.py:292-298).
    """
    nr, nc = n_real, n_complex
    c = np.asarray(covariance, dtype=np.float64)
    packed = []
    if nr:
        lr = np.linalg.cholesky(c[:nr, :nr] + jitter * np.identity(nr))
        packed.extend(lr[np.tril_indices(nr)])
    if nc:
        a = slice(nr, nr + nc)
        b = slice(nr + nc, nr + 2 * nc)
        k = (c[a, a] + c[b, b]) + 1j * (c[b, a] - c[a, b])
        lc = np.linalg.cholesky(np.conj(k) + jitter * np.identity(nc))
        for i in range(nc):
            for j in range(i):
                packed.extend((lc[i, j].real, lc[i, j].imag))
            packed.append(lc[i, i].real)
    return np.asarray(packed, dtype=np.float64)


def adapt_pooled_shape(engine, group=None, jitter=0.0, backend=None):
    """The many-chain counterpart of the reference's per-chain covariance adaptation (metropolis_engine.py:416-427 feeding
    :261-302): pool the ensemble covariance over all ranks (one all-reduce), factor it and install it as the proposal
    shape every chain shares (``cov_mode="pooled"``).  Returns the pooled statistics."""
    stats = pooled_statistics(engine, group, backend=backend)
    engine.set_shared_factor(pooled_factor(stats["covariance"], engine.num_real_params, engine.num_complex_params,
                                           jitter=jitter))
    return stats
