"""World-size-2 gloo test of the multi-GPU layer on CPU: chain sharding and the pooled-moment all-reduce.

The per-rank moment vectors come from the oracle (two shards addressed by global chain id); the product code under
test is metropolisengine_amd.distributed (shard_chains, allreduce_moments, moments_to_statistics), which is the same
code the nccl (RCCL) path runs with CUDA tensors."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from metropolisengine_amd.distributed import allreduce_moments, moments_to_statistics, shard_chains
    from oracle import energies
    from oracle.manychain import ManyChainOracle
    offset, count = shard_chains(n_total, rank, world)
    shard = ManyChainOracle(2, 1, energies.landau_toy(), count, seed=11, temp=0.1, initial_real_params=[0.0, 0.0],
                            initial_complex_params=[0j], chain_offset=offset)
    for _ in range(20):
        shard.step(3)
        shard.measure()
    local = shard.pooled_moments()
    as_numpy = allreduce_moments(local.copy())                      # numpy path
    as_tensor = allreduce_moments(torch.from_numpy(local.copy()))   # tensor path (what nccl uses on the GPU)
    assert np.array_equal(as_numpy, as_tensor.numpy())
    stats = moments_to_statistics(as_numpy, 2, 1)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), total=as_numpy, x=shard.x, offset=offset, count=count,
             mean=stats["mean"], cov=stats["covariance"], acc=stats["acceptance_rate"])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_pooled_moments(tmp_path):
    from metropolisengine_amd.distributed import moments_to_statistics
    from oracle import energies
    from oracle.manychain import ManyChainOracle
    n_total, world = 37, 2                       # ragged split: 19 + 18
    mp.spawn(_worker, args=(world, _free_port(), n_total, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert (int(r0["offset"]), int(r0["count"]), int(r1["offset"]), int(r1["count"])) == (0, 19, 19, 18)
    assert np.array_equal(r0["total"], r1["total"])
    full = ManyChainOracle(2, 1, energies.landau_toy(), n_total, seed=11, temp=0.1, initial_real_params=[0.0, 0.0],
                           initial_complex_params=[0j])
    for _ in range(20):
        full.step(3)
        full.measure()
    # sharded chains are bit-identical to the single-engine run; pooled sums agree to fp64 summation order
    assert np.array_equal(full.x, np.concatenate((r0["x"], r1["x"])))
    assert np.allclose(r0["total"], full.pooled_moments(), rtol=1e-13, atol=1e-13)
    stats = moments_to_statistics(full.pooled_moments(), 2, 1)
    assert np.allclose(stats["mean"], r0["mean"]) and np.allclose(stats["covariance"], r0["cov"])
    assert np.allclose(stats["covariance"], np.cov(full.x.T, bias=True), atol=1e-12)
    assert stats["n_chains"] == n_total


def test_shard_chains_partitions_exactly():
    from metropolisengine_amd.distributed import shard_chains
    for n_total in (1, 7, 8, 1 << 20, (1 << 20) + 3):
        for world in (1, 2, 3, 8):
            spans = [shard_chains(n_total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n_total
            for (o0, c0), (o1, _) in zip(spans, spans[1:]):
                assert o0 + c0 == o1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    with pytest.raises(ValueError):
        shard_chains(8, 2, 2)


def test_pooled_factor_matches_per_chain_convention():
    from metropolisengine_amd.distributed import pooled_factor
    rng = np.random.default_rng(0)
    nr, nc = 2, 2
    a = rng.standard_normal((200, nr))
    z = (rng.standard_normal((200, nc)) + 1j * rng.standard_normal((200, nc))) @ np.array([[1.0, 0.3j], [0, 0.7]])
    x = np.concatenate((a, z.real, z.imag), axis=1)
    cov = np.cov(x.T, bias=True)
    packed = pooled_factor(cov, nr, nc)
    assert packed.shape == (nr * (nr + 1) // 2 + nc * nc,)
    lr = np.zeros((nr, nr))
    lr[np.tril_indices(nr)] = packed[:3]
    assert np.allclose(lr @ lr.T, cov[:nr, :nr])
    zc = z - z.mean(axis=0)
    k = (zc.T @ zc.conj()) / 200                       # E[z z^H]
    lc = np.array([[packed[3], 0], [packed[4] + 1j * packed[5], packed[6]]])
    assert np.allclose(lc @ lc.conj().T, np.conj(k))   # proposals use conj(K) (metropolis_engine.py:292-298)


class _OracleEngine:
    """Stand-in for MetropolisEngine on a GPU-less host: the attribute / method surface metropolisengine_amd.distributed
    drives (pooled moments, their split begin/end form, the shared proposal factor), backed by an oracle shard."""

    def __init__(self, shard, nr, nc, device=0):
        self.shard, self.num_real_params, self.num_complex_params, self.device = shard, nr, nc, device
        self.shared_factor, self._pending = None, None

    def pooled_moments(self):
        return self.shard.pooled_moments()

    def pooled_moments_begin(self):
        assert self._pending is None, "one reduction in flight per engine"
        self._pending = self.shard.pooled_moments()      # the state at the time of _begin

    def pooled_moments_end(self):
        out, self._pending = self._pending, None
        return out

    def set_shared_factor(self, packed):
        self.shared_factor = np.array(packed, dtype=np.float64)

    # the stepping surface metropolisengine_amd.protocols drives (the bench's config 4 / config 5 loops)
    def step_all(self, n_sweeps=1):
        self.shard.step(n_sweeps)

    def measure(self):
        self.shard.measure()

    def sync(self):
        pass

    def acceptance_rate(self):
        return self.shard.accepted / self.shard.proposed

    def time_steps(self, n_launches, n_sweeps=1):
        import time
        t0 = time.perf_counter()
        for _ in range(n_launches):
            self.shard.step(n_sweeps)
        return (time.perf_counter() - t0) * 1e3


def _worker_surface(rank, world, port, n_total, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from metropolisengine_amd.distributed import (adapt_pooled_shape, pooled_statistics, pooled_statistics_begin,
                                                  pooled_statistics_end, shard_chains)
    from oracle import energies
    from oracle.manychain import ManyChainOracle
    offset, count = shard_chains(n_total, rank, world)
    a = (1.0, 2.0)
    shard = ManyChainOracle(2, 2, energies.diag_quadratic(2, 2, a, a), count, seed=5, temp=1.0, initial_real_params=[0.0, 0.0],
                            initial_complex_params=[0j, 0j], chain_offset=offset, sampling_width=0.4)
    eng = _OracleEngine(shard, 2, 2)
    shard.step(60)
    direct = pooled_statistics(eng)
    pooled_statistics_begin(eng)
    shard.step(5)                                   # work enqueued between _begin and _end does not change the result
    split = pooled_statistics_end(eng)
    assert np.array_equal(direct["covariance"], split["covariance"]) and direct["n_chains"] == split["n_chains"] == n_total
    stats = adapt_pooled_shape(eng, jitter=1e-9)     # pool over ranks -> factor -> install
    np.savez(os.path.join(out_dir, "surface%d.npz" % rank), factor=eng.shared_factor, cov=stats["covariance"],
             n=stats["n_chains"], x=shard.x)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_pooled_statistics_surface(tmp_path):
    """pooled_statistics, its begin/end form and adapt_pooled_shape at world size 2: every rank installs the SAME factor,
    the Cholesky factor of the covariance pooled over both ranks' chains."""
    from metropolisengine_amd.distributed import pooled_factor
    n_total, world = 301, 2
    mp.spawn(_worker_surface, args=(world, _free_port(), n_total, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "surface0.npz"), np.load(tmp_path / "surface1.npz")
    assert int(r0["n"]) == int(r1["n"]) == n_total
    assert np.array_equal(r0["factor"], r1["factor"]) and np.array_equal(r0["cov"], r1["cov"])
    x = np.concatenate((r0["x"], r1["x"]))
    cov = np.cov(x.T, bias=True)
    assert np.allclose(r0["cov"], cov, atol=1e-12)
    assert np.allclose(r0["factor"], pooled_factor(cov, 2, 2, jitter=1e-9), atol=1e-9)


def test_rccl_path_refuses_a_device_mismatch(monkeypatch):
    """The nccl branch of pooled_statistics allocates the all-reduce buffer on the ENGINE's GPU and refuses to run when
    torch's current device is another one (a cross-device write would fault instead of raising)."""
    from metropolisengine_amd import distributed
    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_backend", lambda group=None: "nccl")
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 1)

    class _Engine:
        num_real_params, num_complex_params, device = 2, 0, 0

        def pooled_moments_into(self, ptr, n):
            raise AssertionError("must not be reached")
    with pytest.raises(RuntimeError, match="current device"):
        distributed.pooled_statistics(_Engine())


def _worker_protocols(rank, world, port, n_local, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from metropolisengine_amd import protocols
    from oracle import energies
    from oracle.manychain import ManyChainOracle

    def reduce_max(value):
        t = torch.tensor([value], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])

    # config 5's protocol (2 real + 7 complex, cylinder surrogate + wall), n_local chains per rank, weak scaling
    shard = ManyChainOracle(2, 7, energies.cylinder_surrogate(2, 7, 1.0, 0.5, 1.0), n_local, seed=2026, temp=0.1,
                            initial_real_params=[0.1, 0.0], initial_complex_params=[0.05] * 7, chain_offset=rank * n_local,
                            reject=energies.wall_reject(1.0))
    eng = _OracleEngine(shard, 2, 7)
    seen = []
    dt, last = protocols.cycle_protocol(eng, 3, 2, "sync", on_stats=lambda c, st: seen.append((c, st["n_chains"])))
    assert [c for c, _ in seen] == [0, 1, 2] and all(n == world * n_local for _, n in seen)
    seen_overlap = []
    protocols.cycle_protocol(eng, 3, 2, "overlap", on_stats=lambda c, st: seen_overlap.append((c, st["n_chains"])))
    assert [c for c, _ in seen_overlap] == [0, 1, 2] and all(n == world * n_local for _, n in seen_overlap)
    rec5 = protocols.config5(eng, n_local, world, cycles=3, steps_per_measure=2, warm_cycles=1, reduce_max=reduce_max)
    # config 4's pooled protocol (small stand-in: 3 real parameters, dense form)
    amat = np.array([[2.0, 0.3, 0.0], [0.3, 1.0, 0.2], [0.0, 0.2, 0.5]])
    shard4 = ManyChainOracle(3, 0, energies.dense_quadratic(3, 0, amat), n_local, seed=2026, temp=1.0,
                             initial_real_params=[0.0] * 3, chain_offset=rank * n_local, sampling_width=0.5)
    eng4 = _OracleEngine(shard4, 3, 0)
    rec4 = protocols.config4_pooled(eng4, n_local, world, warm_steps=40, steps=5, reduce_max=reduce_max)
    np.savez(os.path.join(out_dir, "protocols%d.npz" % rank), factor=eng4.shared_factor, x4=shard4.x,
             ranks5=rec5["ranks_seen_by_allreduce"], pooled5=rec5["pooled_chains"], rate5=rec5["chain_steps_per_s"],
             us5=rec5["allreduce_us_per_cycle"], ranks4=rec4["ranks_seen_by_allreduce"], pooled4=rec4["pooled_chains"],
             acc5=rec5["acceptance_rate"])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_bench_protocols(tmp_path):
    """The protocol functions bench.py --gpus N runs for BASELINE configs 4 and 5 (metropolisengine_amd/protocols.py) at
    world size 2 over gloo: the per-cycle all-reduce sees both ranks, the overlapped form collects every cycle exactly
    once, and config 4's adapt_pooled_shape installs the same factor of the covariance pooled over BOTH shards."""
    n_local, world = 24, 2
    mp.spawn(_worker_protocols, args=(world, _free_port(), n_local, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "protocols0.npz"), np.load(tmp_path / "protocols1.npz")
    for r in (r0, r1):
        assert int(r["ranks5"]) == world and int(r["pooled5"]) == world * n_local
        assert int(r["ranks4"]) == world and int(r["pooled4"]) == world * n_local
        assert float(r["rate5"]) > 0 and 0.0 < float(r["acc5"]) < 1.0
    assert np.array_equal(r0["factor"], r1["factor"])
    assert float(r0["rate5"]) == float(r1["rate5"])            # max-over-ranks timing: every rank reports the same rate
