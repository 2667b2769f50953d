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
