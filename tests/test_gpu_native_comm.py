"""RCCL behind the C ABI (me_comm_*, me_pooled_moments_allreduce*; include/metropolis_engine.h) on one GPU: a
single-rank communicator.  The all-reduce over one rank is the identity, so the all-reduced moments must equal
me_pooled_moments BIT FOR BIT -- which pins the plumbing (reduction kernels on the engine's stream, ncclAllReduce and
the copy on its second stream behind an event, the pinned staging buffer).  Ranks > 1 need one GPU per rank (RCCL
refuses two ranks on one device); the world-size-2 logic is covered over gloo in tests/test_distributed_gloo.py."""
import numpy as np
import pytest

import metropolisengine_amd as me
from metropolisengine_amd import distributed, protocols

pytestmark = pytest.mark.gpu


def _engine(dtype="f64", n=4096, **kw):
    a = b = (1.0, 2.0, 4.0, 8.0)
    return me.MetropolisEngine(me.DiagQuadratic(a, b), None, [0.1, 0.2, -0.1, 0.0], [0.1j, 0.2, -0.1 + 0.1j, 0.0], temp=1.0,
                               n_chains=n, seed=7, dtype=dtype, **kw)


def test_single_rank_allreduce_equals_local_moments_bitwise():
    eng = _engine()
    eng.step_all(20)
    eng.measure()
    assert eng.comm_info()[:2] == (-1, 0)
    with pytest.raises(me._capi.MetropolisLibraryError):          # ME_ERR_STATE: no communicator yet
        eng.pooled_moments_allreduce()
    uid = eng.comm_unique_id()
    assert len(uid) == 128
    eng.comm_init(uid, 0, 1)
    rank, world, version = eng.comm_info()
    assert (rank, world) == (0, 1) and version > 20000
    local = eng.pooled_moments()
    assert np.array_equal(eng.pooled_moments_allreduce(), local)
    # the split form: steps enqueued between _begin and _end do not change what _end hands out
    eng.pooled_moments_allreduce_begin()
    eng.step_all(5)
    assert np.array_equal(eng.pooled_moments_end(), local)
    assert not np.array_equal(eng.pooled_moments_allreduce(), local)      # ... and the state did move on
    with pytest.raises(me._capi.MetropolisLibraryError):          # one communicator per engine
        eng.comm_init(uid, 0, 1)
    eng.comm_destroy()
    assert eng.comm_info()[:2] == (-1, 0)
    eng.comm_destroy()                                            # idempotent
    eng.close()


def test_native_backend_through_the_distributed_layer(tmp_path):
    """init_native_comm (unique id through a file: no PyTorch involved), pooled_statistics / begin / end /
    adapt_pooled_shape with backend="rccl-native", and the bench's config-5 protocol on top."""
    eng = _engine("f32", n=8192, cov_mode="pooled")
    assert distributed.init_native_comm(eng, rank=0, world_size=1, id_file=str(tmp_path / "uid")) == (0, 1)
    eng.step_all(200)
    ref = distributed.moments_to_statistics(eng.pooled_moments(), 4, 4)
    got = distributed.pooled_statistics(eng, backend="rccl-native")
    assert got["n_chains"] == 8192 and np.array_equal(got["covariance"], ref["covariance"])
    distributed.pooled_statistics_begin(eng, backend="rccl-native")
    eng.step_all(3)
    split = distributed.pooled_statistics_end(eng, backend="rccl-native")
    assert np.array_equal(split["covariance"], ref["covariance"])
    stats = distributed.adapt_pooled_shape(eng, jitter=1e-9, backend="rccl-native")
    assert np.allclose(eng.shared_factor(), distributed.pooled_factor(stats["covariance"], 4, 4, jitter=1e-9))
    rec = protocols.config5(eng, 8192, 1, cycles=5, steps_per_measure=3, warm_cycles=2, backend="rccl-native")
    assert rec["ranks_seen_by_allreduce"] == 1 and rec["pooled_chains"] == 8192 and rec["chain_steps_per_s"] > 0
    eng.close()                                                    # me_destroy also destroys the communicator
