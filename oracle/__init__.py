"""CPU oracle for the many-chain Metropolis hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / the timed CPU baseline.  The
product (``metropolisengine_amd``) never imports this package and has no CPU
fallback.

Contents
--------
``reference_chain``  literal single-chain restatement of the reference
                     algorithm (``metropolisengine/metropolis_engine.py:17-463``),
                     pinned against the imported reference through the golden
                     fixtures in ``tests/golden/`` (made by ``make_golden.py``).
``philox``           Philox4x32-10 + Box-Muller stream definition shared (as a
                     specification) with the HIP kernels; pinned by the
                     Random123 known-answer vectors.
``manychain``        numpy many-chain restatement: the exact semantics the HIP
                     kernels implement (counter-based streams, Cholesky
                     proposals), validated against ``reference_chain`` on
                     identical streams.
``c/``               the same many-chain semantics in plain C (fast checker and
                     strong CPU baseline).
"""
