"""Timed CPU baseline -- TEST/BENCH INFRASTRUCTURE ONLY (used by bench.py's ``cpu_baseline`` leg).

Runs the single-chain restatement of the reference loop (``oracle.reference_chain.ReferenceChain`` with the
reference's own random sources: ``np.random.multivariate_normal`` + ``random.uniform``, i.e. the reference's cost
profile -- SVD inside numpy's mvn, Python control flow) on BASELINE config 2 (16 real parameters, isotropic
quadratic, T = 1, step_all only) for a fixed wall time and prints the number of chain-steps done.

    python -m oracle.cpu_baseline --seconds 10 --seed 1        # one worker = one chain = one core
"""
import argparse
import json
import random
import time

import numpy as np


def run(seconds, seed, n_real=16):
    from oracle.reference_chain import ReferenceChain
    np.random.seed(seed)
    random.seed(seed)
    chain = ReferenceChain(lambda r, c: float(np.dot(r, r)), initial_real_params=[0.0] * n_real, temp=1.0)
    steps = 0
    t0 = time.perf_counter()
    deadline = t0 + seconds
    while True:
        for _ in range(200):
            chain.step_all()
        steps += 200
        now = time.perf_counter()
        if now >= deadline:
            break
    return {"steps": steps, "seconds": now - t0, "accepted": chain.accepted}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--n-real", type=int, default=16)
    args = ap.parse_args()
    print(json.dumps(run(args.seconds, args.seed, args.n_real)))
