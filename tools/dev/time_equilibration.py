"""Dev timing (GPU box): batched equilibration detection against the host restatement."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from metropolisengine_amd import statistics
rng = np.random.default_rng(1)
for n, length in ((64, 500), (1024, 2000)):
    a = np.cumsum(rng.standard_normal((n, length)), axis=1) * 0.05 + rng.standard_normal((n, length))
    statistics.detect_equilibration_batch(a[:2])
    t0 = time.perf_counter(); statistics.detect_equilibration_batch(a); dev = time.perf_counter() - t0
    t0 = time.perf_counter(); statistics.detect_equilibration(a[0]); host = (time.perf_counter() - t0) * n
    print("%d series x %d samples: device %.3f s, host %.1f s (one series timed, x%d)" % (n, length, dev, host, n), flush=True)
