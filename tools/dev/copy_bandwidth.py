"""Dev probe (GPU box): what a plain device-to-device copy achieves at the headline working-set sizes -- the practical
ceiling beside the 8 TB/s datasheet peak that bench.py's roofline.frac is quoted against.
    python tools/dev/copy_bandwidth.py
"""
import torch

for log2_chains in (20, 21, 22, 24):
    n = (1 << log2_chains) * 18            # config 2: 16 params + energy + width per chain, fp32
    src = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
    dst = torch.empty_like(src)
    for _ in range(20):
        dst.copy_(src)
    torch.cuda.synchronize()
    reps = 200
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        dst.copy_(src)
    t1.record()
    torch.cuda.synchronize()
    us = t0.elapsed_time(t1) * 1e3 / reps
    print("2^%d chains: read+write %.1f MB in %.2f us = %.2f TB/s" % (log2_chains, 8 * n / 1e6, us, 8 * n / us / 1e6), flush=True)
