"""Count the vector-ALU instructions one sweep of the headline kernel issues (static, from the compiler's assembly):
the figure behind bench.py's fused-sweep "issue utilisation" (when sweeps are fused the kernel is
instruction-bound and the HBM fraction no longer describes it).

    python tools/valu_count.py > profiles/rNN_kernel_valu.json

The sweep loop is the innermost loop of k_step<float,16,0,EnergyIso,identity>; every `v_*` instruction in it is
counted once (MFMA excluded, none here), `s_*` separately.  A wave64 instruction occupies a SIMD's 16 lanes for 4
cycles, so one wavefront-sweep needs >= 4 x count cycles of one SIMD.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DTYPE = "d" if "--f64" in sys.argv else "f"        # --f64: the reference-precision instantiation
KERNEL = "_ZN2me6k_stepI%sLi16ELi0ENS_9EnergyIsoI%sLi16ELi0EEELi0ELb0ELi0ELb0EEEvNS_8StepArgsIT_EET2_" % (DTYPE, DTYPE)

with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "k16.s")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"),
                    "-DME_NR=16", "-DME_NC=0", "-DME_DENSE=1", "-DME_PER_CHAIN=1", "-S", "--cuda-device-only",
                    os.path.join(ROOT, "metropolisengine_amd", "csrc", "me_kernels.hip"), "-o", out],
                   check=True, stderr=subprocess.DEVNULL)
    text = open(out).read()
start = text.index(KERNEL + ":")
body = text[start:text.index("s_endpgm", start)].splitlines()
# the sweep loop: the inner loop with the largest body.  The assembly annotates every basic block with the loop it belongs to
# ("=>This Inner Loop Header" on the header, "in Loop: Header=BBn_m" on the others; the latch need not branch to the header
# label -- it may fall through), so the body is the union of the blocks carrying one header's name.
blocks, cur = [], None
for line in body:
    m = re.match(r"(\.LBB\d+_\d+):|; %bb\.(\d+):", line)
    if m:
        cur = {"label": (m.group(1) or "").lstrip("."), "note": line, "ops": []}
        blocks.append(cur)
        continue
    if cur is None:
        continue
    text = line.strip()
    if text.startswith(";"):
        cur["note"] += " " + text
    elif text and not text.startswith("."):
        cur["ops"].append(text.split()[0])
best = None
for b in blocks:
    if "Inner Loop Header" not in b["note"] or not b["label"]:
        continue
    members = [b] + [o for o in blocks if o is not b and re.search(r"Header=" + re.escape(b["label"][1:]) + r"\b", o["note"])]
    ops = [op for blk in members for op in blk["ops"]]
    if best is None or len(ops) > len(best):
        best = ops
if best is None:
    sys.exit("no inner loop found")
loop = best
valu = [op for op in loop if op.startswith("v_")]
salu = [op for op in loop if op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop", "s_cbranch"))]
mem = [op for op in loop if op.startswith(("buffer_", "global_", "ds_", "scratch_", "flat_"))]
top = {}
for op in valu:
    top[op] = top.get(op, 0) + 1
print(json.dumps({"kernel": "k_step<%s,16,0,EnergyIso,identity>" % ("double" if DTYPE == "d" else "float"), "scope": "one sweep (innermost loop body)",
                  "valu_instructions": len(valu), "salu_instructions": len(salu), "memory_instructions": len(mem),
                  "cycles_per_wavefront_sweep_lower_bound": 4 * len(valu),
                  "most_frequent": sorted(top.items(), key=lambda kv: -kv[1])[:16 if DTYPE == "d" else 8]}, indent=1))
