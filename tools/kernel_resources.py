"""Dev tool: per-kernel register / scratch / occupancy table for one (n_real, n_complex) kernel set.
    python tools/kernel_resources.py 16 0 [extra hipcc flags]
"""
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
nr, nc = sys.argv[1], sys.argv[2]
extra = sys.argv[3:]
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(root, "include"),
       "-DME_NR=" + nr, "-DME_NC=" + nc, "-DME_DENSE=1", "-DME_PER_CHAIN=" + ("2" if int(nr) > 17 and int(nc) == 0 else ("0" if int(nr) > 32 else "1"))] + extra + \
      ["-c", os.path.join(root, "metropolisengine_amd/csrc/me_kernels.hip"), "-o", "/tmp/kres.o",
       "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: +Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
        continue
    for key, pat in (("sgpr", r"TotalSGPRs: (\d+)"), ("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"),
                     ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)"),
                     ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
print("%5s %5s %5s %8s %4s  %s" % ("SGPR", "VGPR", "AGPR", "scratch", "occ", "kernel"))
for r in rows:
    name = re.sub(r"\(me::.*", "", r["name"]).replace("void me::", "").replace("me::", "")
    print("%5d %5d %5d %8d %4d  %s" % (r.get("sgpr", -1), r.get("vgpr", -1), r.get("agpr", -1), r.get("scratch", -1),
                                        r.get("occ", -1), name[:110]))
