"""Summarise the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) of the headline
kernel into profiles/rNN_pmc_traffic.json, which bench.py reports as roofline.traffic.

    python tools/pmc_summary.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> [kernel substring]
                                [chains_log2] [algorithmic bytes per chain-step] ["bench.py flags of the passes"]
                                [last N dispatches only] > out.json

Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters
are in KiB; FETCH_SIZE under-reports by 2x on this part for the dword-per-lane buffer loads the kernel issues, which
the exactly known read volume of the kernel (72 B per chain) confirms.
"""
import csv
import glob
import json
import os
import statistics
import sys


def per_dispatch(directory, counter, needle):
    sums = {}
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter or needle not in row.get("Kernel_Name", ""):
                    continue
                key = (path, int(row.get("Dispatch_Id")))
                sums[key] = sums.get(key, 0.0) + float(row["Counter_Value"])
    ordered = [sums[k] for k in sorted(sums)]
    return ordered[-LAST_N:] if LAST_N else ordered


LAST_N = int(sys.argv[7]) if len(sys.argv) > 7 else 0      # e.g. only the measures after the 50-measure threshold
fetch_dir, write_dir = sys.argv[1], sys.argv[2]
needle = sys.argv[3] if len(sys.argv) > 3 else "k_step<float, 16, 0"
fetch = per_dispatch(fetch_dir, "FETCH_SIZE", needle)
write = per_dispatch(write_dir, "WRITE_SIZE", needle)
if not fetch or not write:
    sys.exit("no %s dispatches with FETCH_SIZE / WRITE_SIZE found" % needle)
f_kib, w_kib = statistics.median(fetch), statistics.median(write)
chains_log2 = int(sys.argv[4]) if len(sys.argv) > 4 else 20
per_chain = int(sys.argv[5]) if len(sys.argv) > 5 else 144
flags = sys.argv[6] if len(sys.argv) > 6 else "--gpus 1 --steps 50 --warmup 20 --cpu-seconds 0 --fused-sweeps 0 --extras 0"
chains = 1 << chains_log2
out = {
    "command": "rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv -- python3 %s "
               "(one pass per counter); tools/pmc_summary.py" % (flags if flags.startswith("tools/") else "bench.py " + flags),
    "kernel": needle + " ... at 2^%d chains, 1 sweep per launch" % chains_log2,
    "launches_sampled": min(len(fetch), len(write)),
    "FETCH_SIZE_raw_KiB_median": f_kib,
    "WRITE_SIZE_raw_KiB_median": w_kib,
    "fetch_bytes_corrected": 2.0 * f_kib * 1024.0,
    "write_bytes": w_kib * 1024.0,
    "traffic_bytes_per_launch": 2.0 * f_kib * 1024.0 + w_kib * 1024.0,
    "algorithmic_bytes_per_launch": per_chain * chains,
    "traffic_over_algorithmic": (2.0 * f_kib * 1024.0 + w_kib * 1024.0) / (per_chain * chains),
    "note": "FETCH_SIZE x2 per the gfx950 correction in MI355X_MICROARCH.md (HBM section); compare with the kernel's "
            "exactly-known read volume (%d B x 2^%d = %.1f MB). WRITE_SIZE includes the per-wavefront acceptance "
            "slots (8 B per 64 chains). Both counters include Infinity-Cache hits (same section)."
            % (per_chain // 2, chains_log2, per_chain // 2 * chains / 1e6),
}
print(json.dumps(out, indent=1))
