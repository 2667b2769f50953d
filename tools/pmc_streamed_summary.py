"""Summarise tools/profile_streamed.sh (kernel-trace stats + FETCH_SIZE / WRITE_SIZE passes of tools/pmc_workloads_streamed.py)
into one JSON: per kernel the last 12 dispatches -- the per-chain shapes are in use by then -- with the average duration
of the stats pass, the corrected traffic (FETCH_SIZE x 2 as for every other kernel, tools/pmc_summary.py) and the
algorithmic bytes of tools/pmc_workloads_streamed.py's table.

    python tools/pmc_streamed_summary.py gpurun_out/r02_prof_streamed > profiles/r02_streamed_kernels.json
"""
import csv
import glob
import json
import statistics
import sys

base = sys.argv[1]
LAST = 12
N = 1 << 17
ALGORITHMIC = {"k_factor_tile<float": 16640, "k_factor_tile<double": 33280,
               "k_step<float, 64, 0, me::EnergyDense<float, 64, 0>, 3": 8848,
               "k_step<double, 64, 0, me::EnergyDense<double, 64, 0>, 3": 17696,
               "k_measure<float, 64, 0, true": 17924, "k_measure<double, 64, 0, true": 35848}


def per_dispatch(sub, counter):
    sums = {}
    for path in glob.glob("%s/%s/**/*counter_collection.csv" % (base, sub), recursive=True):
        for row in csv.DictReader(open(path, newline="")):
            if row["Counter_Name"] == counter:
                key = (row["Kernel_Name"], int(row["Dispatch_Id"]))
                sums[key] = sums.get(key, 0.0) + float(row["Counter_Value"])
    return sums


def last(sums, name):
    return [v for (k, _), v in sorted(sums.items(), key=lambda kv: kv[0][1]) if k == name][-LAST:]


fetch, write = per_dispatch("pmc_FETCH_SIZE", "FETCH_SIZE"), per_dispatch("pmc_WRITE_SIZE", "WRITE_SIZE")
durations = {}
for path in glob.glob("%s/stats/**/*kernel_trace.csv" % base, recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        durations.setdefault(row["Kernel_Name"], []).append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
out = {"workload": "tools/pmc_workloads_streamed.py: 64 real parameters, dense quadratic form, cov_mode=reference, 2^17 chains; "
                   "last %d dispatches of each kernel" % LAST,
       "peak_GBps": 8000.0, "kernels": {}}
for needle, per_chain in ALGORITHMIC.items():
    for name in sorted(set(k for k, _ in fetch)):
        if needle not in name:
            continue
        f, w, d = last(fetch, name), last(write, name), durations.get(name, [])[-LAST:]
        if not f or not w or not d:
            continue
        fb, wb = 2.0 * statistics.median(f) * 1024.0, statistics.median(w) * 1024.0
        us = statistics.mean(d) / 1e3
        out["kernels"][name.split("(")[0].replace("void me::", "")] = {
            "launches_sampled": len(f), "avg_duration_us": us, "fetch_bytes_corrected": fb, "write_bytes": wb,
            "algorithmic_bytes_per_launch": per_chain * N, "traffic_over_algorithmic": (fb + wb) / (per_chain * N),
            "algorithmic_GBps": per_chain * N / us / 1e3, "frac_of_peak": per_chain * N / us / 1e3 / 8000.0}
print(json.dumps(out, indent=1))
