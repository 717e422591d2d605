#!/usr/bin/env python3
"""Turn the two rocprofv3 PMC passes of tools/profile_bench.sh (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md 'HBM' prescribes) into per-kernel HBM bytes per launch.
Corrections (calibrated on kernels with known byte counts in the same runs -- relu_fwd_vec / relu_bwd_vec /
maxpool_*): both counters are KiB; WRITE_SIZE is exact; FETCH_SIZE reports exactly 1/2 of the bytes read on gfx950,
so it is doubled.
usage: pmc_traffic.py <prof_dir> <out.json>"""
import collections
import csv
import json
import re
import sys

prof, out = sys.argv[1], sys.argv[2]


def per_kernel(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name).split("(")[0]
        name = re.sub(r",\s*(true|false)>$", ">", name)  # compile-time staging variant: same key as the ABI's timer name
        agg[(name, r["Grid_Size"])].append(float(r["Counter_Value"]))
    return agg


fetch = per_kernel(f"{prof}/pmc_fetch/fetch_counter_collection.csv")
write = per_kernel(f"{prof}/pmc_write/write_counter_collection.csv")
res = {}
for key in sorted(set(fetch) | set(write)):
    f = fetch.get(key, [0.0])
    w = write.get(key, [0.0])
    fb = 2.0 * 1024.0 * sum(f) / len(f)
    wb = 1024.0 * sum(w) / len(w)
    res[f"{key[0]}|grid={key[1]}"] = {"fetch_bytes": round(fb), "write_bytes": round(wb), "hbm_bytes": round(fb + wb),
                                      "launches_sampled": len(f)}
json.dump({"unit": "bytes per launch", "corrections": "FETCH_SIZE KiB x2 (gfx950 half-count), WRITE_SIZE KiB x1",
           "kernels": res}, open(out, "w"), indent=1)
print("wrote", out, len(res), "kernels")
