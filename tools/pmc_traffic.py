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


XM_TAG = {"0": "", "1": ",img", "2": ",img+mask", "3": ",img+mask+skip"}


def canonical(name):
    """rocprofv3's demangled kernel name -> the key the ABI's kernel timer uses (cnn_amd_kernel_timing_report)"""
    name = name.replace(" ", "")
    m = re.match(r"igemm_dma_kernel<(\d+,\d+,\d+,\d+,\d+,\d+),(\d),(true|false)>$", name)
    if m:
        return f"igemm_dma_kernel<{m.group(1)}{XM_TAG[m.group(2)]}>"
    m = re.match(r"(igemm_kernel|wgrad_kernel)<(\d+,\d+,\d+,\d+,\d+,\d+),(true|false),(true|false)>$", name)
    if m:
        return f"{m.group(1)}<{m.group(2)}>"
    m = re.match(r"conv_fwd_pool_pk_3_16_3_2<\d+(?:,(true|false))?>$", name)
    if m:  # (second parameter: the packed one-byte pool mask)
        return "conv_fwd_pool_pk<3,16,3,2>" + ("+m8" if m.group(1) == "true" else "")
    m = re.match(r"conv_dgrad_pool_pk_3_16_3_2<\d+,(true|false)>$", name)
    if m:
        return "conv_dgrad_pk<3,16,3,2>" + ("+pool" if m.group(1) == "true" else "+poolm")
    m = re.match(r"conv_dgrad_pool_lds_3_16_3_2<(\d)>$", name)
    if m:  # (round 3: the LDS-staged kernel behind the same ABI entry point / timer key; mode 2 = packed pool mask)
        return "conv_dgrad_pk<3,16,3,2>" + {"0": "+poolm", "1": "+pool", "2": "+poolm8"}[m.group(1)]
    m = re.match(r"conv_wgrad_pk_3_16_3_2<\d+,(\d+)>$", name)
    if m:
        return "conv_wgrad_pk<3,16,3,2>" + {"0": "", "1": "+pool", "2": "+poolm"}[m.group(1)]
    m = re.match(r"conv_dgrad_rd_s2_kernel<(\d+),\d+,\d+,\d+(,(true|false))?>$", name)
    if m:
        return f"conv_dgrad_rd<2,{m.group(1)}>"
    m = re.match(r"conv_dgrad_m16_s2_kernel<(\d+),\d+,(true|false),(\d+)(?:,(?:true|false))?>$", name)
    if m:  # (CO per wave, NW, prepared, dy-channel split)
        return f"conv_dgrad_rd<2,{int(m.group(1)) * int(m.group(3))},m16>"
    m = re.match(r"conv_fwd_m16_kernel<(\d+),(\d+),\d+,(true|false),\d+>$", name)
    if m:
        return f"conv_fwd_rd<{m.group(1)},{m.group(2)},m16>"
    m = re.match(r"conv_fwd_rd_kernel<(\d+),(\d+),(\d+),\d+,\d+>$", name)
    if m:
        return f"conv_fwd_rd<{m.group(1)},{m.group(2)},{m.group(3)}>"
    m = re.match(r"wgrad_rd_kernel<(\d+,\d+,\d+),(true|false)>$", name)
    if m:
        return f"wgrad_rd<{m.group(1)}>" + ("+pool" if m.group(2) == "true" else "")
    m = re.match(r"conv_wgrad_win_kernel<(\d),(true|false)(?:,(?:true|false))?>$", name)
    if m:
        return "conv_wgrad_win<3,16,3,2>" + {"0": "", "1": "+pool", "2": "+poolm", "3": "+poolm8"}[m.group(1)]
    m = re.match(r"conv_wgrad_os_kernel<(\d+),(\d+),\d+,\d+,\d+>$", name)
    if m:
        return f"conv_wgrad_os<{m.group(1)},{m.group(2)}>"
    m = re.match(r"wgrad_rd_kernel_p1<(\d+,\d+,\d+),(true|false)>$", name)
    if m:
        return f"wgrad_rd<{m.group(1)},p1>"
    if name.startswith("conv_dgrad_thin_s1k3"):
        return "conv_dgrad_thin<3,s1>"
    m = re.match(r"(conv_(?:dgrad|fwd|wgrad)_pk2?)_3_16_3_2(<.*>)?$", name)
    if m:
        return f"{m.group(1)}<3,16,3,2>"
    m = re.match(r"conv_dgrad_pk_s2<(\d+),\d+,\d+>$", name)
    if m:
        return f"conv_dgrad_pk_s2<{m.group(1)}>"
    m = re.match(r"(conv_direct_(?:fwd|dgrad))<(\d+,\d+,\d+,\d+)(,\d+)?>$", name)
    if m:
        return f"{m.group(1)}<{m.group(2)}>"
    return name


def per_kernel(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name).split("(")[0]
        name = canonical(name)
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
