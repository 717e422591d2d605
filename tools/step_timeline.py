#!/usr/bin/env python3
"""Timeline of one steady-state train step from a rocprofv3 --kernel-trace CSV: start offset, duration and queue of every
kernel between two consecutive launches of the step's LAST kernel (first_layer_finish in the fused step tail, else sgd_vec), plus
how long the GPU was busy / idle.
usage: step_timeline.py <bench_kernel_trace.csv> [step_index_from_end=3] [delimiter kernel prefix]"""
import csv
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"^void ", "", name).split("(")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
delim = sys.argv[3] if len(sys.argv) > 3 else ("first_layer_finish" if any(r[2].startswith("first_layer_finish") for r in rows) else "sgd_vec")
if delim.startswith("start:"):  # delimit by the FIRST kernel of a step instead (e.g. start:conv_fwd_pool_pk)
    delim = delim[6:]
    ends = [i - 1 for i, r in enumerate(rows) if r[2].startswith(delim) and i > 0]
else:
    ends = [i for i, r in enumerate(rows) if r[2].startswith(delim)]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
lo, hi = ends[-k - 1] + 1, ends[-k] + 1
step = rows[lo:hi]
t0 = rows[ends[-k - 1]][1]
print(f"step of {len(step)} kernels, {(step[-1][1] - t0) / 1e3:.1f} us from the end of the previous {delim} to the end of this one")
busy_until, idle = t0, 0
for s, e, name, q, st in step:
    if s > busy_until:
        idle += s - busy_until
    busy_until = max(busy_until, e)
    print(f"{(s - t0) / 1e3:8.1f} us  +{(e - s) / 1e3:7.1f} us  q{q:>2} s{st:>2}  {name[:90]}")
print(f"GPU idle (no kernel resident) {idle / 1e3:.1f} us")
