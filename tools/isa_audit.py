#!/usr/bin/env python3
"""Codegen audit (round 6): compile every csrc/*.hip to gfx950 assembly and list the kernels whose vector-memory loads are mostly followed by
their own `s_waitcnt vmcnt(0)` -- one memory round trip per load.  That is how hipcc compiles a load inside a runtime-bounded, unrolled
`if (j < n)` body (linear_bwd_fused with a runtime output count: 30 us for 4.7 MB) or a global load between stores (vmcnt counts stores).
Streaming kernels that wait once per loop iteration show up too: read the listing, not just the count.  No GPU needed.
usage: python tools/isa_audit.py [file.hip ...]        (default: every cnn_amd/csrc/*.hip; AUDIT_RATIO = waits per load that lists a kernel, 0.8)
Kernels that stage through LDS DMA or feed MFMA operands from memory wait by design (wgrad_sp, wgrad_rd, conv_rows): known entries."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cnn_amd", "csrc")
files = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
tmp = tempfile.mkdtemp(prefix="isa_audit_")
procs = []
for f in files:
    out = os.path.join(tmp, os.path.basename(f) + ".s")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
           "-S", "--cuda-device-only", f, "-o", out]
    procs.append((f, out, subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)))
found = 0
for f, out, p in procs:
    p.wait()
    if not os.path.exists(out):
        print(f"{os.path.basename(f)}: did not compile")
        continue
    text = open(out).read()
    for m in re.finditer(r"^(_Z[\w]+):\s*; @", text, re.M):
        body = text[m.end():]
        end = body.find("s_endpgm")
        if end < 0:
            continue
        lines = [l.strip() for l in body[:end].split("\n") if l.strip() and not l.strip().startswith(";")]
        loads = sum(1 for l in lines if re.match(r"(global_load|buffer_load)", l))
        waits = sum(1 for l in lines if re.match(r"s_waitcnt vmcnt\(0\)", l))
        spills = sum(1 for l in lines if l.startswith("scratch_"))
        if (loads >= 6 and waits >= loads * float(os.environ.get("AUDIT_RATIO", "0.8"))) or spills:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            print(f"{os.path.basename(f):24s} loads {loads:4d}  vmcnt(0) {waits:4d}  scratch {spills:3d}  {name[:150]}")
            found += 1
print(f"{found} kernel(s) listed")
