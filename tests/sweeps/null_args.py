#!/usr/bin/env python3
"""Error behaviour of the C ABI (include/cnn_amd.h): every entry point that returns an int status is called with all pointers NULL, once with
every size zero and once with every integer 4 / every float 1 (so that "nothing to do" is no excuse) -- in a child process each, so that a
dereference or a division by zero shows up as a signal, not as a dead test session.  Expected: no crash either way, and a non-zero status
(CNN_AMD_E_BADARG ...) with a message in cnn_amd_last_error() for the second call.  Runs without a GPU (argument checks come before any launch).
usage: null_args.py            -> one line per entry point that crashed or accepted the call, exit code = their number
       null_args.py <name>     -> (child) call that one entry point"""
import ctypes as C
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cnn_amd import capi

# calls for which "nothing to do" is a legitimate success, or that take no argument that could be wrong
FINE_WITH_ZEROS = {"cnn_amd_kernel_timing_enable", "cnn_amd_kernel_timing_sampling", "cnn_amd_timing_span_end", "cnn_amd_set_option", "cnn_amd_flush_reduces",
                   "cnn_amd_side_stream_join", "cnn_stream_synchronize", "cnn_amd_publish_next_kernel", "cnn_amd_wait_published", "cnn_device_free",
                   "cnn_comm_available", "cnn_amd_measure_build", "cnn_amd_abi_version", "cnn_amd_published_is_last", "cnn_amd_timing_span_begin",
                   "cnn_conv2d_tune_export", "cnn_host_free_pinned", "cnn_comm_destroy", "cnn_batch_stager_destroy", "cnn_event_destroy", "cnn_stream_destroy"}


NONZERO = False


# (answers, not statuses: "is this geometry supported", "how large is the output")
PREDICATES = {n for n in capi.SIGNATURES if n.endswith("_supported") or n.endswith("_out_dim") or n.endswith("_available")}


def zero_for(t):
    if t in (C.c_float, C.c_double):
        return t(1.0 if NONZERO else 0.0)
    if t in (C.c_void_p, C.c_char_p) or hasattr(t, "contents") or (hasattr(t, "_type_") and isinstance(getattr(t, "_type_", None), type)):
        return None
    return t(4 if NONZERO else 0)


if len(sys.argv) > 1:
    name = sys.argv[1]
    lib = capi.load()
    res, args = capi.SIGNATURES[name]
    getattr(lib, name)(*[zero_for(t) for t in args])  # (zeros: any status, no crash)
    NONZERO = True
    rc = getattr(lib, name)(*[zero_for(t) for t in args])
    print("rc", rc, "msg", lib.cnn_amd_last_error()[:120])
    sys.exit(0 if (rc != 0 or res is not C.c_int or name in PREDICATES) else 3)

from concurrent.futures import ThreadPoolExecutor

names = [n for n, (res, args) in sorted(capi.SIGNATURES.items()) if res in (C.c_int, C.c_size_t, C.c_longlong) and args and n not in FINE_WITH_ZEROS]


def child(name):
    return name, subprocess.run([sys.executable, os.path.abspath(__file__), name], capture_output=True, text=True, timeout=300)


bad = 0
with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 4)) as pool:
    for name, r in pool.map(child, names):
        if r.returncode == 0:
            continue
        bad += 1
        what = "ACCEPTED the call (status 0)" if r.returncode == 3 else f"CRASHED / failed to run (exit {r.returncode})"
        out = (r.stdout + r.stderr).strip()
        print(f"{name}: {what}  {out.splitlines()[-1][:160] if out else ''}")
print(f"NULL ARGS {'OK' if bad == 0 else 'FAILED'}: {len(names)} entry points called, {bad} crashed or accepted NULL pointers")
sys.exit(bad)
