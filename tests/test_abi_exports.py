"""CPU-side checks of the drop-in boundary: libcnn_amd.so loads and exports every symbol include/cnn_amd.h
declares, and the pure host helpers agree with the reference's shape rules.  No device calls."""
import os
import re

import pytest

from cnn_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "cnn_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(cnn_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    lib = capi.load()
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/cnn_amd.h but not exported"
    # and the binding table covers exactly the header
    assert sorted(capi.SIGNATURES) == declared


def test_host_helpers_follow_reference_shape_rules():
    lib = capi.load()
    assert lib.cnn_amd_abi_version() == 2  # (2: cnn_conv2d_desc.flags)
    # conv2d.cpp:41-42 (pad = 0): 224 -> 111 -> (pool) 55 -> 27 -> 13 -> 6
    assert [lib.cnn_conv2d_out_dim(h, 3, 2, 0) for h in (224, 55, 27, 13)] == [111, 27, 13, 6]
    assert lib.cnn_conv2d_out_dim(112, 3, 1, 0) == 110 and lib.cnn_conv2d_out_dim(112, 3, 1, 1) == 112
    assert lib.cnn_maxpool2d_out_dim(111, 2, 2) == 55  # pool2d.cpp:14-15 drops the last row/col


def test_workspace_query_and_bad_args_do_not_need_a_gpu():
    lib = capi.load()
    import ctypes as C

    d = capi.ConvDesc(256, 64, 112, 112, 128, 3, 1, 0)
    assert lib.cnn_conv2d_workspace_bytes(C.byref(d)) > 128 * 64 * 9 * 4
    bad = capi.ConvDesc(1, 3, 2, 2, 4, 3, 1, 0)  # kernel larger than the image
    assert lib.cnn_conv2d_workspace_bytes(C.byref(bad)) == 0
    assert lib.cnn_relu_forward(None, None, 16, None) == 1  # CNN_AMD_E_BADARG, message set
    assert b"null" in lib.cnn_amd_last_error()
    assert isinstance(lib.cnn_amd_device_arch(), bytes)


def test_option_table_needs_no_gpu_and_no_getenv_on_launch_paths():
    """the measurement switches live in one table (cnn_amd_set_option / cnn_amd_get_option): set, read back, remove; names with
    or without the CNN_AMD_ prefix; and no kernel source calls getenv any more (the table reads the environment once)"""
    import glob

    assert capi.get_option("NO_SUCH_SWITCH") is None
    capi.set_option("CNN_AMD_TEST_SWITCH", 7)
    assert capi.get_option("TEST_SWITCH") == "7" and capi.get_option("CNN_AMD_TEST_SWITCH") == "7"
    with capi.option("TEST_SWITCH", "9"):
        assert capi.get_option("TEST_SWITCH") == "9"
    assert capi.get_option("TEST_SWITCH") == "7"
    capi.set_option("TEST_SWITCH", None)
    assert capi.get_option("TEST_SWITCH") is None
    assert capi.load().cnn_amd_set_option(b"", b"1") != 0
    for path in glob.glob(os.path.join(ROOT, "cnn_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "cnn_amd", "host", "src", "*.cpp")):
        text = re.sub(r"//.*", "", open(path).read())
        assert "getenv" not in text, path


def test_product_library_rejects_every_result_changing_switch():
    """the ablations that change results (a kernel without its stores / DMAs / MFMAs), the cycle printers, the LDS
    request override and the exchange-less data-parallel step exist only in the measurement build (-DCNN_AMD_MEASURE,
    libcnn_amd_measure.so).  In the product library: cnn_amd_set_option refuses the names, the environment cannot arm them (checked in a
    fresh process), no launch path queries them through the option table, and the profiling array of the window kernel is not linked in"""
    import glob
    import subprocess
    import sys

    lib = capi.load()
    assert lib.cnn_amd_measure_build() == 0
    names = ["DBG", "ROWS_DBG", "ROWS_LDS", "WIN_DBG", "OS_DBG", "STEM_DBG", "FWD_RD_DBG", "DGRAD_RD_DBG", "RD_DBG", "SP_DBG", "S2_DBG", "DP_SKIP_EXCHANGE"]
    for name in names:
        for spelled in (name, "CNN_AMD_" + name):
            assert lib.cnn_amd_set_option(spelled.encode(), b"1") == 1, spelled  # CNN_AMD_E_BADARG
            assert b"measurement-only" in lib.cnn_amd_last_error()
        assert capi.get_option(name) is None
        assert lib.cnn_amd_set_option(name.encode(), None) == 0  # (removing is harmless)
    # every *_DBG / wrong-result name the sources know is on the rejected list, and none of them is read through the live option table
    for path in glob.glob(os.path.join(ROOT, "cnn_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "cnn_amd", "csrc", "*.h")):
        text = re.sub(r"//.*", "", open(path).read())
        for m in re.finditer(r'CNN_MEASURE_INT\("([A-Z0-9_]+)"', text):
            assert m.group(1) in names, (path, m.group(1))
        for m in re.finditer(r'CNN_OPT(?:_INT|_VAL|_SET)?\("([A-Z0-9_]+)"', text):
            assert m.group(1) not in names and not m.group(1).endswith("DBG"), (path, m.group(1))
    # a fresh process with the variables exported: the table does not take them from the environment
    env = dict(os.environ, CNN_AMD_ROWS_DBG="1", CNN_AMD_DBG="8", CNN_AMD_IGEMM_CFG="215")
    code = ("from cnn_amd import capi; print(capi.get_option('ROWS_DBG'), capi.get_option('DBG'), capi.get_option('IGEMM_CFG'))")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, check=True).stdout.split()
    assert out == ["None", "None", "215"], out
    syms = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "g_prof" not in syms


def test_size_queries_are_memoised_per_option_generation():
    """cnn_conv2d_prepared_bytes / cnn_conv2d_workspace_bytes plan every implicit-GEMM tile candidate; the launch paths call them on
    every launch, so the result is memoised per (desc, option-table generation): a switch that changes the plan must still show"""
    import ctypes as C

    lib = capi.load()
    d = capi.ConvDesc(64, 64, 56, 56, 64, 3, 1, 1)
    sizes = lambda: (int(lib.cnn_conv2d_workspace_bytes(C.byref(d))), int(lib.cnn_conv2d_prepared_bytes(C.byref(d))))
    base = sizes()
    assert base == sizes() and base[0] > 0 and base[1] > 0
    with capi.option("IGEMM_CFG", "215"):  # one forced tile instead of the largest over all tuner candidates
        forced = sizes()
        assert forced[1] < base[1]
    assert sizes() == base
    other = capi.ConvDesc(64, 64, 56, 56, 64, 3, 1, 1, capi.POOL_MASK_PACKED)  # (flags are part of the key; this query ignores them)
    assert int(lib.cnn_conv2d_prepared_bytes(C.byref(other))) == base[1]


def test_packed_pool_mask_host_helpers_need_no_gpu():
    """cnn_conv2d_desc.flags (ABI version 2): size and support queries of the packed pool mask, unknown flag bits are rejected"""
    import ctypes as C

    lib = capi.load()
    d = capi.ConvDesc(4, 3, 224, 224, 16, 3, 2, 0)  # the reference's first layer: 111x111 -> 55x55 pooled, rows of 56 bytes
    assert lib.cnn_conv2d_pool_mask_packed_supported(C.byref(d)) == 1
    assert lib.cnn_conv2d_pool_mask_bytes(C.byref(d)) == 4 * 16 * 55 * 55 * 4
    d.flags = capi.POOL_MASK_PACKED
    assert lib.cnn_conv2d_pool_mask_bytes(C.byref(d)) == 4 * 16 * 55 * 56 + 64
    d.flags = 2
    assert lib.cnn_conv2d_pool_mask_bytes(C.byref(d)) == 0 and b"unknown desc flags" in lib.cnn_amd_last_error()
    assert lib.cnn_conv2d_workspace_bytes(C.byref(d)) == 0
    other = capi.ConvDesc(4, 16, 55, 55, 32, 3, 2, 0)  # not the fused block's geometry
    assert lib.cnn_conv2d_pool_mask_packed_supported(C.byref(other)) == 0
    d.flags = 0
    with capi.option("POOL_MASK_PACKED", "0"):  # the A/B switch turns the packed form off for everybody who asks
        assert lib.cnn_conv2d_pool_mask_packed_supported(C.byref(d)) == 0
    assert lib.cnn_conv2d_pool_mask_packed_supported(C.byref(d)) == 1


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", "/nonexistent/libcnn_amd.so")
    with pytest.raises(capi.CnnAmdError):
        capi.load()


def test_every_entry_point_refuses_null_pointers_without_crashing():
    """tests/sweeps/null_args.py: each status-returning entry point of include/cnn_amd.h in a child process, all pointers NULL -- once with
    every size zero (any status, no signal), once with non-zero sizes (a non-zero status and a message).  Found on its first run:
    cnn_conv2d_out_dim / cnn_maxpool2d_out_dim raised SIGFPE on a stride of 0"""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "sweeps", "null_args.py")], cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "NULL ARGS OK" in r.stdout, (r.stdout + r.stderr)[-2000:]
