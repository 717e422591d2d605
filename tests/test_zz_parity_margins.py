"""Runs LAST (file name): bounds what the tolerance checks of THIS session let pass, and hides nothing.

north_star asks for 1e-4 relative on activations and gradients (tests/util.py: tensor-normalised).  Two kinds of check may exceed a plain
comparison with the fp32 oracle, and both must stay the exception:
  * the fp64-ARBITRATED form (assert_close_arbitrated) exists because deep-net gradients of the fp32 ORACLE itself drift from the exact
    result.  In rounds 4 and 5 no check of the builder's sessions needed that branch; the ceiling below leaves head room
    for a box whose tuner pins other implicit-GEMM tiles (a different summation order moves a deep gradient by ~1e-5) instead of
    failing the whole session on the first such tensor.
  * a DERIVED bound (assert_close_derived_bound): one 3-element tensor, the bias gradient of the stacks' linear layer, is held to
    0.5 * 1e-4 * max|logit| / max|gradient| -- the logits' own 1e-4 pushed through the softmax Jacobian.  Every such
    record is listed here and in the session summary; none is filtered out of the reported worst error.
The full table of the session goes to gpurun_out/parity_margins.json (tests/conftest.py); the round's copy is under profiles/.
"""
import pytest

from tests import util

MAX_FP64_BRANCH_PASSES = 2      # measured: 0 (profiles/r04 .. r06); ADVICE r5: close to the measured value -- two tensors of head room for a box
                                # whose tuner pins other implicit-GEMM tiles, not six
MAX_PLAIN_ERR = 1.0e-4          # every check at north_star's tolerance, whatever its branch
MAX_DERIVED_BOUND_CHECKS = 12   # 2 steps x 4 stack cases + the 2 own-backward tests, at most; each is a linear.b record


@pytest.mark.gpu
def test_arbitrated_tolerance_stays_the_exception():
    recs = [r for r in util.MARGINS if r["test"].split("::")[0].endswith(("test_gpu_parity.py", "test_gpu_stacks.py", "test_host_mirror.py",
                                                                          "test_gpu_batchnorm.py"))]
    if len(recs) < 50:
        pytest.skip("needs the whole -m gpu session (the parity tests run before this file)")
    arb = [r for r in recs if r["branch"] == "fp64-arbitrated"]
    assert len(arb) <= MAX_FP64_BRANCH_PASSES, [(r["test"], r["what"], r["plain_err_vs_fp32_oracle"]) for r in arb]
    # everything checked at north_star's 1e-4 (plain branch and the exact-zero noise bound alike) is within it
    worst = max((r for r in recs if r["tol"] <= 1e-4 and r["branch"] != "fp64-arbitrated"), key=lambda r: r["plain_err_vs_fp32_oracle"])
    assert worst["plain_err_vs_fp32_oracle"] <= MAX_PLAIN_ERR, worst
    # every check held to MORE than 1e-4: only the derived-bound branch may do that, only on the linear layer's bias gradient, and the
    # list is printed (pytest -rA / the summary file) -- count and worst, not a filtered maximum
    above = [r for r in recs if r["tol"] > 1e-4]
    print(f"checks with a tolerance above 1e-4: {len(above)}; worst error among them "
          f"{max([r['plain_err_vs_fp32_oracle'] for r in above] or [0.0]):.3e}")
    for r in above:
        print(f"   {r['what']}: err {r['plain_err_vs_fp32_oracle']:.3e} <= derived bound {r['tol']:.3e}")
        assert r["branch"] == "derived-bound" and "linear.b" in r["what"], r
        assert r["plain_err_vs_fp32_oracle"] <= r["tol"] <= 1e-3, r  # (the derived bound is capped: tests/test_gpu_stacks.py DERIVED_BOUND_CAP)
    assert len(above) <= MAX_DERIVED_BOUND_CHECKS, len(above)
