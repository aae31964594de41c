"""The sympy -> HIP generator (tools/gen_kernels.py, the device-side counterpart of the reference's
init_func.py:24-81) against the hand-optimised kernels the product runs: every generated function
(k, d2k/dxdx0, d2k/dydy0, d2k/dxdy0 and their lx- / ly-derivatives) of families A-D is evaluated on the
device as generated (libsympgpr_probe.so) and diffed against sgpr_kernel_eval_host."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NAMES = {0: "k", 1: "d2kdxdx0", 2: "d2kdydy0", 3: "d2kdxdy0"}


@pytest.mark.parametrize("fam", "ABCD")
def test_generated_functions_match_hand_written(fam):
    from sympgpr_amd import _lib as L
    from sympgpr_amd import ops
    probe = L.load_probe_library()
    rng = np.random.default_rng(ord(fam))
    m = 4096
    xa, xb = rng.uniform(0, 2 * np.pi, m), rng.uniform(0, 2 * np.pi, m)
    ya, yb = rng.uniform(-3, 3, m), rng.uniform(-3, 3, m)
    l = [0.7, 1.3] + ([0.6] if fam == "D" else [])
    lv = L.f64(l)
    for dl, bit in ((0, 0), (1, L.K_DLX), (2, L.K_DLY)):
        for w in range(4):
            gen = np.empty(m)
            L.check(probe.sgpr_probe_generated_eval(L.family_id(fam), w | (dl << 2), m, L.dptr(xa), L.dptr(ya), L.dptr(xb),
                                                    L.dptr(yb), L.dptr(lv), len(lv), L.dptr(gen)))
            hand = ops.kernel_eval(w | bit, xa, ya, xb, yb, l, family=fam)
            scale = max(np.abs(gen).max(), 1e-300)
            err = np.abs(hand - gen).max() / scale
            assert err < 2e-15, (fam, NAMES[w], dl, err)
