"""The task-queue Cholesky (default for 13312 <= n <= 28672; the switches are read once per process, so the checks run in a
child process with SGPR_Q_MIN lowered to reach small orders too) against SciPy through the host entry point,
sgpr_potrf_host = scipy.linalg.cholesky(lower=True) of python/functions/func.py:166,184,193."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_check(sizes, tune=()):
    env = dict(os.environ, SGPR_POTRF_Q="1", SGPR_Q_MIN="2048")
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", "queue_check.py")] + [str(s) for s in sizes] + list(tune),
                          env=env, capture_output=True, text=True, timeout=600)


def test_queue_factor_vs_scipy():
    sizes = [2048, 2304, 4096, 6144, 14336]
    r = run_check(sizes)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count(" ok") == len(sizes), r.stdout


def test_queue_recovers_when_workgroups_are_switched_out():
    """The reproducer of DESIGN 3.9: the queue factors the first panels, the look-ahead driver's launches for the rest are
    enqueued BEHIND the persistent kernels (the tunables q_tail + q_nosync, set through libsympgpr_probe.so) -- on this platform that switches the running
    queues out and in, and some workgroups of the exactly-full worker grid do not get a CU back.  Before the drain-and-
    relaunch recovery this stalled in the first factorisation, every time; now the waiters notice that nothing moves,
    leave, the stranded workgroups finish, and the next kernel instance carries on.  Also the hand-over variant itself."""
    r = run_check([14336], ["q_tail=6144", "q_nosync=1"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count(" ok") == 1, r.stdout


def test_queue_not_positive_definite_reports_lapack_info():
    code = (
        "import numpy as np, sys\n"
        "sys.path.insert(0, %r)\n"
        "from sympgpr_amd import ops\n"
        "n, bad = 4096, 2900\n"
        "rng = np.random.default_rng(3)\n"
        "B = rng.standard_normal((n, n + 3)); A = B @ B.T / n + 0.5 * np.eye(n)\n"
        "A[bad, bad] = -1.0\n"
        "try:\n"
        "    ops.cholesky(A)\n"
        "    print('NO ERROR')\n"
        "except np.linalg.LinAlgError as e:\n"
        "    print('LinAlgError', str(e))\n" % ROOT)
    env = dict(os.environ, SGPR_POTRF_Q="1", SGPR_Q_MIN="2048")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert "LinAlgError %d-th leading minor" % 2901 in r.stdout, r.stdout[-1000:] + r.stderr[-1000:]


def test_queue_give_up_ends_in_a_result():
    """A task-queue factorisation that gives up (a hand-off between its persistent kernels timing out; forced here through
    the test-only entry of libsympgpr_probe.so) leaves Ky half overwritten -- the callers that still hold the matrix's source
    build it again and factor with the look-ahead driver, which the device then keeps using: the fit handle (alpha against
    LAPACK) and sgpr_potrf_host (L against SciPy); the device-pointer caller gets SGPR_E_HIP from sgpr_potrf_info_dev."""
    code = (
        "import numpy as np, sys, scipy.linalg\n"
        "sys.path.insert(0, %r)\n"
        "from sympgpr_amd import _lib as L, ops\n"
        "from sympgpr_amd.fit import SympFit\n"
        "from bench import synth\n"
        "lib, probe = L.load_library(), L.load_probe_library()\n"
        "N = 2048\n"
        "q, P, z, hyp, s2 = synth(N)\n"
        "with SympFit('A', q, P, z, hyp, s2) as f:\n"
        "    f.build(); K = f.matrix()\n"
        "    a_ref = scipy.linalg.cho_solve(scipy.linalg.cho_factor(K, lower=True), z)\n"
        "    probe.sgpr_probe_queue_force_giveup(1)\n"
        "    a = f.run().alpha()                      # the queue gives up, the handle rebuilds and factors again\n"
        "    print('fit', float(np.linalg.norm(a - a_ref) / np.linalg.norm(a_ref)))\n"
        "    a2 = f.run().alpha()                     # the device stays on the look-ahead driver: no second give-up\n"
        "    print('fit2', float(np.linalg.norm(a2 - a_ref) / np.linalg.norm(a_ref)))\n"
        "lib.sgpr_release_device_streams(0)           # the streams come back, and with them the queue\n"
        "Lh = ops.cholesky(K)                         # gives up again, retried from the host copy\n"
        "print('host', float(np.abs(Lh @ Lh.T - K).max() / np.abs(K).max()))\n"
        "print('info', lib.sgpr_potrf_info_dev(-1000001, None), lib.sgpr_potrf_info_dev(7, None), lib.sgpr_potrf_info_dev(0, None))\n" % ROOT)
    env = dict(os.environ, SGPR_POTRF_Q="1", SGPR_Q_MIN="2048")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    out = dict(l.split(" ", 1) for l in r.stdout.strip().splitlines() if " " in l)
    assert r.returncode == 0 and set(out) >= {"fit", "fit2", "host", "info"}, r.stdout[-1500:] + r.stderr[-1500:]
    assert float(out["fit"]) < 1e-10 and float(out["fit2"]) < 1e-10 and float(out["host"]) < 1e-13, out
    assert out["info"].split() == ["-3", "7", "0"], out
