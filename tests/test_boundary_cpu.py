"""CPU suite for the boundary: libsympgpr_hip.so loads, exports every symbol the header declares,
the ctypes table matches the header, the Python mirror has the reference's call surface, and
compute calls FAIL LOUDLY without a GPU (no CPU fallback)."""
import inspect
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "sympgpr_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sgpr_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import ctypes
    from sympgpr_amd import _lib
    lib = ctypes.CDLL(_lib.lib_path())
    syms = _header_symbols()
    assert len(syms) >= 40
    for s in syms:
        assert hasattr(lib, s), "libsympgpr_hip.so does not export " + s


def test_library_exports_nothing_but_the_c_abi_and_the_probe_hooks():
    """csrc/exports.map: the dynamic symbol table of libsympgpr_hip.so holds the C entry points of include/sympgpr_hip.h and the
    internal entry points the measurement library shares state through (listed in the map) -- no other C++ internals"""
    import subprocess
    from sympgpr_amd import _lib
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.lib_path()], capture_output=True, text=True, check=True).stdout
    names = [ln.split()[-1] for ln in out.splitlines() if " T " in ln or " W " in ln or " B " in ln or " D " in ln]
    c_abi = sorted(n for n in names if n.startswith("sgpr_"))
    assert c_abi == _header_symbols()
    rest = [n for n in names if not n.startswith("sgpr_")]
    hooks = open(os.path.join(ROOT, "sympgpr_amd", "csrc", "exports.map")).read()
    stems = re.findall(r"sgpr::([A-Za-z_:]+)\*;", hooks)
    assert stems and len(rest) <= len(stems) + 4, rest
    for n in rest:
        assert n.startswith("_ZN4sgpr"), n                      # mangled sgpr::...
        assert any(st.split("::")[-1] in n for st in stems), n


def test_ctypes_table_matches_header():
    from sympgpr_amd import _lib
    assert sorted(_lib.SIGNATURES) == _header_symbols()
    lib = _lib.load_library()
    assert lib.sgpr_abi_version() == 5


def test_mirror_has_reference_call_surface():
    """names and positional parameters of python/functions/func.py."""
    from sympgpr_amd import func
    expect = {
        "f_kern": ["x", "y", "x0", "y0", "l"], "d2kdxdx0": ["x", "y", "x0", "y0", "l"],
        "d2kdydy0": ["x", "y", "x0", "y0", "l"], "d2kdxdy0": ["x", "y", "x0", "y0", "l"],
        "d2kdydx0": ["x", "y", "x0", "y0", "l"],
        "build_K": ["xin", "x0in", "hyp", "K"], "buildKreg": ["xin", "x0in", "hyp", "K"],
        "gpsolve": ["Ky", "ft"], "solve_cholesky": ["L", "b"],
        "nll_chol_reg": ["hyp", "x", "y", "N"], "nll_chol": ["hyp", "x", "y", "N"],
        "guessP": ["x", "y", "hypp", "xtrainp", "ztrainp", "Kyinvp"],
        "calcQ": ["x", "y", "xtrain", "l", "Kyinv", "ztrain"],
        "calcP": ["x", "y", "l", "hypp", "xtrainp", "ztrainp", "Kyinvp", "xtrain", "ztrain", "Kyinv"],
        "applymap": ["nm", "Ntest", "l", "hypp", "Q0map", "P0map", "xtrainp", "ztrainp", "Kyinvp", "xtrain",
                     "ztrain", "Kyinv"],
        "applymap_henon": ["nm", "Ntest", "l", "hypp", "Q0map", "P0map", "xtrainp", "ztrainp", "Kyinvp", "xtrain",
                           "ztrain", "Kyinv"],
        "quality": ["qmap", "pmap", "H", "ysint", "Ntest", "Nm"],
    }
    for name, params in expect.items():
        assert list(inspect.signature(getattr(func, name)).parameters) == params, name
    for name in ("kern_num", "d2kdxdx0_num", "d2kdydy0_num", "d2kdxdy0_num", "build_dK", "build_dKreg",
                 "nll_grad", "nll_grad_reg"):
        assert hasattr(func, name)
    from sympgpr_amd.fortran.sympgpr import sympgpr
    for name in ("build_k", "buildkreg", "guessp", "calcq", "calcp", "applymap_tok"):
        assert callable(getattr(sympgpr, name))


def test_dropin_modules_resolve(monkeypatch):
    import importlib
    import sys
    monkeypatch.syspath_prepend(os.path.join(ROOT, "sympgpr_amd", "dropin"))
    for m in ("func", "kernels", "fortran", "fortran.sympgpr"):
        sys.modules.pop(m, None)
    func = importlib.import_module("func")
    kernels = importlib.import_module("kernels")
    sp = importlib.import_module("fortran.sympgpr")
    assert func.build_K and kernels.kern_num and sp.sympgpr.build_k
    for m in ("func", "kernels", "fortran", "fortran.sympgpr"):
        sys.modules.pop(m, None)


def test_no_cpu_fallback_without_gpu():
    import sympgpr_amd
    from sympgpr_amd import ops
    from sympgpr_amd.fit import SympFit
    if sympgpr_amd.device_count() > 0:
        pytest.skip("a GPU is present")
    K = np.empty((4, 4), order="F")
    with pytest.raises(sympgpr_amd.NoDeviceError):
        ops.build_k([1.0, 2.0], [0.0, 1.0], [1.0, 2.0], [0.0, 1.0], [0.5, 2.0, 0.4], K)
    with pytest.raises(sympgpr_amd.NoDeviceError):
        ops.cholesky(np.eye(3))
    with pytest.raises(sympgpr_amd.NoDeviceError):
        SympFit("A", [1.0, 2.0], [0.0, 1.0], np.zeros(4), [0.5, 2.0, 0.4], 1e-3)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "sympgpr_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(d, f)).read()
                assert "oracle" not in txt.replace("oracle/", "").lower() or f in ("__init__.py",) or \
                    "import oracle" not in txt and "from oracle" not in txt, f


def test_host_math_matches_libm():
    """devmath.h (the in-house exp / sincos of the Gram kernels) compiled for the host with g++:
    exp <= 1 ulp-ish relative, sin/cos <= 1.5e-16 absolute on the arguments the kernels see."""
    import subprocess
    import tempfile
    src = r'''
#define SGPR_HOST_MATH_TEST
#include "devmath.h"
#include <cstdio>
#include <random>
int main(){ std::mt19937_64 g(1); std::uniform_real_distribution<double> ux(-745,1), uh(-1e4,1e4), us(-7,7);
 double me=0, ms=0, mc=0;
 for(int i=0;i<400000;i++){ double x=ux(g); if(i%2) x=-std::exp(us(g));
  double e=sgpr::exp_fast(x), r=std::exp(x); double d=std::fabs(e-r)/r; if(r>1e-300 && d>me) me=d;
  double h=(i%3)?us(g):uh(g); double s,c; sgpr::sincos_fast(h,s,c);
  long double sl=sinl((long double)h), cl=cosl((long double)h);
  double ds=std::fabs((double)(s-sl)), dc=std::fabs((double)(c-cl)); if(ds>ms)ms=ds; if(dc>mc)mc=dc; }
 printf("%.3e %.3e %.3e\n",me,ms,mc); }
'''
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "t.cpp"), "w").write(src)
        exe = os.path.join(td, "t")
        subprocess.run(["g++", "-O2", "-mfma", "-I", os.path.join(ROOT, "sympgpr_amd", "csrc"),
                        os.path.join(td, "t.cpp"), "-o", exe], check=True)
        me, ms, mc = map(float, subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split())
    assert me < 4e-16 and ms < 1.5e-16 and mc < 1.5e-16


def _probe_header_symbols():
    txt = open(os.path.join(ROOT, "include", "sympgpr_probe.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sgpr_[a-z0-9_]+)\s*\(", txt)))


def test_probe_library_is_separate_and_complete():
    """measurement aids live in libsympgpr_probe.so / include/sympgpr_probe.h: every declared symbol is
    exported there, bound in _lib.PROBE_SIGNATURES, and NOT part of the product header"""
    import ctypes
    from sympgpr_amd import _lib
    probe = ctypes.CDLL(os.path.join(os.path.dirname(_lib.lib_path()), "libsympgpr_probe.so"))
    syms = _probe_header_symbols()
    assert syms and sorted(_lib.PROBE_SIGNATURES) == syms
    for s in syms:
        assert hasattr(probe, s), "libsympgpr_probe.so does not export " + s
    assert not any(s.startswith("sgpr_probe") for s in _header_symbols())


def test_generated_kernels_are_up_to_date(tmp_path, monkeypatch):
    """tools/gen_kernels.py (sympy -> HIP, the device-side counterpart of the reference's init_func.py)
    reproduces the committed csrc/generated/pair_generated.h byte for byte"""
    pytest.importorskip("sympy")
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_kernels", os.path.join(ROOT, "tools", "gen_kernels.py"))
    gk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gk)
    out = tmp_path / "pair_generated.h"
    npz = tmp_path / "user_family.npz"
    monkeypatch.setattr(gk, "OUT", str(out))
    monkeypatch.setattr(gk, "NPZ_OUT", str(npz))          # (the tracked fixture must not be rewritten by a test run)
    golden = os.path.join(ROOT, "tests", "golden", "user_family.npz")
    before = os.stat(golden).st_mtime_ns
    gk.main()
    committed = open(os.path.join(ROOT, "sympgpr_amd", "csrc", "generated", "pair_generated.h")).read()
    assert out.read_text() == committed
    assert os.stat(golden).st_mtime_ns == before
    import numpy as np
    new, old = np.load(str(npz)), np.load(golden)
    assert sorted(new.files) == sorted(old.files)
    for k in old.files:
        if old[k].dtype.kind == "f":
            np.testing.assert_allclose(new[k], old[k], rtol=1e-13, atol=1e-300)


def test_bench_starts_its_own_ranks_before_touching_the_gpu(monkeypatch):
    """`python bench.py --gpus N` typed by hand: the parent only spawns `torch.distributed.run` children
    (no HIP call, no exec of a process that has initialised the GPU)"""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_call(cmd):
        seen["cmd"] = cmd
        return 7
    import subprocess
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    before = set(sys.modules)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "2"]
    assert "sympgpr_amd._lib" not in set(sys.modules) - before      # the product library was not even loaded


def test_family_selection_process_default_and_thread_scope():
    """func.set_family selects the mirrored kernels*.f90 PROCESS-wide, as the reference's compiled module does: a selection made in
    the main thread is what a new thread sees (an optimiser's callback pool).  ops.family_scope shadows it for the calling thread
    only: a scope in one thread does not leak into another, and set_family inside a scope stays inside it."""
    import threading
    from sympgpr_amd import func, ops
    seen = {}
    ready, go = threading.Event(), threading.Event()

    def other():
        seen["fresh"] = ops.get_family()               # the main thread's process-wide "C", not a hard-wired default
        with ops.family_scope("D"):
            ready.set()
            go.wait(10)
            seen["scoped"] = func.get_family()
        seen["after"] = func.get_family()

    func.set_family("C")
    try:
        with ops.family_scope("B"):
            t = threading.Thread(target=other)
            t.start()
            assert ready.wait(10)
            assert ops.get_family() == "B"              # the other thread's scope "D" did not land here
            func.set_family("A")                        # inside a scope: changes the scope, not the process default
            assert ops.get_family() == "A"
            go.set()
            t.join()
        assert ops.get_family() == "C"
        assert seen == {"fresh": "C", "scoped": "D", "after": "C"}
        with pytest.raises(ValueError):
            func.set_family("Z")
        with pytest.raises(ValueError):
            with ops.family_scope("Z"):
                pass
    finally:
        func.set_family("A")


def test_family_table_and_period_parameter():
    """the five family ids of include/sympgpr_hip.h and which of them carry a period p in hyp (no GPU needed)"""
    from sympgpr_amd import _lib as L
    assert L.FAMILIES == {"A": 0, "B": 1, "C": 2, "D": 3, "USER": 4}
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "sympgpr_hip.h")).read()
    for name, val in L.FAMILIES.items():
        assert re.search(r"SGPR_FAM_%s\s*=\s*%d\b" % (name, val), hdr), name
    assert L.family_has_p("D") and not any(L.family_has_p(f) for f in "ABC")
    assert L.family_has_p("USER") in (False, True)


@pytest.mark.parametrize("T,cap", [(1, 16), (5, 16), (22, 4), (128, 8), (133, 128), (134, 128), (300, 7), (768, 128), (1024, 64)])
def test_block_solve_ticket_order(T, cap):
    """The block solve deals its stream work out as tickets (csrc/trsm.hip: piece_of; host logic, no GPU): per strip the helper
    pieces of its streamed tiles, then the folds of its min(strip, F) tiles next to the diagonal, then the owner (the last piece)
    -- strips ascending, so every wait of a ticket is for a smaller one --, no piece longer than the cap, the partial-sum slots
    of the helpers dense and unique, and the totals the host sizes the grid and the scratch with."""
    import ctypes as C
    from sympgpr_amd import _lib as L
    probe = L.load_probe_library()
    F = 5                                                    # TRSM_FOLD: tiles next to the diagonal are folded, not streamed
    counts = (C.c_ulonglong * 2)()
    L.check(probe.sgpr_probe_trsm_counts(T, cap, counts))
    ntick, npart = int(counts[0]), int(counts[1])
    out = (C.c_int * 5)()
    expect, slots = [], set()
    for tk in range(T):
        ns = max(tk - F, 0)
        npc = max(1, -(-ns // cap))
        expect += [(tk, p, 0) for p in range(npc - 1)] + [(tk, 0, f) for f in range(1, min(tk, F) + 1)] + [(tk, npc - 1, 0)]
    assert ntick == len(expect)
    for u, (tk, p, f) in enumerate(expect):
        L.check(probe.sgpr_probe_trsm_piece(u, cap, T, out))
        got_tk, got_p, npc, x0, fold = (int(v) for v in out)
        assert (got_tk, got_p, fold) == (tk, p, f), (u, tuple(out))
        ns = max(tk - F, 0)
        assert npc == max(1, -(-ns // cap))
        if not fold:
            q0, q1 = p * ns // npc, (p + 1) * ns // npc       # the piece's share of the strip's tiles (stream_task)
            assert q1 - q0 <= cap
            if p < npc - 1:
                assert x0 + p not in slots
                slots.add(x0 + p)
    for u in (ntick, ntick + 1, ntick + 1000):
        L.check(probe.sgpr_probe_trsm_piece(u, cap, T, out))
        assert out[0] == T                                    # past the end: the grid's exit
    assert slots == set(range(npart))
