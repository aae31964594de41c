"""ctypes binding of libsympgpr_hip.so (include/sympgpr_hip.h).  No compute happens here."""
import ctypes as C
import os

import numpy as np

FAMILIES = {"A": 0, "B": 1, "C": 2, "D": 3, "USER": 4}   # USER: the generated-only slot (tools/gen_kernels.py)
K_KERN, K_DXDX0, K_DYDY0, K_DXDY0 = 0, 1, 2, 3
K_DLX, K_DLY = 4, 8
K_DX, K_DY, K_DX0, K_DY0, K_DXDX0DY0, K_DYDY0DY0, K_DXDY0DY0 = 16, 17, 18, 19, 20, 21, 22
G_QQ, G_PQ, G_QP, G_PP, G_ALL, G_LOWER, G_OCML, G_DLX, G_DLY = 1, 2, 4, 8, 15, 16, 32, 64, 128
FIT_LOWER_ONLY, FIT_REG, FIT_BLOCK_QQ, FIT_BLOCK_PP = 1, 4, 8, 16
MAP_WRAP_Q, MAP_WRAP_P, MAP_EXPLICIT, MAP_LOSS_NEGP = 1, 2, 4, 8
E_ARG, E_NODEVICE, E_HIP, E_NOMEM, E_STATE = -1, -2, -3, -4, -5

ABI_VERSION = 5      # include/sympgpr_hip.h: SGPR_ABI_VERSION
_dp = C.POINTER(C.c_double)
_vp = C.c_void_p


class SympGPRError(RuntimeError):
    pass


class NoDeviceError(SympGPRError):
    pass


def lib_path():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libsympgpr_hip.so")


# every symbol include/sympgpr_hip.h declares: (restype, argtypes)
SIGNATURES = {
    "sgpr_abi_version": (C.c_int, []),
    "sgpr_last_error": (C.c_char_p, []),
    "sgpr_device_count": (C.c_int, []),
    "sgpr_set_device": (C.c_int, [C.c_int]),
    "sgpr_build_k_host": (C.c_int, [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, C.c_int, _dp, C.c_size_t]),
    "sgpr_buildkreg_host": (C.c_int, [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, C.c_int, _dp, C.c_size_t]),
    "sgpr_build_dk_host": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, C.c_int, _dp,
                                     C.c_size_t]),
    "sgpr_build_dkreg_host": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, C.c_int, _dp,
                                        C.c_size_t]),
    "sgpr_build_k_nd_host": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _dp, C.c_size_t, _dp, C.c_size_t, _dp,
                                       C.c_int, _dp, C.c_size_t]),
    "sgpr_kernel_eval_host": (C.c_int, [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, C.c_int, _dp]),
    "sgpr_potrf_host": (C.c_int, [C.c_int, _dp, C.c_size_t]),
    "sgpr_potrs_host": (C.c_int, [C.c_int, _dp, C.c_size_t, _dp, C.c_size_t, C.c_int]),
    "sgpr_fit_create": (C.c_int, [C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_int, C.c_double, C.c_uint, _vp,
                                  C.POINTER(_vp)]),
    "sgpr_fit_create_nd": (C.c_int, [C.c_int, C.c_int, C.c_int, _dp, C.c_size_t, _dp, _dp, C.c_int, C.c_double,
                                     C.c_uint, _vp, C.POINTER(_vp)]),
    "sgpr_fit_set_hyp": (C.c_int, [_vp, _dp, C.c_int, C.c_double]),
    "sgpr_fit_set_targets": (C.c_int, [_vp, _dp]),
    "sgpr_fit_build": (C.c_int, [_vp]),
    "sgpr_fit_factor": (C.c_int, [_vp]),
    "sgpr_fit_solve": (C.c_int, [_vp]),
    "sgpr_fit_run": (C.c_int, [_vp]),
    "sgpr_fit_alpha": (C.c_int, [_vp, _dp]),
    "sgpr_fit_nll": (C.c_int, [_vp, _dp]),
    "sgpr_fit_ldiag": (C.c_int, [_vp, _dp]),
    "sgpr_fit_get_matrix": (C.c_int, [_vp, _dp, C.c_size_t]),
    "sgpr_fit_solve_rhs": (C.c_int, [_vp, _dp, C.c_size_t, C.c_int]),
    "sgpr_fit_solve_rhs_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    "sgpr_fit_predict_rows": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp, _dp]),
    "sgpr_fit_nll_grad": (C.c_int, [_vp, _dp]),
    "sgpr_fit_nll_grad_terms": (C.c_int, [_vp, _dp]),
    "sgpr_fit_eig": (C.c_int, [_vp, _dp, _dp]),
    "sgpr_syev_host": (C.c_int, [C.c_int, _dp, C.c_size_t, _dp]),
    "sgpr_fit_inverse": (C.c_int, [_vp, _dp, C.c_size_t]),
    "sgpr_fit_predict_nd": (C.c_int, [_vp, C.c_int, _dp, C.c_size_t, _dp]),
    "sgpr_fit_stage_ms": (C.c_int, [_vp, _dp, _dp, _dp]),
    "sgpr_fit_cond_estimate": (C.c_int, [_vp, C.c_int, _dp]),
    "sgpr_fit_trim": (C.c_int, [_vp]),
    "sgpr_fit_solve_rhs_ms": (C.c_int, [_vp, _dp]),
    "sgpr_potrf_info_dev": (C.c_int, [C.c_int, _vp]),
    "sgpr_trim": (C.c_int, []),
    "sgpr_fit_device_ptrs": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(C.c_size_t), C.POINTER(_vp)]),
    "sgpr_fit_destroy": (C.c_int, [_vp]),
    "sgpr_gram_pairs_dev": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _dp, C.c_int, _vp, _vp, _vp,
                                      _vp, C.c_size_t, C.c_long, C.c_double, C.c_uint, _vp]),
    "sgpr_gram_nd_dev": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_size_t, _vp, C.c_size_t, _dp, C.c_int,
                                   _vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_long, C.c_double, _vp]),
    "sgpr_gram_reg_dev": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _dp, C.c_int, _vp, C.c_size_t,
                                    C.c_long, C.c_double, _vp]),
    "sgpr_gram_nd_sel_dev": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_size_t, _vp, C.c_size_t, _dp, C.c_int,
                                       _vp, C.c_size_t, C.POINTER(C.c_long), C.POINTER(C.c_long), _vp]),
    "sgpr_potrf_workspace": (C.c_size_t, [C.c_int]),
    "sgpr_potrf_inverses_bytes": (C.c_size_t, [C.c_int]),
    "sgpr_release_device_streams": (C.c_int, [C.c_int]),
    "sgpr_family_has_p": (C.c_int, [C.c_int]),
    "sgpr_potrf_dev": (C.c_int, [C.c_int, _vp, C.c_size_t, _vp, C.c_size_t, _vp, _vp]),
    "sgpr_trsm_rlt_dev": (C.c_int, [C.c_int, C.c_int, _vp, C.c_size_t, _vp, C.c_size_t, _vp, _vp]),
    "sgpr_gemm_nt_dev": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_double, _vp, C.c_size_t, _vp, C.c_size_t,
                                   C.c_double, _vp, C.c_size_t, C.c_int, C.c_long, _vp]),
    "sgpr_profile_begin": (C.c_int, []),
    "sgpr_profile_end": (C.c_int, [_dp]),
    "sgpr_profile_launches": (C.c_int, [_dp, C.c_int]),
    "sgpr_gemm_nt_bc_dev": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_double, _vp, C.c_size_t, _vp, C.c_size_t,
                                      C.c_double, _vp, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "sgpr_trsv_dev": (C.c_int, [C.c_int, _vp, C.c_size_t, _vp, _vp, C.c_int, _vp]),
    "sgpr_gemv_sub_dev": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, C.c_size_t, _vp, _vp, _vp]),
    "sgpr_gemm_nn_dev": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_double, _vp, C.c_size_t, _vp, C.c_size_t, C.c_double, _vp,
                                   C.c_size_t, _vp]),
    "sgpr_trsm_rl_dev": (C.c_int, [C.c_int, C.c_int, _vp, C.c_size_t, _vp, C.c_size_t, _vp, _vp]),
    "sgpr_copy_blocks_dev": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, C.c_size_t, C.c_size_t, _vp, C.c_size_t, C.c_size_t, _vp]),
    "sgpr_predict_rows_dev": (C.c_int, [C.c_int, C.c_int, _vp, _vp, C.c_int, _vp, _vp, _dp, C.c_int, _vp, _vp, _vp,
                                        _vp]),
    "sgpr_predict_nd_dev": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, C.c_size_t, C.c_int, _vp, C.c_size_t, _dp, C.c_int,
                                      _vp, _vp, _vp]),
    "sgpr_predict_reg_dev": (C.c_int, [C.c_int, C.c_int, _vp, _vp, C.c_int, _vp, _vp, _dp, C.c_int, _vp, _vp, _vp]),
    "sgpr_applymap_host": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _dp, C.c_int, C.c_int, _dp, _dp, _dp, _dp,
                                     C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]),
    "sgpr_potrs_vec_dev": (C.c_int, [C.c_int, _vp, C.c_size_t, _vp, _vp, _vp]),
    "sgpr_solve_status_dev": (C.c_int, [C.c_int, _vp, C.c_size_t, _vp, _vp]),
    "sgpr_fit_batch_max_order": (C.c_int, []),
    "sgpr_fit_batch": (C.c_int, [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_int, _dp, C.c_uint, _dp, _dp,
                                 C.POINTER(C.c_int)]),
}

# include/sympgpr_probe.h: measurement aids in their own library (never loaded by a product path)
PROBE_SIGNATURES = {
    "sgpr_probe_mfma_f64": (C.c_int, [C.c_int, C.c_int, _dp]),
    "sgpr_probe_mfma_clock": (C.c_int, [C.c_int, C.c_int, C.c_int, _dp]),
    "sgpr_probe_gemm": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _dp]),
    "sgpr_probe_gemm_debug": (C.c_int, [C.c_int]),
    "sgpr_probe_leaf": (C.c_int, [_dp]),
    "sgpr_probe_lat": (C.c_int, [_dp]),
    "sgpr_probe_cumask": (C.c_int, [C.POINTER(C.c_uint), C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "sgpr_probe_xcc": (C.c_int, [C.c_int, C.POINTER(C.c_int)]),
    "sgpr_probe_hbm_write": (C.c_int, [C.c_size_t, C.c_int, _dp]),
    "sgpr_probe_generated_eval": (C.c_int, [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, C.c_int, _dp]),
    "sgpr_probe_queue_plan": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_uint), C.c_int,
                                        C.POINTER(C.c_int)]),
    "sgpr_probe_queue_plan_partial": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_uint),
                                                C.c_int, C.POINTER(C.c_int)]),
    "sgpr_probe_queue_trace_begin": (C.c_int, [C.c_int]),
    "sgpr_probe_queue_trace_end": (C.c_int, [C.POINTER(C.c_ulonglong), C.c_int]),
    "sgpr_probe_queue_postmortem": (C.c_int, [C.c_int]),
    "sgpr_probe_queue_force_giveup": (C.c_int, [C.c_int]),
    "sgpr_probe_tune": (C.c_int, [C.c_char_p, C.c_double]),
    "sgpr_probe_map_calls": (C.c_uint, []),
    "sgpr_probe_map_team": (C.c_int, [C.c_int, C.c_int]),
    "sgpr_probe_trsm_piece": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "sgpr_probe_trsm_counts": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_ulonglong)]),
    "sgpr_probe_queue_trace_clear": (C.c_int, []),
    "sgpr_probe_census": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint), C.c_int,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint), C.c_int,
                                    C.POINTER(C.c_ulonglong)]),
}

_LIB = None
_PROBE = None


def load_library():
    """Load libsympgpr_hip.so and bind every declared symbol.  Raises SympGPRError when the
    library has not been built -- there is deliberately no fallback implementation."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64 /
    # libhsa-runtime64.  Loaded after /opt/rocm's copy they become a SECOND runtime that finds
    # no GPU ("No HIP GPUs are available"); loaded first, our NEEDED libamdhip64.so.7 resolves
    # to the copy already in the process.  So when torch is installed, it goes first.
    if os.environ.get("SYMPGPR_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(path):
        raise SympGPRError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(or make -C sympgpr_amd/csrc); there is no CPU fallback" % path)
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.sgpr_abi_version() != ABI_VERSION:
        raise SympGPRError("ABI version mismatch: %s is version %d, this package binds version %d -- rebuild it "
                           "(make -C sympgpr_amd/csrc)" % (path, lib.sgpr_abi_version(), ABI_VERSION))
    _LIB = lib
    # The library keeps a few streams per device (a high-priority side stream, a CU-masked pair for the task-queue Cholesky).
    # Hand them back while the HIP runtime is still whole: left to process teardown, a run under rocprofv3 that had used the
    # masked streams crashed in an exit handler after the profile was written.
    import atexit
    atexit.register(_release_streams_at_exit)
    return lib


def _release_streams_at_exit():
    lib = _LIB
    if lib is None:
        return
    try:
        for dev in range(max(0, lib.sgpr_device_count())):
            lib.sgpr_release_device_streams(dev)
    except Exception:
        pass


def load_probe_library():
    """libsympgpr_probe.so (include/sympgpr_probe.h): calibration probes and kernel diagnostics for
    tools/ and bench.py's roofline calibration.  Links against libsympgpr_hip.so."""
    global _PROBE
    if _PROBE is not None:
        return _PROBE
    load_library()
    path = os.path.join(os.path.dirname(lib_path()), "libsympgpr_probe.so")
    if not os.path.exists(path):
        raise SympGPRError("%s is missing: run make -C sympgpr_amd/csrc" % path)
    lib = C.CDLL(path)
    for name, (res, args) in PROBE_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _PROBE = lib
    return lib


def device_count():
    return load_library().sgpr_device_count()


def check(rc, what=""):
    """0 -> ok; >0 -> LinAlgError like SciPy's cholesky; <0 -> SympGPRError."""
    if rc == 0:
        return
    if rc > 0:
        raise np.linalg.LinAlgError("%d-th leading minor of the array is not positive definite" % rc)
    msg = load_library().sgpr_last_error().decode(errors="replace")
    if rc == E_NODEVICE:
        raise NoDeviceError(msg)
    raise SympGPRError("%s failed (%d): %s" % (what or "libsympgpr_hip call", rc, msg))


def dptr(a):
    return a.ctypes.data_as(_dp)


def f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def family_has_p(fam):
    """family D, or a user kernel that uses the period parameter: hyp = (lx, ly, p, sig)"""
    return bool(load_library().sgpr_family_has_p(family_id(fam)))


def family_id(fam):
    if isinstance(fam, str):
        return FAMILIES[fam]
    return int(fam)
