"""ctypes binding of libdqmc_hip.so (include/dqmc_hip.h).

The product path has no CPU fallback: if the HIP library is missing or cannot be
loaded, importing the symbols fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DQMC_HIP_LIB selects another build of the same library (diagnostic / A-B builds), never a fallback
LIB_PATH = os.environ.get("DQMC_HIP_LIB") or os.path.join(_HERE, "libdqmc_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "dqmc_hip.h")

OK, ERR_INVALID, ERR_HIP, ERR_NO_DEVICE, ERR_STATE, ERR_RNG = 0, -1, -2, -3, -4, -5
ATTRACTIVE, REPULSIVE = 0, 1
K_FAMILIES = ("gemm", "qr", "trsm", "sweep", "misc", "flush")


class DQMCError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libdqmc_hip error %d: %s" % (code, msg))
        self.code = code


class Params(C.Structure):
    _fields_ = [
        ("n_sites", C.c_int32), ("model_kind", C.c_int32), ("slices", C.c_int32), ("safe_mult", C.c_int32),
        ("n_walkers", C.c_int32), ("device_id", C.c_int32), ("check_propagation_error", C.c_int32),
        ("check_sign_problem", C.c_int32), ("delta_tau", C.c_double), ("U", C.c_double),
        ("eT", C.POINTER(C.c_double)), ("eTinv", C.POINTER(C.c_double)), ("eT2", C.POINTER(C.c_double)),
        ("eTinv2", C.POINTER(C.c_double)),
    ]


class MagStats(C.Structure):
    _fields_ = [("max", C.c_double), ("min", C.c_double), ("sum", C.c_double), ("count", C.c_int64)]


class Stats(C.Structure):
    _fields_ = [("prop_local", C.c_int64), ("acc_local", C.c_int64), ("imaginary_probability", MagStats),
                ("negative_probability", MagStats), ("propagation_error", MagStats)]


_dp = C.POINTER(C.c_double)
_i64p = C.POINTER(C.c_int64)
_H = C.c_void_p

# name -> (restype, argtypes); every symbol declared in include/dqmc_hip.h
SIGNATURES = {
    "dqmc_create": (C.c_int, [C.POINTER(Params), C.POINTER(_H)]),
    "dqmc_destroy": (C.c_int, [_H]),
    "dqmc_last_error": (C.c_char_p, [_H]),
    "dqmc_device_count": (C.c_int, []),
    "dqmc_set_conf": (C.c_int, [_H, C.c_int32, C.c_void_p]),
    "dqmc_get_conf": (C.c_int, [_H, C.c_int32, C.c_void_p]),
    "dqmc_get_conf_bits": (C.c_int, [_H, C.c_int32, C.POINTER(C.c_uint64)]),
    "dqmc_set_conf_bits": (C.c_int, [_H, C.c_int32, C.POINTER(C.c_uint64)]),
    "dqmc_set_uniforms": (C.c_int, [_H, C.c_int32, _dp, C.c_size_t]),
    "dqmc_seed": (C.c_int, [_H, C.c_int32, C.c_uint64]),
    "dqmc_uniforms_used": (C.c_int, [_H, C.c_int32, C.POINTER(C.c_uint64)]),
    "dqmc_get_state": (C.c_int, [_H, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "dqmc_prepare": (C.c_int, [_H]),
    "dqmc_build_stack": (C.c_int, [_H]),
    "dqmc_propagate": (C.c_int, [_H]),
    "dqmc_sweep_spatial": (C.c_int, [_H]),
    "dqmc_update": (C.c_int, [_H]),
    "dqmc_sweep": (C.c_int, [_H, C.c_int32]),
    "dqmc_update_until_measure": (C.c_int, [_H, C.POINTER(C.c_int32)]),
    "dqmc_synchronize": (C.c_int, [_H]),
    "dqmc_get_greens_eff": (C.c_int, [_H, C.c_int32, _dp]),
    "dqmc_set_greens_eff": (C.c_int, [_H, C.c_int32, _dp]),
    "dqmc_get_greens": (C.c_int, [_H, C.c_int32, _dp]),
    "dqmc_calculate_greens_at": (C.c_int, [_H, C.c_int32, C.c_int32, _dp]),
    "dqmc_replay_greens": (C.c_int, [_H, C.c_int32]),
    "dqmc_wrap_greens": (C.c_int, [_H, C.c_int32, C.c_int32]),
    "dqmc_get_stats": (C.c_int, [_H, C.c_int32, C.POINTER(Stats)]),
    "dqmc_accumulate_greens": (C.c_int, [_H]),
    "dqmc_accumulator_size": (C.c_int, [_H, C.POINTER(C.c_size_t)]),
    "dqmc_reset_accumulators": (C.c_int, [_H]),
    "dqmc_get_accumulators": (C.c_int, [_H, _dp]),
    "dqmc_export_accumulators": (C.c_int, [_H, C.c_void_p]),
    "dqmc_set_pair_directions": (C.c_int, [_H, C.POINTER(C.c_int32), C.c_int32]),
    "dqmc_accumulate_correlations": (C.c_int, [_H]),
    "dqmc_correlations_size": (C.c_int, [_H, C.POINTER(C.c_size_t)]),
    "dqmc_get_correlations": (C.c_int, [_H, _dp]),
    "dqmc_export_correlations": (C.c_int, [_H, C.c_void_p]),
    "dqmc_set_local_targets": (C.c_int, [_H, C.POINTER(C.c_int32), C.c_int32]),
    "dqmc_accumulate_pairing": (C.c_int, [_H]),
    "dqmc_pairing_size": (C.c_int, [_H, C.POINTER(C.c_size_t)]),
    "dqmc_get_pairing": (C.c_int, [_H, _dp]),
    "dqmc_export_pairing": (C.c_int, [_H, C.c_void_p]),
    "dqmc_ut_build_stack": (C.c_int, [_H]),
    "dqmc_ut_get_stack": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_int32, _dp, _dp, _dp]),
    "dqmc_ut_greens": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_int32]),
    "dqmc_ut_get": (C.c_int, [_H, C.c_int32, C.c_int32, _dp]),
    "dqmc_ut_export": (C.c_int, [_H, C.c_int32, C.c_void_p]),
    "dqmc_greens_iterator_begin": (C.c_int, [_H, C.c_int32, C.c_int32]),
    "dqmc_greens_iterator_next": (C.c_int, [_H, C.POINTER(C.c_int32)]),
    "dqmc_combined_iterator_begin": (C.c_int, [_H, C.c_int32]),
    "dqmc_combined_iterator_next": (C.c_int, [_H, C.POINTER(C.c_int32)]),
    "dqmc_accumulate_susceptibilities": (C.c_int, [_H, C.c_int32]),
    "dqmc_susceptibilities_size": (C.c_int, [_H, C.POINTER(C.c_size_t)]),
    "dqmc_get_susceptibilities": (C.c_int, [_H, _dp]),
    "dqmc_export_susceptibilities": (C.c_int, [_H, C.c_void_p]),
    "dqmc_comm_unique_id": (C.c_int, [C.c_void_p]),
    "dqmc_comm_init": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(_H)]),
    "dqmc_comm_destroy": (C.c_int, [_H]),
    "dqmc_reduce": (C.c_int, [_H, _H]),
    "dqmc_reduce_size": (C.c_int, [_H, C.POINTER(C.c_size_t)]),
    "dqmc_reduce_export": (C.c_int, [_H, _dp]),
    "dqmc_reduce_import": (C.c_int, [_H, _dp]),
    "dqmc_get_reduced": (C.c_int, [_H, C.c_int32, _dp]),
    "dqmc_get_reduced_stats": (C.c_int, [_H, C.POINTER(Stats)]),
    "dqmc_vmul": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _dp, _dp, _dp]),
    "dqmc_udt_pivot": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _dp, _dp, _dp, _i64p, C.c_int32]),
    "dqmc_rdivp": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _dp, _dp, _i64p]),
    "dqmc_calculate_greens": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _dp, _dp, _dp, _dp, _dp, _dp, _dp]),
    "dqmc_set_checkerboard": (C.c_int, [_H, C.c_int32, C.c_int32, _dp, C.POINTER(C.c_int32), _dp, _dp,
                              C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "dqmc_qr_fallbacks": (C.c_int, [_H, C.POINTER(C.c_int64)]),
    "dqmc_device_errors": (C.c_int, [_H, C.POINTER(C.c_int32)]),
    "dqmc_udt_one_launch_sites": (C.c_int, [_H, C.POINTER(C.c_int32)]),
    "dqmc_build_commit": (C.c_char_p, []),
    "dqmc_build_source_hash": (C.c_char_p, []),
    "dqmc_timing_enable": (C.c_int, [_H, C.c_int32]),
    "dqmc_timing_get": (C.c_int, [_H, _dp, _i64p]),
    "dqmc_mfma_f64_peak": (C.c_int, [C.c_int32, C.c_int32, _dp]),
}

_lib = None


def lib():
    """Load libdqmc_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libdqmc_hip.so not found at %s — build it with __graft_entry__.build() "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)  # AttributeError if the export is missing
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def check(rc, handle=None):
    if rc != 0:
        msg = lib().dqmc_last_error(handle)
        raise DQMCError(rc, msg.decode() if msg else "")


def dptr(a):
    return a.ctypes.data_as(_dp)


def i64ptr(a):
    return a.ctypes.data_as(_i64p)
