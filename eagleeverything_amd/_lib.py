"""ctypes loader for libeaglehip.so (the C ABI of include/eagle_hip.h).

The library is built in-tree by __graft_entry__.build() / `make -C eagleeverything_amd/csrc`.
There is no fallback: a missing library or a missing gfx950 device is an error.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libeaglehip.so")

c_dp = C.POINTER(C.c_double)
c_lp = C.POINTER(C.c_long)
MESSAGE_FN = C.CFUNCTYPE(None, C.c_char_p, C.c_void_p)


class EagleBest(C.Structure):
    _fields_ = [("tsqmax", C.c_double), ("index0", C.c_long), ("near_ties", C.c_long)]


# name -> (restype, argtypes); must list every symbol include/eagle_hip.h declares
SIGNATURES = {
    "eagle_open": (C.c_void_p, [C.c_int]),
    "eagle_open_devices": (C.c_void_p, [C.POINTER(C.c_int), C.c_int]),
    "eagle_open_env": (C.c_void_p, []),
    "eagle_device_count": (C.c_int, [C.c_void_p]),
    "eagle_open_error": (C.c_char_p, []),
    "eagle_close": (None, [C.c_void_p]),
    "eagle_last_error": (C.c_char_p, [C.c_void_p]),
    "eagle_set_message_callback": (None, [C.c_void_p, MESSAGE_FN, C.c_void_p]),
    "eagle_drop_cache": (None, [C.c_void_p]),
    "eagle_device_info": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "eagle_set_scan_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "eagle_set_scan_slices": (C.c_int, [C.c_void_p, C.c_int]),
    "eagle_set_scan_rounding": (C.c_int, [C.c_void_p, C.c_int]),
    "eagle_set_scan_budget": (C.c_int, [C.c_void_p, C.c_double]),
    "eagle_last_scan_budget": (C.c_int, [C.c_void_p, c_dp, C.POINTER(C.c_int), c_dp]),
    "eagle_last_scan_enforced": (C.c_int, [C.c_void_p, c_dp, C.POINTER(C.c_long)]),
    "eagle_prepare_scan": (C.c_int, [C.c_void_p, C.c_long, C.c_long]),
    "eagle_set_w_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "eagle_last_w_info": (C.c_int, [C.c_void_p, C.c_void_p]),
    "eagle_last_scan_digits": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "eagle_get_row_column": (C.c_int, [C.c_void_p, C.c_char_p, c_lp]),
    "eagle_create_M_ascii": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p,
                                       C.c_double, c_lp, C.c_int, C.c_char_p]),
    "eagle_create_Mt_ascii": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_double, c_lp, C.c_int]),
    "eagle_read_block": (C.c_int, [C.c_void_p, C.c_char_p, C.c_long, C.c_long, C.c_long, c_dp]),
    "eagle_calculateMMt": (C.c_int, [C.c_void_p, C.c_char_p, C.c_double, C.c_int, c_dp, C.c_long, c_lp, C.c_int, c_dp]),
    "eagle_calculate_a_and_vara": (C.c_int, [C.c_void_p, C.c_char_p, c_dp, C.c_long, c_dp, c_dp, C.c_double, c_lp, c_dp,
                                             C.c_int, c_dp, c_dp]),
    "eagle_spectral_prepare": (C.c_int, [C.c_void_p, C.c_char_p, c_lp, c_dp, C.c_double]),
    "eagle_spectral_scan": (C.c_int, [C.c_void_p, c_dp, c_dp, c_dp, C.c_long, C.c_double, C.c_double, c_dp, C.c_long, c_dp, c_dp]),
    "eagle_scan_with_W": (C.c_int, [C.c_void_p, C.c_char_p, c_dp, C.c_long, c_dp, c_dp, C.c_double, c_lp, C.c_int, c_dp, c_dp]),
    "eagle_calculate_reduced_a": (C.c_int, [C.c_void_p, C.c_char_p, C.c_double, c_dp, c_dp, C.c_double, c_lp, c_dp,
                                            C.c_long, C.c_int, c_dp]),
    "eagle_extract_geno": (C.c_int, [C.c_void_p, C.c_char_p, C.c_double, C.c_long, c_lp, C.POINTER(C.c_int)]),
    "eagle_sym_eig": (C.c_int, [C.c_void_p, c_dp, C.c_long, c_dp, c_dp]),
    "eagle_chol2inv": (C.c_int, [C.c_void_p, c_dp, C.c_long, c_dp]),
    "eagle_inverse": (C.c_int, [C.c_void_p, c_dp, C.c_long, c_dp]),
    "eagle_matmul": (C.c_int, [C.c_void_p, c_dp, c_dp, C.c_long, C.c_long, C.c_long, c_dp]),
    "eagle_mmt_sqrt_and_sqrtinv": (C.c_int, [C.c_void_p, c_dp, C.c_long, c_dp, c_dp, c_dp]),
    "eagle_last_scan_argmax": (C.c_int, [C.c_void_p, c_lp, c_dp, c_lp]),
    "eagle_last_mmt_normalised": (C.c_int, [C.c_void_p, c_dp, c_dp]),
    "eagle_pad": (C.c_long, [C.c_long]),
    "eagle_dev_load_ascii": (C.c_int, [C.c_void_p, C.c_char_p, C.c_long, C.c_long, C.c_long, C.c_long, C.c_void_p,
                                       C.c_long, C.c_double, C.c_int]),
    "eagle_dev_decode_ascii": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_void_p, C.c_long,
                                         C.c_void_p, C.c_void_p]),
    "eagle_dev_transpose_i8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_void_p, C.c_long,
                                         C.c_void_p]),
    "eagle_dev_i8_to_f64_colmajor": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_void_p,
                                               C.c_void_p]),
    "eagle_dev_mmt_accumulate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_void_p, C.c_void_p]),
    "eagle_dev_mmt_accumulate_f4": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_void_p, C.c_void_p]),
    "eagle_dev_mmt_downdate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_void_p, C.c_long, C.c_void_p,
                                         C.c_void_p]),
    "eagle_dev_mmt_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_void_p, C.c_long, C.c_void_p,
                                       C.c_void_p]),
    "eagle_dev_mmt_normalise": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_void_p, C.c_void_p]),
    "eagle_dev_scan_operands_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_long,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "eagle_dev_fold_upper": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]),
    "eagle_dev_scan_operands": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_long,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "eagle_dev_gemv_i8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_void_p, C.c_double,
                                    C.c_void_p, C.c_void_p]),
    "eagle_dev_vara_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    "eagle_vara_i8_workspace_bytes": (C.c_int64, [C.c_long, C.c_long, C.c_int]),
    "eagle_dev_marker_shift": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_long, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p]),
    "eagle_scan_certify_workspace_bytes": (C.c_int64, [C.c_long]),
    "eagle_dev_scan_certify": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_long, C.c_void_p, C.c_void_p, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "eagle_dev_scan_certify_lb": (C.c_int, [C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "eagle_dev_scan_certify_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_long, C.c_void_p, C.c_void_p,
                                               C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]),
    "eagle_last_scan_certificate": (C.c_int, [C.c_void_p, c_lp, c_lp, C.POINTER(C.c_int)]),
    "eagle_last_stream_stats": (C.c_int, [C.c_void_p, C.c_void_p]),
    "eagle_last_scan_timing": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "eagle_scan_operand_cache_stats": (C.c_int, [C.c_void_p, c_lp, c_lp]),
    "eagle_dev_vara_i8_extend": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_long, C.c_int, C.c_void_p,
                                           C.c_void_p, C.c_void_p]),
    "eagle_dev_vara_i8_mfma_shifted": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_int, C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.c_void_p]),
    "eagle_vara_f6_workspace_bytes": (C.c_int64, [C.c_long, C.c_long, C.c_int]),
    "eagle_dev_pack_fp4": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_void_p, C.c_void_p]),
    "eagle_dev_transpose_pack_fp4": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_void_p, C.c_long, C.c_void_p]),
    "eagle_dev_vara_f6_prepare": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_void_p, C.c_int,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "eagle_dev_vara_f6_mfma": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p]),
    "eagle_dev_vara_i8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_void_p, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "eagle_dev_vara_i8_prepare": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_void_p, C.c_int,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "eagle_dev_vara_i8_mfma": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p]),
    "eagle_dev_spectral_zbuild": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_void_p, C.c_void_p, C.c_void_p]),
    "eagle_spectral_zbuild_i8_workspace_bytes": (C.c_int64, [C.c_long, C.c_int]),
    "eagle_dev_spectral_zbuild_i8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                               C.c_void_p]),
    "eagle_dev_spectral_pass": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p]),
    "eagle_dev_spectral_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_long, C.c_long, C.c_void_p, C.c_void_p, C.c_double,
                                            C.c_void_p, C.c_void_p, C.c_void_p]),
    "eagle_dev_zero_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_long,
                                      C.c_void_p]),
    "eagle_dev_tsq_argmax": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p]),
    # internal helpers exported for tests / the sharded driver
    "eagle_dev_set_tune": (None, [C.c_void_p, C.c_int]),
    "eagle_w8_host_work_list": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_long,
                                          C.POINTER(C.c_int), C.c_void_p, C.POINTER(C.c_int)]),
    "eagle_w8_host_bound": (C.c_double, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_long]),
    "eagle_dev_set_spectral": (None, [C.c_void_p, C.c_int]),
    "eagle_dev_gemm_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]),
    "eagle_dev_colgemv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_void_p, C.c_void_p, C.c_void_p]),
}

_lib = None


class EagleError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("libeaglehip error %d: %s" % (code, text))
        self.code = code
        self.text = text


def load():
    """dlopen libeaglehip.so and bind every declared symbol (no device needed for this)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError(
                "libeaglehip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C eagleeverything_amd/csrc`).  There is no CPU fallback.")
        # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7,
        # the same SONAME /opt/rocm's copy has).  If torch is going to be used in this process (device memory,
        # streams, torch.distributed over RCCL) it must be loaded FIRST so that libeaglehip.so's DT_NEEDED
        # libamdhip64.so.7 resolves to the copy torch already mapped; two HSA runtimes in one process cannot both
        # open the GPU.  Without torch (the R package case) the system ROCm runtime is used.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib
