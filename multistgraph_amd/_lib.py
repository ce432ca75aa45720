"""ctypes binding of libmatgcn.so (include/matgcn.h).  Fails loudly: there is no fallback.

torch is imported first on purpose: libmatgcn.so needs libamdhip64.so.7 and must share the HIP
runtime instance torch already loaded (same SONAME), so that torch's streams / device pointers are
valid inside the library.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (loads the HIP runtime the library binds to)

ABI_VERSION = 11
MAX_LAYERS = 4
MAX_HEADS = 8
MAX_EXT = 16

ADP_NONE, ADP_UNI, ADP_BI = 0, 1, 2
ADP_CODES = {"none": ADP_NONE, "unidirection": ADP_UNI, "bidirection": ADP_BI}

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libmatgcn.so")
if os.environ.get("MATGCN_LIB"):   # lab builds of the same library (build.build_variant); still HIP-only, still loud
    LIB_PATH = os.path.abspath(os.environ["MATGCN_LIB"])


class MatgcnError(RuntimeError):
    pass


class Dims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "batch", "nodes", "in_steps", "x_steps", "x_feat", "out_channels", "out_dim", "start_dim",
        "hidden", "layers", "feat_in", "embed_dim", "adj_rank", "adp_mode", "n_static", "cheb_k",
        "scale_by_g", "n_heads", "n_ts", "diag_static_mask", "gcn_off", "fnn_off")] + [
        ("head_begin", C.c_int32 * MAX_HEADS),
        ("ext_src", C.c_int32 * MAX_EXT),
    ]


class AgcnParams(C.Structure):
    _fields_ = [("weights_g", C.c_void_p), ("weights_pool", C.c_void_p), ("bias_pool", C.c_void_p)]


class LinearParams(C.Structure):
    _fields_ = [("weight", C.c_void_p), ("bias", C.c_void_p)]


class Series(C.Structure):
    """matgcn_series: the batch as label starts into the device-resident raw series"""
    _fields_ = [("series", C.c_void_p), ("series_steps", C.c_int64), ("label_start", C.c_void_p),
                ("rel_steps", C.POINTER(C.c_int32))]


class Params(C.Structure):
    _fields_ = [
        ("node_emb", C.c_void_p), ("node_vec1", C.c_void_p), ("node_vec2", C.c_void_p),
        ("static_supports", C.c_void_p), ("weight_tsg", C.c_void_p),
        ("weight_ts", C.c_void_p * MAX_HEADS), ("weights_gru", C.c_void_p),
        ("gate", AgcnParams * MAX_LAYERS), ("update", AgcnParams * MAX_LAYERS),
        ("res_gate", LinearParams * MAX_LAYERS), ("res_update", LinearParams * MAX_LAYERS),
        ("end_conv_weight", C.c_void_p), ("end_conv_bias", C.c_void_p),
    ]


class MetricScale(C.Structure):
    """matgcn_metric_scale"""
    _fields_ = [("mean", C.c_void_p), ("std", C.c_void_p), ("per_node", C.c_int32), ("mean2", C.c_void_p),
                ("std2", C.c_void_p), ("clamp_min", C.c_float), ("truth_min", C.c_float), ("min_s", C.c_float)]


METRIC_SUMS = 14
METRICS = ("MAE", "MAPE", "MSE", "RMSE", "masked_MAE", "masked_MAPE", "masked_MSE", "masked_RMSE", "R2", "EVAR")

# every symbol include/matgcn.h declares, with its argument types
_P = C.c_void_p
_SIGNATURES = {
    "matgcn_abi_version": (C.c_int, []),
    "matgcn_error_string": (C.c_char_p, [C.c_int]),
    "matgcn_prepared_bytes": (C.c_int, [C.POINTER(Dims), C.POINTER(C.c_size_t)]),
    "matgcn_workspace_bytes": (C.c_int, [C.POINTER(Dims), C.POINTER(C.c_size_t)]),
    "matgcn_supports_layout": (C.c_int, [C.POINTER(Dims), C.POINTER(C.c_int64 * 4)]),
    "matgcn_weights_layout": (C.c_int, [C.POINTER(Dims), C.c_int, C.c_int, C.POINTER(C.c_int64 * 4)]),
    "matgcn_prepare": (C.c_int, [C.POINTER(Dims), C.POINTER(Params), _P, C.c_size_t, _P, C.c_size_t, _P]),
    "matgcn_forward": (C.c_int, [C.POINTER(Dims), C.POINTER(Params), _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "matgcn_forward_series": (C.c_int, [C.POINTER(Dims), C.POINTER(Params), _P, _P, C.c_int64, _P,
                                        C.POINTER(C.c_int32), _P, _P, _P, C.c_size_t, _P]),
    "matgcn_series_violations": (C.c_int, [C.POINTER(C.c_int64), C.c_int]),
    "matgcn_fuse_heads": (C.c_int, [C.POINTER(Dims), C.POINTER(Params), _P, _P, _P, C.c_size_t, _P]),
    "matgcn_agcn_gate_fwd": (C.c_int, [C.POINTER(Dims), C.POINTER(Params), _P, C.c_int, _P, _P, _P, _P,
                                       C.c_size_t, _P]),
    "matgcn_atgru_cell_fwd": (C.c_int, [C.POINTER(Dims), C.POINTER(Params), _P, C.c_int, _P, _P, _P, _P,
                                        C.c_size_t, _P]),
    "matgcn_res_cell_fwd": (C.c_int, [C.POINTER(Dims), C.POINTER(Params), _P, C.c_int, _P, _P, _P, _P,
                                      C.c_size_t, _P]),
    "matgcn_encoder_fwd": (C.c_int, [C.POINTER(Dims), C.POINTER(Params), _P, _P, _P, _P, _P, _P,
                                     C.c_size_t, _P]),
    "matgcn_output_head": (C.c_int, [C.POINTER(Dims), C.POINTER(Params), _P, _P, _P, _P, C.c_size_t, _P]),
    "matgcn_masked_mae": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_float, C.c_float, C.c_float, C.c_float, _P, _P, _P]),
    "matgcn_masked_mae_grad": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_float, C.c_float, C.c_float, C.c_float, _P, _P, _P, _P]),
    "matgcn_metric_sums": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.POINTER(MetricScale), _P, _P, C.c_int, _P]),
    "matgcn_metric_table": (C.c_int, [_P, C.c_int, C.c_int, _P, _P]),
    "matgcn_train_bytes": (C.c_int, [C.POINTER(Dims), C.POINTER(C.c_size_t)]),
    "matgcn_forward_train": (C.c_int, [C.POINTER(Dims), C.POINTER(Params), _P, _P, C.POINTER(Series), _P, _P, _P, _P,
                                       C.c_size_t, _P, C.c_size_t, _P]),
    # matgcn_grads has the layout of matgcn_params (non-const pointers): the same ctypes struct serves both
    "matgcn_backward": (C.c_int, [C.POINTER(Dims), C.POINTER(Params), _P, _P, C.POINTER(Series), _P, _P, _P,
                                  C.POINTER(Params), _P, _P, C.c_size_t, _P, C.c_size_t, _P]),
    "matgcn_debug_gemm": (C.c_int, [_P, _P, _P, C.POINTER(C.c_int64), C.c_float, C.c_float, _P]),
    "matgcn_set_wavefront": (C.c_int, [C.c_int]),
    "matgcn_set_stream_pool": (C.c_int, [C.c_int]),
    "matgcn_set_mix_precision": (C.c_int, [C.c_int]),
    "matgcn_set_batch_split": (C.c_int, [C.c_int]),
    "matgcn_set_lazy_prepare": (C.c_int, [C.c_int]),
    "matgcn_prepare_join": (C.c_int, [_P]),
    "matgcn_profile_enable": (C.c_int, [C.c_int, C.c_int]),
    "matgcn_profile_collect": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]),
    "matgcn_profile_disable": (C.c_int, []),
}
# matgcn_profile_* kernel kinds (MATGCN_PROF_*): k_mix = k_mix<1> (per-step graph mix), k_mix_pre = k_mix<0>,
# k_gate = k_gate16, k_update = k_update16<0|1> (update [+ residual cell]), k_res_gru = k_update16<2>, k_px = k_px16
PROF_KINDS = {1: "k_mix", 2: "k_gate", 4: "k_update", 8: "k_res_gru", 16: "k_px", 32: "k_head", 64: "k_mix_pre"}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def load() -> C.CDLL:
    """Load libmatgcn.so or raise; never falls back to anything else."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MatgcnError(
            "libmatgcn.so is not built (%s). Run `python -m multistgraph_amd.build` "
            "(hipcc --offload-arch=gfx950); there is no CPU or PyTorch fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = ABI mismatch, let it propagate
        fn.restype = res
        fn.argtypes = args
    if lib.matgcn_abi_version() != ABI_VERSION:
        raise MatgcnError("libmatgcn.so ABI version %d, binding expects %d" % (lib.matgcn_abi_version(), ABI_VERSION))
    # this binding owns `prepared` for the lifetime of a HotPath and reads it only through the library (or behind
    # matgcn_prepare_join): matgcn_prepare may leave its weight streams running beside the start of the next forward
    if os.environ.get("MATGCN_LAZY_PREPARE", "1") != "0":
        lib.matgcn_set_lazy_prepare(1)
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().matgcn_error_string(status).decode()
        raise MatgcnError("%s failed: %s (status %d)" % (what, msg, status))
