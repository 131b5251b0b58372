"""ctypes binding of libarctic_hip.so (include/arctic_hip.h).

There is no fallback: if the library is missing this module raises at import of `lib()`, and
compute entry points raise RuntimeError when no HIP device is visible.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_uint64, c_void_p

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libarctic_hip.so"
LIB_PATH = os.environ.get("AIC_LIB_PATH") or os.path.join(_PKG, LIB_NAME)   # AIC_LIB_PATH: A/B builds while tuning
CSRC = os.path.join(_PKG, "csrc")

AIC_OK = 0
AIC_ERR_INVALID, AIC_ERR_NO_DEVICE, AIC_ERR_HIP, AIC_ERR_NOT_FOUND, AIC_ERR_EXISTS, AIC_ERR_UNSUPPORTED = -1, -2, -3, -4, -5, -6
AIC_ERR_BUFFER_TOO_SMALL = -7
DT_F32, DT_F16, DT_BF16, DT_FP8_E4M3, DT_FP8_E5M2 = 0, 1, 2, 3, 4

_lib = None


class NativeError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libarctic_hip: {msg} (status {code})")
        self.code = code


class LstmConfig(ctypes.Structure):
    _fields_ = [("vocab_size", c_int32), ("vocab_offset", c_int32), ("input_hidden_dim", c_int32),
                ("inner_dim", c_int32), ("n_predict", c_int32), ("scale_input", c_int32),
                ("max_batch", c_int32), ("head_fp8_max_batch", c_int32)]


class LstmWeights(ctypes.Structure):
    _fields_ = [("forget_emb", c_void_p), ("proj0", c_void_p), ("proj1", c_void_p), ("cell_ln_w", c_void_p),
                ("cell_ln_b", c_void_p), ("state_ln_w", c_void_p), ("state_ln_b", c_void_p), ("head", c_void_p),
                ("head_fp8", c_void_p), ("head_fp8_scale", c_float)]


class MlpWeights(ctypes.Structure):
    _fields_ = [("num_heads", c_int32), ("emb", c_void_p * 8), ("proj", c_void_p * 8), ("ln_w", c_void_p * 8),
                ("ln_b", c_void_p * 8), ("head", c_void_p * 8)]


class MlpStack(ctypes.Structure):
    _fields_ = [("n_emb", c_int32), ("n_proj", c_int32), ("n_ln", c_int32), ("pad", c_int32)] + [
        (name, (c_void_p * 3) * 8) for name in ("emb_ln_w", "emb_ln_b", "emb_lin", "proj_ln_w", "proj_ln_b", "proj_lin",
                                                 "ln_lin", "ln_ln_w", "ln_ln_b")]


def build(force: bool = False) -> str:
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", CSRC, "-j8"], stdout=subprocess.DEVNULL)  # incremental
    return LIB_PATH


_SIGNATURES = {
    # name: (restype, [argtypes])
    "aic_last_error": (c_char_p, []),
    "aic_version": (c_int, []),
    "aic_device_count": (c_int, []),
    "aic_profile_enable": (c_int, [c_int]),
    "aic_profile_read": (c_int, [POINTER(c_double), POINTER(c_int)]),
    "aic_profile_event_overhead": (c_int, [c_void_p, c_int, POINTER(c_double), POINTER(c_double)]),
    "aic_st_create": (c_void_p, [c_int]),
    "aic_st_destroy": (None, [c_void_p]),
    "aic_st_num_seqs": (c_int, [c_void_p]),
    "aic_st_append": (c_int, [c_void_p, c_int, c_int]),
    "aic_st_extend": (c_int, [c_void_p, c_int, POINTER(c_int32), c_int]),
    "aic_st_speculate": (c_int, [c_void_p, POINTER(c_int32), c_int, c_int, c_float, c_float, c_float, c_int,
                                 POINTER(c_int32), POINTER(c_int32), POINTER(c_float), c_int, POINTER(c_float),
                                 POINTER(c_int32), c_void_p]),
    "aic_st_export": (c_int, [c_void_p, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32), POINTER(c_int32),
                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "aic_st_selfcheck": (c_int, [c_void_p]),
    "aic_sc_create": (c_void_p, [c_int]),
    "aic_sc_destroy": (None, [c_void_p]),
    "aic_sc_has_prompt": (c_int, [c_void_p, c_int64]),
    "aic_sc_cache_prompt": (c_int, [c_void_p, c_int64, c_void_p, c_int]),
    "aic_sc_cache_prompt_async": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_void_p, c_int]),
    "aic_sc_cache_prompts": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int]),
    "aic_sc_evict_prompt": (c_int, [c_void_p, c_int64]),
    "aic_sc_update_response": (c_int, [c_void_p, c_int64, c_void_p, c_int]),
    "aic_sc_warm": (c_int, [c_void_p, c_int, c_void_p]),
    "aic_sc_update_responses": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "aic_sc_speculate_batch": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p]),
    "aic_sc_speculate_batch_tree": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_void_p]),
    "aic_debug_tree_mode_stats": (c_int, [c_void_p, c_void_p]),
    "aic_debug_tree_mode_on_host": (c_int, [c_int]),
    "aic_sc_last_stats": (c_int, [c_void_p, POINTER(c_float), POINTER(c_int64), POINTER(c_int64)]),
    "aic_sc_last_timing": (c_int, [c_void_p, POINTER(c_float), POINTER(c_float)]),
    "aic_sc_global_tree": (c_void_p, [c_void_p]),
    "aic_sc_prompt_tree": (c_void_p, [c_void_p, c_int64]),
    "aic_reshape_and_cache_flash_bulk": (c_int, [c_void_p, c_void_p, POINTER(c_void_p), POINTER(c_void_p), c_void_p,
                                                 c_int, c_int, c_int, c_int, c_int, c_int64, c_int64, c_int64, c_int,
                                                 c_int, POINTER(c_void_p), POINTER(c_void_p), c_void_p]),
    "aic_ulysses_pack_pair": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "aic_ulysses_reorder_split_kv": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, POINTER(c_int32), c_void_p]),
    "aic_debug_attn_trace": (c_int, [c_void_p, c_int]),
    "aic_debug_attn_phase_trace": (c_int, [c_void_p, c_int]),
    "aic_debug_attn_layout": (c_int, [c_int, c_int]),
    "aic_debug_attn_sequential": (c_int, [c_int]),
    "aic_debug_attn_light": (c_int, [c_int]),
    "aic_debug_attn_graph": (c_int, [c_int]),
    "aic_debug_attn_long_splits": (c_int, [c_int]),
    "aic_debug_attn_long_dma": (c_int, [c_int]),
    "aic_debug_attn_graph_stats": (c_int, [POINTER(c_uint64), POINTER(c_uint64)]),
    "aic_row_gather": (c_int, [c_int, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int64), POINTER(c_int64),
                               POINTER(c_int32), c_void_p, c_int, c_int, c_void_p]),
    "aic_rejection_workspace_bytes": (c_size_t, [c_int, c_int]),
    "aic_rejection_greedy": (c_int, [c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                     c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p]),
    "aic_rejection_random": (c_int, [c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p]),
    "aic_lstm_create": (c_int, [POINTER(LstmConfig), POINTER(LstmWeights), POINTER(c_void_p)]),
    "aic_mlp_create": (c_int, [POINTER(LstmConfig), POINTER(MlpWeights), POINTER(c_void_p)]),
    "aic_mlp_create_stacked": (c_int, [POINTER(LstmConfig), POINTER(MlpWeights), POINTER(MlpStack), POINTER(c_void_p)]),
    "aic_mlp_set_embedding_rows": (c_int, [c_void_p, c_void_p]),
    "aic_lstm_destroy": (None, [c_void_p]),
    "aic_quantize_fp8_per_tensor": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "aic_lstm_padding_size": (c_int, [c_int]),
    "aic_lstm_propose": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "aic_debug_lstm_fused": (c_int, [c_int]),
    "aic_debug_lstm_cell_launches": (c_int, [c_int]),
    "aic_debug_lstm_cell_trace": (c_int, [c_void_p]),
    "aic_lstm_begin": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "aic_lstm_head": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "aic_verify_attention_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "aic_verify_attention": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p,
                                     c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                     c_int, c_float, c_void_p, c_int64, c_void_p, c_size_t, c_int, c_void_p]),
    "aic_verify_attention_layers": (c_int, [c_void_p, c_int64, c_int64, POINTER(c_void_p), POINTER(c_void_p), c_int, c_int64,
                                            c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int,
                                            c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int64, c_int64,
                                            c_void_p, c_size_t, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "aic_verify_attention_ex": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p,
                                        c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                        c_int, c_float, c_void_p, c_int64, c_void_p, c_size_t, c_int, c_void_p, c_int,
                                        c_void_p, c_int, c_void_p]),
    "aic_verify_attention_win": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p,
                                         c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                         c_int, c_float, c_void_p, c_int64, c_void_p, c_size_t, c_int, c_void_p, c_int,
                                         c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "aic_step_build": (c_int, [c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int,
                               c_int, c_int, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "aic_step_parse": (c_int, [c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p,
                               c_void_p, c_void_p]),
    "aic_ulysses_pack_qkv": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int, c_int,
                                     c_int, c_int, c_void_p]),
    "aic_ulysses_split_qkv": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "aic_ulysses_unpack_out": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES.keys())


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `make -C {CSRC}` "
                              "(or __graft_entry__.build()); there is no fallback implementation")
        # torch first: PyTorch-ROCm ships its own libamdhip64, and the library must bind to the HIP runtime torch
        # initialises (loaded the other way round the process holds two runtimes and this one sees no device: found when
        # a script touched lib() before importing torch)
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(code: int) -> int:
    """Raise on a negative status; non-negative values pass through."""
    if code < 0:
        msg = lib().aic_last_error().decode("utf-8", "replace")
        if code == AIC_ERR_NOT_FOUND or code == AIC_ERR_EXISTS:
            raise ValueError(msg)
        raise NativeError(code, msg)
    return code


def current_stream_ptr() -> int:
    import torch
    return int(torch.cuda.current_stream().cuda_stream)


def torch_dtype_code(dtype) -> int:
    import torch
    table = {torch.float32: DT_F32, torch.float16: DT_F16, torch.bfloat16: DT_BF16}
    for name, code in (("float8_e4m3fn", DT_FP8_E4M3), ("float8_e5m2", DT_FP8_E5M2)):
        if hasattr(torch, name):
            table[getattr(torch, name)] = code
    if dtype not in table:
        raise NativeError(AIC_ERR_UNSUPPORTED, f"unsupported dtype {dtype}")
    return table[dtype]


def attn_graph_stats():
    """(graph launches of aic_verify_attention_layers so far, distinct graphs instantiated)."""
    a, b = c_uint64(), c_uint64()
    check(lib().aic_debug_attn_graph_stats(ctypes.byref(a), ctypes.byref(b)))
    return a.value, b.value
