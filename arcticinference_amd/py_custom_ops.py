"""Drop-in for arctic_inference.py_custom_ops (/root/reference/arctic_inference/py_custom_ops.py:9-54).

The reference looks for `custom_ops*.so` next to the package and `torch.ops.load_library`s it; callers
treat a False return as "use the per-layer vLLM op instead" (llama_swiftkv.py:370-371, :629-656).  Here
the op lives in libarctic_hip.so behind a C ABI, so loading means: the library is present, exports the
symbol, and a HIP device is visible.  The loader never raises, like the reference's.
"""
from __future__ import annotations

import logging
from typing import List

import torch

logger = logging.getLogger(__name__)
_loaded = None
_torch_lib = None          # keeps the torch.library fragment (and with it the registration) alive

# The reference's schema, verbatim in meaning (csrc/custom_ops/torch_bindings.cpp:5-18): caches are mutated in place,
# nothing is returned.  The implementation is registered for the CUDA dispatch key only (= HIP devices under
# PyTorch-ROCm), like the reference's `ops.impl(..., torch::kCUDA, ...)`: CPU tensors raise NotImplementedError from the
# dispatcher — there is no CPU fallback.
OP_QUALNAME = "arctic_inference::reshape_and_cache_flash_bulk"
_SCHEMA = ("reshape_and_cache_flash_bulk(Tensor keys, Tensor values, Tensor(c!)[] key_caches, Tensor(d!)[] value_caches, "
           "Tensor slot_mapping, str kv_cache_dtype, Tensor(e)[] k_scales, Tensor(f)[] v_scales, int num_heads, "
           "int head_size) -> ()")
_writers = {}              # (cache pointers, dtype string, scale pointers, heads, head size) -> ops.KvBulkWriter


def _bulk_impl(keys, values, key_caches, value_caches, slot_mapping, kv_cache_dtype, k_scales, v_scales, num_heads, head_size):
    from . import ops
    if len(key_caches) == 0 and len(value_caches) == 0 and len(k_scales) == 0 and len(v_scales) == 0:
        return                                             # kernels.cu:99-101: no layers, no-op
    key = (tuple(int(c.data_ptr()) for c in key_caches), tuple(int(c.data_ptr()) for c in value_caches), kv_cache_dtype,
           tuple(int(t.data_ptr()) for t in k_scales), tuple(int(t.data_ptr()) for t in v_scales), int(num_heads),
           int(head_size), tuple(key_caches[0].shape) if len(key_caches) else (), int(key_caches[0].stride(0)) if len(key_caches) else 0)
    w = _writers.get(key)
    if w is None:
        if len(_writers) >= 16:                            # engines come and go in tests; a model has one or two sets
            _writers.clear()
        w = _writers[key] = ops.KvBulkWriter(list(key_caches), list(value_caches), kv_cache_dtype, list(k_scales),
                                             list(v_scales), num_heads, head_size)
    w(keys, values, slot_mapping)


def _bulk_fake(keys, values, key_caches, value_caches, slot_mapping, kv_cache_dtype, k_scales, v_scales, num_heads, head_size):
    return None


def register_torch_ops() -> None:
    """Defines torch.ops.arctic_inference.reshape_and_cache_flash_bulk (once per process) with the reference's schema, a
    CUDA-key implementation over libarctic_hip.so and a fake implementation, so that Dynamo / AOT autograd trace through a
    caller (functionalised as a mutation of the cache lists) instead of breaking the graph at a ctypes call."""
    global _torch_lib
    if _torch_lib is not None:
        return
    lib = torch.library.Library("arctic_inference", "FRAGMENT")
    lib.define(_SCHEMA)
    lib.impl("reshape_and_cache_flash_bulk", _bulk_impl, "CUDA")
    torch.library.register_fake(OP_QUALNAME, _bulk_fake, lib=lib)
    _torch_lib = lib


def try_load_torch_library() -> bool:
    """True when the op is usable: libarctic_hip.so is present, exports the symbol, a HIP device is visible — and the
    torch op is registered (the reference's torch.ops.load_library does the registration as a side effect)."""
    global _loaded
    if _loaded is None:
        try:
            from . import _native
            lib = _native.lib()
            _loaded = hasattr(lib, "aic_reshape_and_cache_flash_bulk") and lib.aic_device_count() > 0
            if hasattr(lib, "aic_reshape_and_cache_flash_bulk"):
                register_torch_ops()
            if _loaded:
                logger.info("Loaded MI355X custom ops library from %s", _native.LIB_PATH)
        except Exception as e:  # missing .so, wrong arch, ...
            logger.info("Unable to load custom library: %s", e)
            _loaded = False
    return _loaded


def reshape_and_cache_flash_bulk(keys: torch.Tensor, values: torch.Tensor, key_caches: List[torch.Tensor],
                                 value_caches: List[torch.Tensor], slot_mapping: torch.Tensor, kv_cache_dtype: str,
                                 k_scales: List[torch.Tensor], v_scales: List[torch.Tensor], num_heads: int,
                                 head_size: int) -> None:
    """py_custom_ops.py:40-54: through the registered torch op, like the reference."""
    register_torch_ops()
    torch.ops.arctic_inference.reshape_and_cache_flash_bulk(keys, values, key_caches, value_caches, slot_mapping,
                                                            kv_cache_dtype, k_scales, v_scales, num_heads, head_size)
