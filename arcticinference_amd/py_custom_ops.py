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


def try_load_torch_library() -> bool:
    global _loaded
    if _loaded is None:
        try:
            from . import _native
            lib = _native.lib()
            _loaded = hasattr(lib, "aic_reshape_and_cache_flash_bulk") and lib.aic_device_count() > 0
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
    from . import ops
    ops.reshape_and_cache_flash_bulk(keys, values, key_caches, value_caches, slot_mapping, kv_cache_dtype, k_scales,
                                     v_scales, num_heads, head_size)
