from arcticinference_amd.py_custom_ops import reshape_and_cache_flash_bulk, try_load_torch_library  # noqa: F401
