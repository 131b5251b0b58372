def has_kv_transfer_group() -> bool:
    return False


def get_kv_transfer_group():
    raise RuntimeError("no KV transfer group in the stand-in")
