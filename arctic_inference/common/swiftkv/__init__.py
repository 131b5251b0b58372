from arcticinference_amd.swiftkv_config import LlamaSwiftKVConfig  # noqa: F401
