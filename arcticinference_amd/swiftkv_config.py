"""HF config of a SwiftKV Llama (`model_type = "llama_swiftkv"`,
/root/reference/arctic_inference/common/swiftkv/configs.py:21-39): a LlamaConfig plus `num_key_value_layers`, the number
of leading layers that compute their own keys and values (all of them when not given)."""
from __future__ import annotations

from typing import Optional

from transformers import LlamaConfig


class LlamaSwiftKVConfig(LlamaConfig):
    model_type = "llama_swiftkv"

    def __init__(self, num_key_value_layers: Optional[int] = None, **kwargs):
        super().__init__(**kwargs)
        self.num_key_value_layers = num_key_value_layers or self.num_hidden_layers
