"""Llama building blocks of the stand-in: vLLM's class names, constructor signatures and forward conventions
(LlamaMLP, LlamaAttention, LlamaDecoderLayer), plain torch inside."""
from typing import Any, Optional

import torch

from vllm.attention.backends.abstract import AttentionType
from vllm.attention.layer import Attention
from vllm.distributed.parallel_state import get_tp_group
from vllm.model_executor.layers.activation import SiluAndMul
from vllm.model_executor.layers.layernorm import RMSNorm
from vllm.model_executor.layers.linear import MergedColumnParallelLinear, QKVParallelLinear, RowParallelLinear
from vllm.model_executor.layers.rotary_embedding import get_rope


class LlamaMLP(torch.nn.Module):
    def __init__(self, hidden_size: int, intermediate_size: int, hidden_act: str, quant_config=None, bias: bool = False,
                 prefix: str = "", reduce_results: bool = True):
        super().__init__()
        assert hidden_act == "silu"
        self.gate_up_proj = MergedColumnParallelLinear(hidden_size, [intermediate_size] * 2, bias=bias,
                                                       prefix=f"{prefix}.gate_up_proj")
        self.down_proj = RowParallelLinear(intermediate_size, hidden_size, bias=bias, prefix=f"{prefix}.down_proj")
        self.act_fn = SiluAndMul()

    def forward(self, x):
        x, _ = self.gate_up_proj(x)
        x, _ = self.down_proj(self.act_fn(x))
        return x


class LlamaAttention(torch.nn.Module):
    def __init__(self, config, hidden_size: int, num_heads: int, num_kv_heads: int, rope_theta: float = 10000,
                 rope_scaling: Optional[dict] = None, max_position_embeddings: int = 8192, quant_config=None,
                 bias: bool = False, bias_o_proj: bool = False, cache_config=None, prefix: str = "",
                 attn_type: str = AttentionType.DECODER):
        super().__init__()
        tp_size = get_tp_group().world_size
        self.hidden_size = hidden_size
        self.total_num_heads = num_heads
        self.num_heads = num_heads // tp_size
        self.total_num_kv_heads = num_kv_heads
        self.num_kv_heads = max(1, num_kv_heads // tp_size)
        self.head_dim = getattr(config, "head_dim", None) or hidden_size // num_heads
        self.q_size, self.kv_size = self.num_heads * self.head_dim, self.num_kv_heads * self.head_dim
        self.scaling = self.head_dim ** -0.5
        self.qkv_proj = QKVParallelLinear(hidden_size, self.head_dim, self.total_num_heads, self.total_num_kv_heads, bias=bias,
                                          prefix=f"{prefix}.qkv_proj")
        self.o_proj = RowParallelLinear(self.total_num_heads * self.head_dim, hidden_size, bias=bias_o_proj,
                                        prefix=f"{prefix}.o_proj")
        self.rotary_emb = get_rope(self.head_dim, rotary_dim=self.head_dim, max_position=max_position_embeddings,
                                   base=rope_theta, rope_scaling=rope_scaling)
        self.attn = Attention(self.num_heads, self.head_dim, self.scaling, num_kv_heads=self.num_kv_heads,
                              cache_config=cache_config, quant_config=quant_config, prefix=f"{prefix}.attn",
                              attn_type=attn_type)

    def forward(self, positions, hidden_states):
        qkv, _ = self.qkv_proj(hidden_states)
        q, k, v = qkv.split([self.q_size, self.kv_size, self.kv_size], dim=-1)
        q, k = self.rotary_emb(positions, q, k)
        out, _ = self.o_proj(self.attn(q, k, v))
        return out


class LlamaDecoderLayer(torch.nn.Module):
    def __init__(self, config, cache_config=None, quant_config=None, prefix: str = ""):
        super().__init__()
        self.hidden_size = config.hidden_size
        self.self_attn = LlamaAttention(
            config=config, hidden_size=self.hidden_size, num_heads=config.num_attention_heads,
            num_kv_heads=getattr(config, "num_key_value_heads", config.num_attention_heads),
            rope_theta=getattr(config, "rope_theta", 10000), rope_scaling=getattr(config, "rope_scaling", None),
            max_position_embeddings=getattr(config, "max_position_embeddings", 8192), quant_config=quant_config,
            bias=getattr(config, "attention_bias", False), cache_config=cache_config, prefix=f"{prefix}.self_attn")
        self.mlp = LlamaMLP(self.hidden_size, config.intermediate_size, config.hidden_act, quant_config=quant_config,
                            bias=getattr(config, "mlp_bias", False), prefix=f"{prefix}.mlp")
        self.input_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.post_attention_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)

    def forward(self, positions, hidden_states, residual):
        if residual is None:
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        hidden_states = self.self_attn(positions=positions, hidden_states=hidden_states)
        hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
        return self.mlp(hidden_states), residual
