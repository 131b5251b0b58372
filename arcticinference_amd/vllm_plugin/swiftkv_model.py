"""LlamaSwiftKVForCausalLM for vLLM — the model path around the SwiftKV hot-path pieces of `arcticinference_amd/swiftkv.py`
(reference: /root/reference/arctic_inference/vllm/swiftkv/llama_swiftkv.py, class by class below).

A SwiftKV Llama computes the K/V of its LATER layers from the output of layer `num_key_value_layers - 1`, so only the
first `num_key_value_layers` layers have to see every token of a step; the rest run on the tokens that are sampled:

    first half   embed -> layers[:n_kv] on all T tokens                                    (LlamaSwiftKVPrefillRunner :219-281)
                 under Ulysses SP: all-gather hidden / residual / positions over the SP group (C7, :250-257)
                 norm_swiftkv(hidden + residual) -> kv_proj_swiftkv of every later layer, rotary on K (:262-276)
    between      K/V of all later layers into their paged caches in ONE launch (A16), attention metadata rewritten to the
                 sampled rows, the five per-token tensors gathered in ONE launch — into the decode runner's graph buffers
                 when the batch fits a captured size                                          (swiftkv_select :573-685)
    second half  layers[n_kv:] (q from q_proj_swiftkv, K/V as projected above) + final norm on the sampled rows, always in
                 full TP = SP x TP (`set_shift_parallel_mode(True)`)                        (LlamaSwiftKVDecodeRunner :283-321)
    scatter      the sampled rows' outputs back into the [T, hidden] tensor the runner indexes (:700-712)

The dense layers are vLLM's own modules (LlamaDecoderLayer, RMSNorm, the parallel linears); what this package adds is the
orchestration and the two fused launches.  MI355X notes: the later layers' K/V are projected straight into column slices
of ONE [T, Lkv * Hkv * D] buffer each (no torch.cat copies) — exactly the strided layout the bulk KV write reads — and
`SwiftKVSelector` keeps the pointer tables of the caches across steps.

vLLM is imported lazily (`swiftkv_classes()`): the package must import without it.  In the tests the classes run against
tests/stubs/vllm (names and signatures of vLLM 0.9.2); agreement with the real vLLM's internals cannot be checked here.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Set, Tuple

_CLASSES = None
_NAMES = ("LlamaSwiftKVAttention", "LlamaSwiftKVDecoderLayer", "LlamaSwiftKVPrefillRunner", "LlamaSwiftKVDecodeRunner",
          "LlamaSwiftKVModel", "LlamaSwiftKVForCausalLM")


def swiftkv_classes():
    """The model classes, built on first use (needs an importable `vllm`)."""
    global _CLASSES
    if _CLASSES is None:
        _CLASSES = _build()
    return _CLASSES


def __getattr__(name: str):
    if name in _NAMES:
        return getattr(swiftkv_classes(), name)
    raise AttributeError(name)


# checkpoint name piece -> (fused parameter name piece, shard id): q/k/v and gate/up are stored separately in HF checkpoints
# and fused in vLLM's modules; SwiftKV adds the k/v pair of the second half's projection (llama_swiftkv.py:728-737)
FUSED_SHARDS = (
    (".q_proj.", ".qkv_proj.", "q"),
    (".k_proj.", ".qkv_proj.", "k"),
    (".v_proj.", ".qkv_proj.", "v"),
    (".gate_proj.", ".gate_up_proj.", 0),
    (".up_proj.", ".gate_up_proj.", 1),
    (".k_proj_swiftkv.", ".kv_proj_swiftkv.", "k"),
    (".v_proj_swiftkv.", ".kv_proj_swiftkv.", "v"),
)


def _build():
    from types import SimpleNamespace

    import torch
    from torch import nn

    import vllm.distributed.parallel_state as parallel_state
    from vllm.attention.backends.abstract import AttentionType
    from vllm.compilation.decorators import support_torch_compile
    from vllm.forward_context import get_forward_context
    from vllm.model_executor.layers.layernorm import RMSNorm
    from vllm.model_executor.layers.linear import ColumnParallelLinear, QKVParallelLinear
    from vllm.model_executor.layers.logits_processor import LogitsProcessor
    from vllm.model_executor.layers.vocab_parallel_embedding import (DEFAULT_VOCAB_PADDING_SIZE, ParallelLMHead,
                                                                     VocabParallelEmbedding)
    from vllm.model_executor.model_loader.weight_utils import default_weight_loader, maybe_remap_kv_scale_name
    from vllm.model_executor.models.llama import LlamaAttention, LlamaDecoderLayer, LlamaMLP
    from vllm.model_executor.models.utils import AutoWeightsLoader, maybe_prefix

    from ..swiftkv import SwiftKVSelector, sp_all_gather, swiftkv_select
    from . import model_runner as runner

    def step_metadata():
        """The step's attention metadata (one object shared by every layer in V1), or None while profiling / capturing."""
        meta = get_forward_context().attn_metadata
        if meta is None:
            return None
        if isinstance(meta, dict):
            first = next(iter(meta.values()))
            assert all(m is first for m in meta.values()), "SwiftKV expects one attention metadata object for all layers"
            return first
        return meta

    class LlamaSwiftKVAttention(LlamaAttention):
        """Attention of a second-half layer (:67-137): the query comes from its own projection of the layer input, K and V
        are handed in (projected by the first half from the SwiftKV hidden state and already rotated)."""

        def __init__(self, config, hidden_size: int, num_heads: int, num_kv_heads: int, rope_theta: float = 10000,
                     rope_scaling=None, max_position_embeddings: int = 8192, quant_config=None, bias: bool = False,
                     bias_o_proj: bool = False, cache_config=None, prefix: str = "", attn_type: str = AttentionType.DECODER):
            super().__init__(config=config, hidden_size=hidden_size, num_heads=num_heads, num_kv_heads=num_kv_heads,
                             rope_theta=rope_theta, rope_scaling=rope_scaling,
                             max_position_embeddings=max_position_embeddings, quant_config=quant_config, bias=bias,
                             bias_o_proj=bias_o_proj, cache_config=cache_config, prefix=prefix, attn_type=attn_type)
            self.q_proj_swiftkv = ColumnParallelLinear(input_size=hidden_size, output_size=self.total_num_heads * self.head_dim,
                                                       bias=bias, gather_output=False, quant_config=quant_config,
                                                       prefix=f"{prefix}.q_proj_swiftkv")
            # a QKV projection with no query heads: the K / V pair of THIS layer, applied by the first half
            self.kv_proj_swiftkv = QKVParallelLinear(hidden_size=hidden_size, head_size=self.head_dim, total_num_heads=0,
                                                     total_num_kv_heads=self.total_num_kv_heads, bias=bias,
                                                     quant_config=quant_config, prefix=f"{prefix}.kv_proj_swiftkv")

        def forward(self, positions, hidden_states, k, v):
            q, _ = self.q_proj_swiftkv(hidden_states)
            q, _ = self.rotary_emb(positions, q, torch.empty_like(k))      # K was rotated where it was projected
            out, _ = self.o_proj(self.attn(q, k, v))
            return out

    class LlamaSwiftKVDecoderLayer(nn.Module):
        """A second-half layer (:140-216): Llama's layer with LlamaSwiftKVAttention."""

        def __init__(self, config, cache_config=None, quant_config=None, prefix: str = ""):
            super().__init__()
            self.hidden_size = config.hidden_size
            rope_scaling = getattr(config, "rope_scaling", None)
            if rope_scaling is not None and getattr(config, "original_max_position_embeddings", None):
                rope_scaling["original_max_position_embeddings"] = config.original_max_position_embeddings
            self.self_attn = LlamaSwiftKVAttention(
                config=config, hidden_size=self.hidden_size, num_heads=config.num_attention_heads,
                num_kv_heads=getattr(config, "num_key_value_heads", config.num_attention_heads),
                rope_theta=getattr(config, "rope_theta", 10000), rope_scaling=rope_scaling,
                max_position_embeddings=getattr(config, "max_position_embeddings", 8192), quant_config=quant_config,
                bias=getattr(config, "attention_bias", False) or getattr(config, "bias", False), cache_config=cache_config,
                prefix=f"{prefix}.self_attn")
            self.mlp = LlamaMLP(hidden_size=self.hidden_size, intermediate_size=config.intermediate_size,
                                hidden_act=config.hidden_act, quant_config=quant_config,
                                bias=getattr(config, "mlp_bias", False), prefix=f"{prefix}.mlp")
            self.input_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
            self.post_attention_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)

        def forward(self, positions, hidden_states, k_states, v_states, residual):
            if residual is None:
                residual = hidden_states
                hidden_states = self.input_layernorm(hidden_states)
            else:
                hidden_states, residual = self.input_layernorm(hidden_states, residual)
            hidden_states = self.self_attn(positions=positions, hidden_states=hidden_states, k=k_states, v=v_states)
            hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
            return self.mlp(hidden_states), residual

    @support_torch_compile
    class LlamaSwiftKVPrefillRunner(nn.Module):
        """First half (:219-281), compiled like the reference's (both runners carry @support_torch_compile there).  Holds the
        model in a list so that nn.Module does not register it a second time.  Everything in forward() is traceable by
        Dynamo with fullgraph=True (tests/test_vllm_swiftkv_gpu.py compiles both runners): attention is one opaque custom op
        (custom_ops.py), and the switch to the SP x TP group around the second half's projections is the reference's plain
        pair of global stores (:258-262, :279-280) — set_shift_parallel_mode()'s generator-based context manager would be a
        graph break."""

        def __init__(self, *, vllm_config, model, prefix: str = ""):
            super().__init__()
            self.config = vllm_config.model_config.hf_config
            self._model = [model]

        @property
        def model(self):
            return self._model[0]

        def forward(self, input_ids, positions):
            m, n_kv = self.model, self.config.num_key_value_layers
            hidden_states, residual = m.get_input_embeddings(input_ids), None
            for layer in m.layers[:n_kv]:
                hidden_states, residual = layer(positions, hidden_states, residual)
            sp = getattr(parallel_state, "_SP", None)
            if sp is not None and sp.world_size > 1 and not runner.is_shift_parallel_mode():
                # C7: the second half runs in full TP, every rank needs every token
                hidden_states, residual, positions = sp_all_gather((hidden_states, residual, positions), sp.world_size,
                                                                    sp.device_group)
            later = m.layers[n_kv:]
            # the projections below belong to second-half layers (SP x TP shards): TP group := SP_TP while they run
            saved_mode, saved_tp = runner.SP_TP_MODE, parallel_state._TP
            if getattr(parallel_state, "_SP_TP", None) is not None:
                if not runner.is_shift_parallel_mode():
                    parallel_state._ORIG_TP = parallel_state._TP
                runner.SP_TP_MODE = True
                parallel_state._TP = parallel_state._SP_TP
            swiftkv_hidden = m.norm_swiftkv(hidden_states + residual)
            T = hidden_states.shape[0]
            kv_size = later[0].self_attn.kv_size
            # every later layer's K / V go into its column slice of one buffer each: the layout the bulk KV write reads
            k_states = torch.empty(T, len(later) * kv_size, dtype=hidden_states.dtype, device=hidden_states.device)
            v_states = torch.empty_like(k_states)
            rotary = m.layers[0].self_attn.rotary_emb
            q_scratch = torch.empty(T, kv_size, dtype=hidden_states.dtype, device=hidden_states.device)
            for i, layer in enumerate(later):
                kv, _ = layer.self_attn.kv_proj_swiftkv(swiftkv_hidden)
                k, v = kv.split([kv_size, kv_size], dim=-1)
                _, k = rotary(positions, q_scratch, k)
                k_states[:, i * kv_size:(i + 1) * kv_size] = k
                v_states[:, i * kv_size:(i + 1) * kv_size] = v
            runner.SP_TP_MODE = saved_mode
            parallel_state._TP = saved_tp
            return hidden_states, residual, positions, k_states, v_states

    @support_torch_compile
    class LlamaSwiftKVDecodeRunner(nn.Module):
        """Second half (:283-321): the later layers on the surviving tokens, then the final norm."""

        def __init__(self, *, vllm_config, model, prefix: str = ""):
            super().__init__()
            self.config = vllm_config.model_config.hf_config
            self._model = [model]

        @property
        def model(self):
            return self._model[0]

        def forward(self, hidden_states, residual, positions, k_states, v_states):
            later = self.model.layers[self.config.num_key_value_layers:]
            kv_size = k_states.shape[-1] // len(later)
            for i, layer in enumerate(later):
                cols = slice(i * kv_size, (i + 1) * kv_size)
                hidden_states, residual = layer(positions, hidden_states, k_states[:, cols], v_states[:, cols], residual)
            hidden_states, _ = self.model.norm(hidden_states, residual)
            return hidden_states

    class LlamaSwiftKVModel(nn.Module):
        """(:324-725)"""

        def __init__(self, *, vllm_config, prefix: str = ""):
            super().__init__()
            config = vllm_config.model_config.hf_config
            self.vllm_config, self.config = vllm_config, config
            self.quant_config = getattr(vllm_config, "quant_config", None)
            self.vocab_size = self.org_vocab_size = config.vocab_size
            self.embed_tokens = VocabParallelEmbedding(self.vocab_size, config.hidden_size, org_num_embeddings=config.vocab_size,
                                                       quant_config=self.quant_config)
            n_kv = config.num_key_value_layers
            self.layers = nn.ModuleList([
                LlamaDecoderLayer(config=config, cache_config=vllm_config.cache_config, quant_config=self.quant_config,
                                  prefix=f"{prefix}.layers.{i}") for i in range(n_kv)])
            # the second half is always built (and its weights sharded) over SP x TP ranks (:352-365)
            with runner.set_shift_parallel_mode(True):
                self.layers.extend([
                    LlamaSwiftKVDecoderLayer(config=config, cache_config=vllm_config.cache_config,
                                             quant_config=self.quant_config, prefix=f"{prefix}.layers.{i}")
                    for i in range(n_kv, config.num_hidden_layers)])
                self.norm_swiftkv = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
                self.norm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
            for p in self.layers[n_kv:].parameters():
                p.shift_parallel_mode = True
            self.prefill_runner = LlamaSwiftKVPrefillRunner(vllm_config=vllm_config, model=self)
            self.decode_runner = LlamaSwiftKVDecodeRunner(vllm_config=vllm_config, model=self)
            sizes = vllm_config.compilation_config.cudagraph_capture_sizes
            self.cuda_graph_max_batch_size = max(sizes) if sizes else 0
            self._selector: Optional[SwiftKVSelector] = None

        def get_input_embeddings(self, input_ids):
            return self.embed_tokens(input_ids)

        def selector(self, like) -> SwiftKVSelector:
            """Graph input buffers + cache pointer tables, made with the first tensor that tells dtype and device (:392-413)."""
            if self._selector is None:
                attn = self.layers[-1].self_attn.attn
                n_later = self.config.num_hidden_layers - self.config.num_key_value_layers
                self._selector = SwiftKVSelector(self.config.hidden_size, n_later, attn.num_kv_heads, attn.head_size, like.dtype,
                                                 like.device, cuda_graph_max_batch_size=self.cuda_graph_max_batch_size,
                                                 pad_for_cudagraph=self.vllm_config.pad_for_cudagraph)
                self.decode_runner.inputs = self._selector.inputs
            return self._selector

        def swiftkv_select(self, hidden_states, residual, positions, k_states, v_states):
            """Write the later layers' K/V, keep the sampled rows (:573-685; FlashAttention-layout metadata)."""
            ctx = get_forward_context()
            later = self.layers[self.config.num_key_value_layers:]
            attns = [layer.self_attn.attn for layer in later]
            caches = [a.kv_cache[ctx.virtual_engine] for a in attns]
            return swiftkv_select(self.selector(hidden_states), hidden_states, residual, positions, k_states, v_states,
                                  step_metadata(), caches, getattr(attns[-1], "kv_cache_dtype", "auto"),
                                  [a._k_scale for a in attns], [a._v_scale for a in attns])

        def forward(self, input_ids, positions):
            hidden_states, residual, positions, k_states, v_states = self.prefill_runner(input_ids, positions)
            all_rows = hidden_states
            hidden_states, residual, positions, k_states, v_states = self.swiftkv_select(hidden_states, residual, positions,
                                                                                         k_states, v_states)
            with runner.set_shift_parallel_mode(True):
                hidden_states = self.decode_runner(hidden_states, residual, positions, k_states, v_states)
            meta = step_metadata()
            if meta is not None:
                rows = meta.swiftkv_logits_indices
                all_rows[rows] = hidden_states[:rows.numel()]      # graph-padded rows past the batch are dropped
            return all_rows

        def load_weights(self, weights: Iterable[Tuple[str, torch.Tensor]]) -> Set[str]:
            """Checkpoint names onto the fused parameters (FUSED_SHARDS); second-half parameters are loaded with the TP group
            they were built under (:726-791)."""
            params = dict(self.named_parameters())
            loaded: Set[str] = set()

            def put(param, *args):
                loader = getattr(param, "weight_loader", default_weight_loader)
                with runner.set_shift_parallel_mode(getattr(param, "shift_parallel_mode", None)):
                    loader(param, *args)

            for name, tensor in weights:
                if "rotary_emb.inv_freq" in name or "rotary_emb.cos_cached" in name or "rotary_emb.sin_cached" in name:
                    continue
                scale_name = self.quant_config.get_cache_scale(name) if self.quant_config is not None else None
                if scale_name:
                    put(params[scale_name], tensor if tensor.dim() == 0 else tensor[0])
                    loaded.add(scale_name)
                    continue
                if "scale" in name:
                    name = maybe_remap_kv_scale_name(name, params)
                    if name is None:
                        continue
                for piece, fused, shard in FUSED_SHARDS:
                    if piece in name:
                        name = name.replace(piece, fused)
                        if name.endswith(".bias") and name not in params:
                            break
                        put(params[name], tensor, shard)
                        loaded.add(name)
                        break
                else:
                    if name.endswith(".bias") and name not in params:
                        continue
                    put(params[name], tensor)
                    loaded.add(name)
            return loaded

    class LlamaSwiftKVForCausalLM(nn.Module):
        """(:794-868)"""
        packed_modules_mapping = {"qkv_proj": ["q_proj", "k_proj", "v_proj"], "gate_up_proj": ["gate_proj", "up_proj"],
                                  "kv_proj_swiftkv": ["k_proj_swiftkv", "v_proj_swiftkv"]}

        def __init__(self, *, vllm_config, prefix: str = ""):
            super().__init__()
            config = vllm_config.model_config.hf_config
            self.config = config
            self.model = LlamaSwiftKVModel(vllm_config=vllm_config, prefix=maybe_prefix(prefix, "model"))
            self.unpadded_vocab_size = config.vocab_size
            self.lm_head = ParallelLMHead(self.unpadded_vocab_size, config.hidden_size, org_num_embeddings=config.vocab_size,
                                          padding_size=DEFAULT_VOCAB_PADDING_SIZE,
                                          quant_config=getattr(vllm_config, "quant_config", None),
                                          prefix=maybe_prefix(prefix, "lm_head"))
            if getattr(config, "tie_word_embeddings", False):
                self.lm_head = self.lm_head.tie_weights(self.model.embed_tokens)
            self.logits_processor = LogitsProcessor(self.unpadded_vocab_size, config.vocab_size,
                                                    getattr(config, "logit_scale", 1.0))

        def get_input_embeddings(self, input_ids):
            return self.model.get_input_embeddings(input_ids)

        def forward(self, input_ids, positions, intermediate_tensors=None, inputs_embeds=None):
            assert intermediate_tensors is None and inputs_embeds is None, "SwiftKV runs without pipeline parallelism / embeds"
            return self.model(input_ids, positions)

        def compute_logits(self, hidden_states, sampling_metadata=None):
            return self.logits_processor(self.lm_head, hidden_states, sampling_metadata)

        def load_weights(self, weights: Iterable[Tuple[str, torch.Tensor]]) -> Set[str]:
            skip = ["lm_head."] if getattr(self.config, "tie_word_embeddings", False) else None
            return AutoWeightsLoader(self, skip_prefixes=skip).load_weights(weights)

    return SimpleNamespace(LlamaSwiftKVAttention=LlamaSwiftKVAttention, LlamaSwiftKVDecoderLayer=LlamaSwiftKVDecoderLayer,
                           LlamaSwiftKVPrefillRunner=LlamaSwiftKVPrefillRunner, LlamaSwiftKVDecodeRunner=LlamaSwiftKVDecodeRunner,
                           LlamaSwiftKVModel=LlamaSwiftKVModel, LlamaSwiftKVForCausalLM=LlamaSwiftKVForCausalLM)
