"""A Llama-shaped toy target model for the stand-in (seeded weights, no norm / MLP): embedding, L x (qkv projection ->
Attention -> output projection, residual), LM head.  Tensor-parallel like vLLM's Llama: q/k/v heads and the output
projection's input are split over get_tp_group(); the output projection all-reduces.  Because the weights are drawn
from the seed for the FULL model and then sliced, every parallel layout of the same config computes the same function
— which is what the shift-parallel tests rely on."""
import torch

from vllm.attention.layer import Attention
from vllm.distributed.parallel_state import get_tp_group


class ToyLayer(torch.nn.Module):
    def __init__(self, hf, index: int, dtype, device, gen, prefix: str):
        super().__init__()
        tp = get_tp_group()
        H, D = hf.hidden_size, hf.head_dim
        hq, hkv = hf.num_attention_heads, hf.num_key_value_heads
        std = 0.5 / H ** 0.5
        wq = torch.randn(hq * D, H, generator=gen) * std
        wk = torch.randn(hkv * D, H, generator=gen) * std
        wv = torch.randn(hkv * D, H, generator=gen) * std
        wo = torch.randn(H, hq * D, generator=gen) * std
        ways, r = tp.world_size, tp.rank_in_group
        self.hq, self.hkv = hq // ways, max(1, hkv // ways)
        kv_rank = r if hkv >= ways else r // (ways // hkv)   # (in units of self.hkv heads) replicated when there are fewer
        sl = lambda w, heads, idx: w.view(-1, D, H)[idx * heads:(idx + 1) * heads].reshape(heads * D, H)
        self.wq = sl(wq, self.hq, r).to(dtype).to(device)
        self.wk = sl(wk, self.hkv, kv_rank).to(dtype).to(device)
        self.wv = sl(wv, self.hkv, kv_rank).to(dtype).to(device)
        self.wo = wo.view(H, hq, D)[:, r * self.hq:(r + 1) * self.hq].reshape(H, self.hq * D).to(dtype).to(device)
        self.tp = tp
        # gpt-oss-like options of the config: a sliding window on every other layer, learned sinks on all of them
        window = getattr(hf, "sliding_window", None) if index % 2 == 1 else None
        sinks = None
        if getattr(hf, "attention_sinks", False):
            all_sinks = torch.randn(hq, generator=gen) * 2
            sinks = all_sinks[r * self.hq:(r + 1) * self.hq].to(dtype).to(device)     # sharded over TP like vLLM's parameter
        self.attn = Attention(self.hq, D, D ** -0.5, num_kv_heads=self.hkv, prefix=f"{prefix}layers.{index}.attn",
                              per_layer_sliding_window=window, **({"sinks": sinks} if sinks is not None else {}))
        self.attn._k_scale = self.attn._k_scale.to(device)      # vLLM keeps the cache scales on the layer's device
        self.attn._v_scale = self.attn._v_scale.to(device)

    def forward(self, x):
        q, k, v = x @ self.wq.T, x @ self.wk.T, x @ self.wv.T
        a = self.attn(q, k, v)
        o = a @ self.wo.T
        if self.tp.world_size > 1:
            o = self.tp.all_reduce(o.float()).to(x.dtype)
        return x + o


class ToyLlamaModel(torch.nn.Module):
    def __init__(self, hf, dtype, device, prefix=""):
        super().__init__()
        gen = torch.Generator().manual_seed(hf.seed)
        self.embed = (torch.randn(hf.vocab_size, hf.hidden_size, generator=gen) * 0.5).to(dtype).to(device)
        self.layers = torch.nn.ModuleList([ToyLayer(hf, i, dtype, device, gen, prefix) for i in range(hf.num_hidden_layers)])
        self.lm_head = (torch.randn(hf.vocab_size, hf.hidden_size, generator=gen) * 0.5).to(dtype).to(device)
        self.anchor = torch.nn.Parameter(torch.zeros(1, device=device), requires_grad=False)


class ToyLlamaForCausalLM(torch.nn.Module):
    def __init__(self, *, vllm_config, prefix: str = ""):
        super().__init__()
        hf = vllm_config.model_config.hf_config
        dtype = vllm_config.model_config.dtype or torch.float32
        self.model = ToyLlamaModel(hf, dtype, vllm_config.device_config.device, prefix)
        self.logit_hook = None      # tests may replace the logits (a synthetic target that follows a known stream)

    def forward(self, input_ids=None, positions=None, intermediate_tensors=None, inputs_embeds=None):
        x = self.model.embed[input_ids]
        for layer in self.model.layers:
            x = layer(x)
        return x

    def compute_logits(self, hidden_states, sampling_metadata=None):
        logits = (hidden_states @ self.model.lm_head.T).float()
        if self.logit_hook is not None:
            logits = self.logit_hook(hidden_states, logits)
        return logits
