"""CPU: the SwiftKV model path's registration, construction and checkpoint-name handling through the stand-in for vLLM
(the forward needs the GPU: tests/test_vllm_swiftkv_gpu.py)."""
import pytest
import torch

import vllm_harness as H
from test_vllm_swiftkv_gpu import HID, HKV, HQ, D, INTER, L, N_KV, V, _checkpoint, _hf


def test_swiftkv_registration_construction_and_weight_names(stub_vllm):
    H.load_plugin()
    from transformers import AutoConfig
    from vllm import ModelRegistry
    from vllm.config import (CacheConfig, CompilationConfig, DeviceConfig, ModelConfig, ParallelConfig, SchedulerConfig,
                             VllmConfig, set_current_vllm_config)
    from vllm.v1.worker.gpu_model_runner import GPUModelRunner
    from arctic_inference.common.swiftkv import LlamaSwiftKVConfig
    import arctic_inference.vllm.swiftkv as alias
    # plugins.py:86-98: HF config type and model name
    assert type(AutoConfig.for_model("llama_swiftkv", num_hidden_layers=6)) is LlamaSwiftKVConfig
    assert AutoConfig.for_model("llama_swiftkv", num_hidden_layers=6).num_key_value_layers == 6       # default: every layer
    assert ModelRegistry.models["LlamaSwiftKVForCausalLM"] == "arctic_inference.vllm.swiftkv:LlamaSwiftKVForCausalLM"
    cls = ModelRegistry.resolve("LlamaSwiftKVForCausalLM")
    assert cls is alias.LlamaSwiftKVForCausalLM and cls.packed_modules_mapping["kv_proj_swiftkv"] == ["k_proj_swiftkv", "v_proj_swiftkv"]

    cfg = VllmConfig(model_config=ModelConfig(hf_config=_hf(), max_model_len=400, dtype=torch.float32),
                     parallel_config=ParallelConfig(), scheduler_config=SchedulerConfig(max_num_seqs=8),
                     cache_config=CacheConfig(block_size=16),
                     compilation_config=CompilationConfig(level=0, cudagraph_capture_sizes=(8, 4, 2, 1)),
                     device_config=DeviceConfig("cpu"))
    H.init_single_process_groups(cfg)
    runner = GPUModelRunner(cfg, torch.device("cpu"))
    set_current_vllm_config(cfg)
    runner.load_model()
    m = runner.model
    core = m.model
    assert type(m) is cls and len(core.layers) == L and core.cuda_graph_max_batch_size == 8
    first, later = core.layers[:N_KV], core.layers[N_KV:]
    assert all(type(x).__name__ == "LlamaDecoderLayer" for x in first)
    assert all(type(x).__name__ == "LlamaSwiftKVDecoderLayer" for x in later)
    a = later[0].self_attn
    assert a.q_proj_swiftkv.weight.shape == (HQ * D, HID) and a.kv_proj_swiftkv.weight.shape == (2 * HKV * D, HID)
    assert all(getattr(p, "shift_parallel_mode", False) for p in later.parameters())
    assert not any(getattr(p, "shift_parallel_mode", False) for p in first.parameters())
    # the runners hold the model without registering it a second time
    assert core.prefill_runner.model is core and not list(core.prefill_runner.parameters())

    ck = {k: v.float() for k, v in _checkpoint().items()}
    loaded = m.load_weights(ck.items())
    assert "model.layers.0.self_attn.qkv_proj.weight" in loaded and "model.norm_swiftkv.weight" in loaded
    assert not any(".q_proj." in n or ".k_proj_swiftkv." in n or "rotary" in n for n in loaded)
    p = "model.layers.%d.self_attn." % N_KV
    kv = a.kv_proj_swiftkv.weight
    assert torch.equal(kv[:HKV * D], ck[p + "k_proj_swiftkv.weight"]) and torch.equal(kv[HKV * D:], ck[p + "v_proj_swiftkv.weight"])
    gu = first[1].mlp.gate_up_proj.weight
    assert torch.equal(gu[:INTER], ck["model.layers.1.mlp.gate_proj.weight"]) and torch.equal(gu[INTER:], ck["model.layers.1.mlp.up_proj.weight"])
    assert torch.equal(m.lm_head.weight, ck["lm_head.weight"]) and m.compute_logits(torch.zeros(2, HID)).shape == (2, V)
    # a missing parameter name is an error, not a silent skip
    with pytest.raises(KeyError):
        m.load_weights([("model.layers.0.self_attn.no_such.weight", torch.zeros(1))])
