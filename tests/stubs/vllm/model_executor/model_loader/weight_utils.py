"""Weight-loading helpers of the stand-in (names and signatures of vllm.model_executor.model_loader.weight_utils)."""
import torch


def default_weight_loader(param: torch.Tensor, loaded_weight: torch.Tensor) -> None:
    assert param.shape == loaded_weight.shape, (tuple(param.shape), tuple(loaded_weight.shape))
    param.data.copy_(loaded_weight.to(param.dtype))


def maybe_remap_kv_scale_name(name: str, params_dict: dict):
    """vLLM maps checkpoint names of fp8 kv scales onto `attn.k_scale` / `attn.v_scale`; names it cannot place drop out."""
    for old, new in ((".k_proj.k_scale", ".attn.k_scale"), (".v_proj.v_scale", ".attn.v_scale")):
        if name.endswith(old):
            name = name[:-len(old)] + new
    return name if name in params_dict else None
