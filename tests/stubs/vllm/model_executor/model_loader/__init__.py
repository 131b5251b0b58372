from vllm import ModelRegistry
from vllm.config import get_current_vllm_config, set_current_vllm_config

loaded = []     # (architecture, tp world size at construction) of every model built (tests look at it)


def get_model(*, vllm_config):
    from vllm.distributed.parallel_state import get_tp_group
    from vllm.model_executor.models.toy import ToyLlamaForCausalLM
    arch = vllm_config.model_config.hf_config.architectures[0]
    ctor = ToyLlamaForCausalLM if arch == "ToyLlamaForCausalLM" else ModelRegistry.resolve(arch)
    prev = get_current_vllm_config()
    set_current_vllm_config(vllm_config)
    try:
        model = ctor(vllm_config=vllm_config, prefix=getattr(vllm_config, "_prefix", ""))
    finally:
        set_current_vllm_config(prev)
    loaded.append((arch, get_tp_group().world_size))
    from vllm.compilation import cuda_graphs
    # (draft models of the plugin are not nn.Modules with a forward: nothing to wrap)
    return cuda_graphs.install(model, vllm_config) if hasattr(model, "forward") else model
