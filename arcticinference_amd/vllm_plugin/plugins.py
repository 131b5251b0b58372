"""Entry point `arctic_inference.vllm.plugins:arctic_inference_plugin` (vLLM entry-point group
"vllm.general_plugins"; reference: /root/reference/arctic_inference/vllm/plugins.py:66-126).

Deliberate deviation from the reference: its platform gate returns early unless `is_cuda()`
(plugins.py:74-77), which disables the plugin on ROCm.  This build is for MI355X, so the gate accepts the
ROCm platform (and CUDA-like platforms are refused: the native library is gfx950 only)."""
from __future__ import annotations

import logging
import os

logger = logging.getLogger(__name__)

COMPATIBLE_VLLM_VERSION = "0.9.2"  # pyproject.toml:41-43 of the reference


def arctic_inference_plugin() -> None:
    try:
        import vllm
    except ImportError:
        logger.warning("ArcticInference (MI355X build): vLLM is not installed. Ignoring plugin!")
        return
    if vllm.__version__ != COMPATIBLE_VLLM_VERSION and not vllm.__version__.startswith("0.1.dev"):
        logger.warning("ArcticInference requires vllm==%s but found vllm==%s. Ignoring plugin!",
                       COMPATIBLE_VLLM_VERSION, vllm.__version__)
        return
    from vllm.platforms import current_platform
    if not current_platform.is_rocm():
        logger.warning("ArcticInference (MI355X build) requires the ROCm platform. Ignoring plugin!")
        return
    if os.getenv("VLLM_USE_V1") == "0":
        logger.warning("ArcticInference only supports vLLM V1, but detected V0 engine. Ignoring plugin!")
        return
    # Nothing here may touch the HIP runtime: vLLM loads general plugins in the API-server and EngineCore processes too,
    # which later FORK the workers, and a child of a HIP-initialised parent cannot use the GPU (the reference defers
    # everything GPU-side to WorkerBasePatch for the same reason, plugins.py:54-63).  Only the file's presence is checked
    # (there is no CPU fallback to degrade to); the device probe runs in the worker (model_runner.WorkerBasePatch).
    from .. import _native
    if not os.path.exists(_native.LIB_PATH):
        raise ImportError(f"{_native.LIB_PATH} is missing: build it with `make -C {_native.CSRC}`; there is no fallback")

    # SwiftKV: HF config type + model class (plugins.py:86-98); the class is resolved from its "module:Class" string on
    # first use, so vLLM's model zoo is not imported here
    from transformers import AutoConfig
    from ..swiftkv_config import LlamaSwiftKVConfig
    try:
        AutoConfig.register("llama_swiftkv", LlamaSwiftKVConfig)
    except ValueError:          # already registered (the plugin is loaded once per process, tests load it repeatedly)
        pass
    from vllm import ModelRegistry
    ModelRegistry.register_model("LlamaSwiftKVForCausalLM", "arctic_inference.vllm.swiftkv:LlamaSwiftKVForCausalLM")
    ModelRegistry.register_model("ArcticMLPSpeculatorPreTrainedModel",
                                 "arcticinference_amd.vllm_plugin.model_runner:ArcticMLPSpeculatorForVllm")
    ModelRegistry.register_model("ArcticLSTMSpeculatorPreTrainedModel",
                                 "arcticinference_amd.vllm_plugin.model_runner:ArcticLSTMSpeculatorForVllm")
    ModelRegistry.register_model("MLPVariantSpeculatorPreTrainedModel",
                                 "arcticinference_amd.vllm_plugin.model_runner:ArcticLSTMSpeculatorForVllm")

    from .args import build_args_patches
    from .config import build_config_patches
    from .custom_ops import build_compilation_patches
    from .model_runner import build_bootstrap_patches
    from .stats import build_stats_patches
    from .ulysses import build_ulysses_patches

    # same order as the reference (plugins.py:111-126): bootstrap, arguments / configs / stats, then the Ulysses set.
    # The GPUModelRunner patch is applied by WorkerBasePatch inside each worker, after the fork.
    for patch in (build_bootstrap_patches() + build_args_patches() + build_config_patches() + build_compilation_patches() +
                  build_stats_patches() + build_ulysses_patches()):
        patch.apply_patch()
