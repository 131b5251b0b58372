from arcticinference_amd.vllm_plugin.plugins import arctic_inference_plugin  # noqa: F401
