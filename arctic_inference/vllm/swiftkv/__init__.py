def __getattr__(name):      # the model classes import vLLM: resolved on first use (ModelRegistry's "module:Class" string)
    from arcticinference_amd.vllm_plugin import swiftkv_model
    return getattr(swiftkv_model, name)
