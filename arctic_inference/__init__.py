"""Import-path compatibility package: `arctic_inference.*` names of the reference resolve to the
MI355X implementation in `arcticinference_amd` so an existing vLLM install picks this build up
unchanged (entry point `arctic_inference.vllm.plugins:arctic_inference_plugin`)."""
