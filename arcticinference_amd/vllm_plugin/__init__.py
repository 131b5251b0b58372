"""vLLM plugin surface (mirror of /root/reference/arctic_inference/vllm/): entry point, flag / config names
and defaults, stats growth, and the model-runner glue.  Everything that needs vLLM is built lazily inside
functions, so this package imports (and its pure logic is tested) without vLLM."""
