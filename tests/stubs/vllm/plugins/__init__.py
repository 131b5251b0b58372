load_calls = 0


def load_general_plugins() -> None:
    """vLLM walks the `vllm.general_plugins` entry points; the stand-in calls the one plugin under test."""
    global load_calls
    load_calls += 1
