def support_torch_compile(cls=None, **kwargs):
    """vLLM wraps the class for torch.compile; the stand-in runs it eagerly."""
    if cls is None:
        return lambda c: c
    return cls
