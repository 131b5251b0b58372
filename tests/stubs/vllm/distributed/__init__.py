from . import parallel_state  # noqa: F401
