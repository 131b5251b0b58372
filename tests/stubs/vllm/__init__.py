"""tests-only stand-in for vLLM 0.9.2: names and signatures the ArcticInference plugin binds to (tests/stubs/README.md)."""
__version__ = "0.9.2"


class _ModelRegistry:
    def __init__(self):
        self.models = {}

    def register_model(self, arch: str, target) -> None:
        self.models[arch] = target

    def resolve(self, arch: str):
        target = self.models[arch]
        if isinstance(target, str):
            import importlib
            mod, name = target.split(":")
            target = getattr(importlib.import_module(mod), name)
        return target


ModelRegistry = _ModelRegistry()
