"""maybe_prefix / AutoWeightsLoader of the stand-in: checkpoint names are routed to the child module that owns them (its
own load_weights when it has one), plain parameters are loaded through their weight_loader."""
from typing import Iterable, Optional, Tuple

import torch

from vllm.model_executor.model_loader.weight_utils import default_weight_loader


def maybe_prefix(prefix: str, name: str) -> str:
    return name if not prefix else f"{prefix}.{name}"


class AutoWeightsLoader:
    def __init__(self, module: torch.nn.Module, skip_prefixes: Optional[list] = None, ignore_unexpected_prefixes=None):
        self.module, self.skip = module, tuple(skip_prefixes or ())

    def load_weights(self, weights: Iterable[Tuple[str, torch.Tensor]], mapper=None) -> set:
        groups, loaded = {}, set()
        for name, w in weights:
            if self.skip and name.startswith(self.skip):
                continue
            head, _, rest = name.partition(".")
            groups.setdefault(head, []).append((rest, w))
        params = dict(self.module.named_parameters())
        for head, items in groups.items():
            child = getattr(self.module, head, None)
            if isinstance(child, torch.nn.Module) and hasattr(child, "load_weights"):
                loaded |= {f"{head}.{n}" for n in child.load_weights(items)}
                continue
            for rest, w in items:
                full = f"{head}.{rest}" if rest else head
                p = params[full]
                getattr(p, "weight_loader", default_weight_loader)(p, w)
                loaded.add(full)
        return loaded
