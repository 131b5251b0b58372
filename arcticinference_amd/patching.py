"""In-place patching of classes and modules, with the surface of the reference's ArcticPatch
(/root/reference/arctic_inference/patching.py:25-139): `class X(ArcticPatch[Target])` declares the
additions / replacements, `X.apply_patch()` installs them on Target and records, per attribute, which
patch owns it in `Target._arctic_patches`; patching the same attribute twice is a ValueError; a patch
declared without `[Target]`, or with a target that is neither a class nor a module, is a TypeError.
"""
from __future__ import annotations

import logging
from types import ModuleType
from typing import Dict

logger = logging.getLogger(__name__)

_TARGET_ATTR = "_arctic_patch_target"
# bookkeeping names of a class body that are never copied onto the target
_SKIPPED = frozenset({_TARGET_ATTR, "__dict__", "__weakref__", "__module__", "__doc__", "__parameters__",
                      "__qualname__", "__orig_bases__", "__annotations__", "__firstlineno__",
                      "__static_attributes__"})


class ArcticPatch:
    def __init_subclass__(cls, **kwargs):
        super().__init_subclass__(**kwargs)
        if not hasattr(cls, _TARGET_ATTR):
            raise TypeError("Subclasses of ArcticPatch must be defined as ArcticPatch[Target] to specify a patch target")

    @classmethod
    def __class_getitem__(cls, target):
        if not isinstance(target, (type, ModuleType)):
            raise TypeError(f"ArcticPatch can only target a class or module, not {type(target)}")
        # an intermediate base that remembers the target; the user's class derives from it
        return type(f"{cls.__name__}[{target.__name__}]", (cls,), {_TARGET_ATTR: target})

    @classmethod
    def apply_patch(cls) -> None:
        if cls is ArcticPatch or not hasattr(cls, _TARGET_ATTR):
            raise TypeError("apply_patch() must be called on a subclass of ArcticPatch")
        target = getattr(cls, _TARGET_ATTR)
        # the registry is per target: a derived class does not share its base's dict
        if "_arctic_patches" not in vars(target):
            setattr(target, "_arctic_patches", {})
        owners: Dict[str, type] = vars(target)["_arctic_patches"]
        for name, attr in list(vars(cls).items()):
            if name in _SKIPPED:
                continue
            if name in owners:
                raise ValueError(f"{target.__name__}.{name} is already patched by {owners[name].__name__}")
            owners[name] = cls
            existed = hasattr(target, name)
            # raw descriptors are copied as they are: a classmethod / staticmethod / property object set
            # on the target binds to whichever (derived) class it is reached through
            setattr(target, name, attr)
            logger.info("%s %s %s.%s", cls.__name__, "replaced" if existed else "added", target.__name__, name)
