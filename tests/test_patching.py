"""ArcticPatch semantics — the same six behaviours the reference's unit test pins
(/root/reference/tests/unit_tests/test_patching.py:21-140), through both import paths."""
import types

import pytest

from arctic_inference.patching import ArcticPatch as CompatPatch
from arcticinference_amd.patching import ArcticPatch


def _classes():
    class Target:
        my_field = "original field"

        def my_method(self):
            return "original method"

        @classmethod
        def my_classmethod(cls):
            return "original classmethod"

    class Derived(Target):
        def my_method(self):
            return super().my_method() + " derived"

    return Target, Derived


def test_compat_import_path_is_same_class():
    assert CompatPatch is ArcticPatch


def test_adds_new_attributes():
    T, _ = _classes()

    class P(ArcticPatch[T]):
        new_field = "new field"

        def new_method(self):
            return "new method"

    P.apply_patch()
    assert T().new_field == "new field" and T().new_method() == "new method"
    assert "_arctic_patches" in T.__dict__ and T._arctic_patches["new_method"] is P and "new_field" in T._arctic_patches


def test_replaces_existing_attributes():
    T, _ = _classes()

    class P(ArcticPatch[T]):
        def my_method(self):
            return "patched"

    P.apply_patch()
    assert T().my_method() == "patched" and T._arctic_patches["my_method"] is P


def test_cannot_patch_twice():
    T, _ = _classes()

    class A(ArcticPatch[T]):
        def my_method(self):
            return "a"

    class B(ArcticPatch[T]):
        def my_method(self):
            return "b"

    A.apply_patch()
    with pytest.raises(ValueError, match="is already patched by"):
        B.apply_patch()


def test_method_and_classmethod_with_inheritance():
    T, D = _classes()

    class P(ArcticPatch[T]):
        def my_method(self):
            return self.__class__.__name__

        @classmethod
        def my_classmethod(cls):
            return f"patched classmethod for {cls.__name__}"

    P.apply_patch()
    assert T().my_method() == "Target" and D().my_method() == "Derived derived"
    assert T.my_classmethod() == "patched classmethod for Target"
    assert D.my_classmethod() == "patched classmethod for Derived"


def test_registry_is_per_class():
    T, D = _classes()

    class P(ArcticPatch[T]):
        my_field = "patched field"

    class Q(ArcticPatch[D]):
        my_field = "patched field"

    P.apply_patch()
    Q.apply_patch()
    assert T._arctic_patches == {"my_field": P} and D._arctic_patches == {"my_field": Q}


def test_errors_and_module_target():
    with pytest.raises(TypeError):
        class NoTarget(ArcticPatch):
            pass
    with pytest.raises(TypeError):
        ArcticPatch[42]
    with pytest.raises(TypeError):
        ArcticPatch.apply_patch()
    mod = types.ModuleType("some_module")

    class M(ArcticPatch[mod]):
        NEW_CONSTANT = 7

        @staticmethod
        def new_function():
            return "f"

    M.apply_patch()
    assert mod.NEW_CONSTANT == 7 and mod.new_function() == "f"
