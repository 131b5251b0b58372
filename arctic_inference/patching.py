from arcticinference_amd.patching import ArcticPatch  # noqa: F401
