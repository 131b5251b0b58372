"""The host suffix tree (arena, flat hash, best-child bookkeeping, tree-mode speculation) under AddressSanitizer and
UndefinedBehaviorSanitizer: a CPU-only build of csrc/suffix_host.hpp with g++ driven by tests/native/fuzz_host_tree.cpp
(GPU sanitizers are not available on the pool; the device side only reads the mirror the host side writes)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_host_tree_fuzz_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "fuzz_host_tree")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-fno-omit-frame-pointer", "-I" + os.path.join(ROOT, "arcticinference_amd", "csrc"),
           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "fuzz_host_tree.cpp"), "-o", exe]
    subprocess.check_call(cmd)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([exe], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-4000:]
    assert out.stdout.strip().startswith("ok"), out.stdout[-2000:]
