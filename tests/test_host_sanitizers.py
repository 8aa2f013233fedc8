"""Host-side sanitizer runs (CPU only: GPU AddressSanitizer is not available on the pool).

tests/native/spec_tree_asan.cpp drives the speculation builders of csrc/matchers.h (HC round trees,
MC chains) with random accept/reject outcomes under -fsanitize=address,undefined.  Regression for a
heap-use-after-free in SpecTree::build_rounds (a pointer into rounds_ read after emplace_back had
reallocated the vector) that made one GMapping golden flaky on the GPU box."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="no host compiler")
def test_speculation_builders_under_asan(tmp_path):
    exe = str(tmp_path / "spec_tree_asan")
    hip_inc = "/opt/rocm/include"
    if not os.path.exists(os.path.join(hip_inc, "hip", "hip_runtime.h")):
        pytest.skip("HIP headers not found")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-fno-omit-frame-pointer", "-D__HIP_PLATFORM_AMD__", "-I" + hip_inc, "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "slam-constructor_amd", "csrc"),
           os.path.join(ROOT, "tests", "native", "spec_tree_asan.cpp"),
           os.path.join(ROOT, "slam-constructor_amd", "csrc", "mt_block.cpp"), "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("ok ")
