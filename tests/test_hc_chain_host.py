"""CPU suite: the logic of the device-resident hill climbing (csrc/hc_chain.h, hc_shape.h) run lane by lane
on the host (tests/native/hc_chain_test.cpp) against the plain accept loop of
PoseEnumerationScanMatcher::process_scan over HillClimbingPoseEnumerator: 240 matches (failed-round limits
1..250, quantised scores so that ties occur, small and boosted speculation shapes) must give the same
scorer-call sequence bit for bit.  Built with ASan/UBSan (GPU sanitizers are not available on the pool)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="no host compiler")
def test_chain_logic_equals_reference_loop(tmp_path):
    exe = str(tmp_path / "hc_chain_test")
    hip_inc = "/opt/rocm/include"
    if not os.path.exists(os.path.join(hip_inc, "hip", "hip_runtime.h")):
        pytest.skip("HIP headers not found")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-fno-omit-frame-pointer", "-ffp-contract=off", "-D__HIP_PLATFORM_AMD__", "-I" + hip_inc,
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "slam-constructor_amd", "csrc"),
           os.path.join(ROOT, "tests", "native", "hc_chain_test.cpp"), "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=600)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("ok 240 matches")


@pytest.mark.skipif(shutil.which("g++") is None, reason="no host compiler")
def test_mc_chain_logic_equals_reference_loop(tmp_path):
    """csrc/mc_chain.h (candidate j of a state in closed form, first acceptance, the enumerator state after the
    consumed candidates) against the plain accept loop over GaussianPoseEnumerator: 168 matches, limits from 1/50 to
    4096/4096, 1..384 candidates per super-step, quantised scores (ties), three matches per engine."""
    exe = str(tmp_path / "mc_chain_test")
    hip_inc = "/opt/rocm/include"
    if not os.path.exists(os.path.join(hip_inc, "hip", "hip_runtime.h")):
        pytest.skip("HIP headers not found")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-fno-omit-frame-pointer", "-ffp-contract=off", "-D__HIP_PLATFORM_AMD__", "-I" + hip_inc,
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "slam-constructor_amd", "csrc"),
           os.path.join(ROOT, "tests", "native", "mc_chain_test.cpp"),
           os.path.join(ROOT, "slam-constructor_amd", "csrc", "mt_block.cpp"), "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=600)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("ok 168 matches")


@pytest.mark.skipif(shutil.which("g++") is None, reason="no host compiler")
def test_block_mt19937_is_std_mt19937(tmp_path):
    """csrc/mt_block.cpp (the Monte-Carlo matcher's engine: 624 words per refill, vectorizable, an AVX2 clone picked at
    run time) against std::mt19937: six seeds, two million words each."""
    exe = str(tmp_path / "mt_block_test")
    cmd = ["g++", "-std=c++17", "-O2", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-I" + os.path.join(ROOT, "slam-constructor_amd", "csrc"),
           os.path.join(ROOT, "tests", "native", "mt_block_test.cpp"),
           os.path.join(ROOT, "slam-constructor_amd", "csrc", "mt_block.cpp"), "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=600)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("ok")


@pytest.mark.skipif(shutil.which("g++") is None, reason="no host compiler")
def test_pair_tape_is_std_normal_distribution(tmp_path):
    """The Monte-Carlo matcher's tape of Marsaglia pairs (matchers.h PairTape: block engine + vectorized attempts)
    against std::normal_distribution over std::mt19937 -- what the reference's GaussianRV1D draws from
    (src/core/random_utils.h:17-34): 4 x 300 000 pairs and 200 000 candidates of three distributions on one engine."""
    exe = str(tmp_path / "pair_tape_test")
    hip_inc = "/opt/rocm/include"
    if not os.path.exists(os.path.join(hip_inc, "hip", "hip_runtime.h")):
        pytest.skip("HIP headers not found")
    cmd = ["g++", "-std=c++17", "-O2", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-ffp-contract=off", "-D__HIP_PLATFORM_AMD__", "-I" + hip_inc,
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "slam-constructor_amd", "csrc"),
           os.path.join(ROOT, "tests", "native", "pair_tape_test.cpp"),
           os.path.join(ROOT, "slam-constructor_amd", "csrc", "mt_block.cpp"), "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=600)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("ok")
