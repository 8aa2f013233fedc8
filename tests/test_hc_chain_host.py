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
    # r06: ... and again with the inert tail in closed form (hc_inert): same traces, the tail taken wherever the limit
    # lies behind the point at which the steps vanish (limits 128 and 250: 96 of the 240 matches)
    assert "inert tails taken in" in r.stdout and int(r.stdout.split("inert tails taken in")[1].split()[0]) >= 96


@pytest.mark.skipif(shutil.which("g++") is None, reason="no host compiler")
def test_certificate_bound_against_the_plain_accept_loop(tmp_path):
    """r06: the CERTIFICATE behind the closed-form tail (csrc/hc_chain.h hc_cert_beam, the very function the kernel calls) is
    an argument about rounding.  tests/native/hc_cert_test.cpp holds it, in host arithmetic, against the plain accept loop
    over a 1-cell scorer whose cells all differ: 20 000 matches with adversarial geometry -- 1 ... 12 beams, levers up
    to 30 m, end points placed 1e-14 ... 1e-3 m from cell edges, steps starting on either side of that distance, the
    certificate consulted after every 1 ... 42 failed rounds -- must give the reference's scorer-call sequence bit for
    bit (2 000 000 of them did, 1.35 G scorer calls: LOG r06).  And the check can fail: the same program over a header
    whose translation bound is loosened by half, or whose rotation bound is a third of what it must be, does."""
    hip_inc = "/opt/rocm/include"
    if not os.path.exists(os.path.join(hip_inc, "hip", "hip_runtime.h")):
        pytest.skip("HIP headers not found")
    csrc = os.path.join(ROOT, "slam-constructor_amd", "csrc")
    src = os.path.join(ROOT, "tests", "native", "hc_cert_test.cpp")

    def build(exe, first_inc=None, sanitize=True):
        cmd = ["g++", "-std=c++17", "-O1", "-g", "-ffp-contract=off", "-D__HIP_PLATFORM_AMD__", "-I" + hip_inc,
               "-I" + os.path.join(ROOT, "include")]
        if sanitize:
            cmd += ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer"]
        if first_inc:
            cmd.append("-I" + first_inc)
        cmd += ["-I" + csrc, src, "-o", exe]
        subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=600)

    exe = str(tmp_path / "hc_cert_test")
    build(exe)
    r = subprocess.run([exe, "20000"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("ok 20000 matches")
    early = int(r.stdout.split("before the identical-pose rule could in")[1].split(";")[0])
    assert early > 10000  # (the certificate really decides most of them)
    header = open(os.path.join(csrc, "hc_chain.h")).read()
    for k, (old, new) in enumerate([("*t_t = ok ? avail * 0.99 : 0.0;", "*t_t = ok ? avail * 1.5 : 0.0;"),
                                    ("avail / (1.5 * ar)", "avail / (0.5 * ar)")]):
        assert header.count(old) == 1
        d = tmp_path / ("mutant%d" % k)
        d.mkdir()
        (d / "hc_chain.h").write_text(header.replace(old, new))
        mexe = str(tmp_path / ("hc_cert_mutant%d" % k))
        build(mexe, first_inc=str(d), sanitize=False)
        rm = subprocess.run([mexe, "20000"], capture_output=True, text=True, timeout=900)
        assert rm.returncode == 1 and rm.stdout.startswith("FAIL"), (k, rm.stdout[:300])


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
