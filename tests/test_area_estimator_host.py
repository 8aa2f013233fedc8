"""CPU suite: the device restatement of AreaOccupancyEstimator::estimate_occupancy
(slam-constructor_amd/csrc/area_estimator_device.h -- the four edge intersections kept in fixed slots with flags, no
arrays: 0 bytes of scratch in every K6 kernel) compiled for the host and compared, bit for bit, with the oracle's
vector-shaped restatement (oracle/area_estimator.h, itself pinned to the reference's 64 unit-test cases and to goldens
of the compiled reference) over three million beams aimed at the special cases: corners, edges, edge lines, far cells."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_area_estimator_equals_the_oracle_twin(tmp_path):
    exe = str(tmp_path / "ae_host_test")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-I" + ROOT,
                           os.path.join(ROOT, "tests", "native", "area_estimator_host_test.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 mismatches" in r.stdout
