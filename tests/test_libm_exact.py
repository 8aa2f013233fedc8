"""CPU suite: csrc/libm_exact.h -- glibc 2.35's sin / cos / exp restated operation for operation --
against the RUNNING libm, bit for bit, over more than 10^8 arguments per function (tests/native/libm_exact_check.cpp),
in both builds glibc ships on x86-64:
  * the FMA build (what libm's ifunc picks where AVX2 + FMA are usable -- this container and the GPU box's EPYC),
  * the plain build, reached by masking AVX2 / FMA out of libm's choice with GLIBC_TUNABLES, and -- for sin / cos --
    through sincos(), which has no FMA build at all;
plus the two lookup tables (computed from their definitions by tools/gen_libm_tables.py, 18 published low words of
glibc's sincostab.c kept as errata) and the library's own probe of which build the host runs."""
import json
import os
import platform
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "libm_exact_check.cpp")
glibc = platform.libc_ver()
pytestmark = pytest.mark.skipif(glibc[0] != "glibc" or tuple(int(v) for v in glibc[1].split(".")[:2]) < (2, 28),
                                reason="the restatement is of glibc >= 2.28's sin / cos / exp")


def cpu_has(*flags):
    try:
        words = open("/proc/cpuinfo").read().split("flags", 1)[1].split("\n", 1)[0].split()
    except (OSError, IndexError):
        return False
    return all(f in words for f in flags)


@pytest.fixture(scope="module")
def check_bin(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("libm") / "libm_exact_check")
    flags = ["-mfma"] if cpu_has("fma") else []
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-pthread", *flags, SRC, "-o", out], check=True)
    return out


def run(check_bin, fma, millions, via_sincos=0, env=None):
    r = subprocess.run([check_bin, str(fma), str(millions), str(os.cpu_count() or 1), str(via_sincos)], capture_output=True,
                       text=True, env=dict(os.environ, **(env or {})), timeout=900)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    return d, r.stderr[-1500:]


@pytest.mark.skipif(not cpu_has("fma", "avx2"), reason="libm runs its FMA build only where AVX2 + FMA are usable")
def test_fma_build_against_the_running_libm(check_bin):
    d, err = run(check_bin, 1, 10)  # 10 M per range: 110 M sin, 110 M cos, 70 M exp arguments
    assert d["sin_args"] > 1e8 and d["cos_args"] > 1e8 and d["exp_args"] >= 7e7
    assert (d["sin_bad"], d["cos_bad"], d["exp_bad"]) == (0, 0, 0), err


def test_plain_build_against_libm_with_fma_masked_out(check_bin):
    d, err = run(check_bin, 0, 10, env={"GLIBC_TUNABLES": "glibc.cpu.hwcaps=-AVX2,-FMA"})
    assert d["sin_args"] > 1e8 and (d["sin_bad"], d["cos_bad"], d["exp_bad"]) == (0, 0, 0), err


def test_plain_build_is_what_sincos_computes(check_bin):
    """sincos() has no per-CPU variant in glibc 2.35: the pair gcc fuses std::sin + std::cos of one argument into (the
    reference's CachedTrigonometryProvider::set_base_angle, trigonometry_utils.h:57-60) is the plain build everywhere"""
    d, err = run(check_bin, 0, 3, via_sincos=1)
    assert (d["sin_bad"], d["cos_bad"]) == (0, 0), err


@pytest.mark.skipif(not cpu_has("fma", "avx2"), reason="needs a host whose libm runs the FMA build")
def test_the_two_builds_do_differ(check_bin):
    """(a check of the check: the FMA restatement against sincos() -- the plain build -- must show mismatches)"""
    d, _ = run(check_bin, 1, 1, via_sincos=1)
    assert d["sin_bad"] > 100 and d["cos_bad"] > 100


def test_tables_are_what_the_generator_computes():
    pytest.importorskip("mpmath")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_libm_tables as gen
    sc, ex = gen.sincos_table(), gen.exp_table()
    hdr = open(os.path.join(ROOT, "slam-constructor_amd", "csrc", "libm_exact_tables.h")).read()
    sc_txt = hdr.split("#define SLAMHIP_LIBM_SINCOS_TABLE", 1)[1].split("// exp:", 1)[0].replace("\\", " ")
    got_sc = [float.fromhex(t.strip()) for t in sc_txt.replace("\n", " ").split(",") if t.strip()]
    assert len(got_sc) == 440 and all(gen.bits(a) == gen.bits(b) for a, b in zip(got_sc, sc))
    ex_txt = hdr.split("#define SLAMHIP_LIBM_EXP_TABLE", 1)[1].split("}", 1)[0].replace("\\", " ")
    got_ex = [int(t.strip().rstrip("ul"), 16) for t in ex_txt.replace("\n", " ").split(",") if t.strip()]
    assert got_ex == ex
    # without the errata the table is the correctly rounded one: exactly 18 low words differ, none by more than 64 ulps
    plain = gen.sincos_table(errata=False)
    assert sum(gen.bits(a) != gen.bits(b) for a, b in zip(plain, sc)) == 18 == len(gen.SINCOSTAB_ERRATA)
    libm = "/lib/x86_64-linux-gnu/libm.so.6"
    if os.path.exists(libm) and glibc[1] == "2.35":
        assert gen.check_libm(libm, sc, ex)


def test_the_library_finds_the_hosts_build():
    import __graft_entry__ as ge
    pkg = ge.load_package()
    v = pkg.libm_variant()
    assert v == (1 if cpu_has("fma", "avx2") else 0)
    code = "import __graft_entry__ as ge; print(ge.load_package().libm_variant())"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT,
                       env=dict(os.environ, GLIBC_TUNABLES="glibc.cpu.hwcaps=-AVX2,-FMA"), timeout=300)
    assert r.stdout.strip().splitlines()[-1] == "0", r.stderr[-500:]
