"""GPU suite: bench.py itself, run the way the driver runs it (a child process, one JSON line on stdout) -- the
headline alone and one particle-filter leg, short.  The committed lines under profiles/ are held to the contract by
tests/test_bench_contract.py; this one makes sure the program that writes them still does."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


DRIVER_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
               "vs_baseline", "dtype", "data", "config", "roofline")


def run_bench(*args):
    """-> the FULL record (the sidecar of --detail-out), after holding the stdout line to what the driver needs of it:
    ONE JSON line, the last non-empty line of stdout, at most 8 KB (r05's 23.5 KB line came back
    unparsed), carrying the driver's keys with the sidecar's values."""
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        side = os.path.join(tmp, "detail.json")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--detail-out", side, *args],
                           capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        out_lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        lines = [ln for ln in out_lines if ln.startswith("{")]
        assert len(lines) == 1, "exactly one JSON line on stdout"
        assert out_lines[-1] == lines[0], "the JSON line is the last non-empty line of stdout (nothing after it)"
        assert len(lines[0].encode()) <= 8192, len(lines[0])
        line = json.loads(lines[0])
        full = json.load(open(side))
    for k in DRIVER_KEYS:
        assert k in line, k
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "vs_baseline"):
        assert line[k] == full[k], k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    assert abs(line["roofline"]["frac"] - full["roofline"]["frac"]) <= 1e-6 * full["roofline"]["frac"]
    assert "workload" in line["config"] and "legs" in line
    return full


def test_headline_line_from_a_live_run():
    d = run_bench("--legs", "none", "--no-cpu", "--steps", "8", "--warmup", "2")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 8 and d["warmup"] == 2 and d["dtype"] == "f64" and d["vs_baseline"] is None
    c, r = d["config"], d["roofline"]
    assert "cfg2" in c["workload"] and c["beams_after_filter"] == 1080 and c["scorer_calls_per_step"] > 700
    assert abs(d["value"] - c["scorer_calls_per_step"] * 1080 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert 0.05 < d["ms_per_step"] < 1.0  # a device chain, not seconds
    # one co-resident launch per match (csrc/hc_resident.hip), none gave up; the step takes the RAW scan (filter +
    # weights + beam trig + upload inside), the resident-scan figure rides along
    assert r["kernel"] == "k_hc_chain_resident" and r["launches"] > 0 and 0.0 < r["frac"] < 1.0
    assert c["resident"]["matches"] > 0 and c["resident"]["gave_up"] == 0
    # (8 timed steps each: a sanity bound, not a measurement -- one slow step moves either figure by 10 %)
    assert c["includes_filter_and_upload"] is True and 0.05 < c["ms_per_step_resident"] <= d["ms_per_step"] * 1.5
    assert d["parity"]["scenes"] == 0 and "--no-cpu" in d["parity"]["note"]
    assert abs(r["achieved"] - r["units_launched"] * r["bytes_per_unit"] / (r["avg_launch_us"] * 1e-6 * r["launches"]) / 1e9) \
        <= 1e-6 * r["achieved"]
    assert "roofline_sweep" in d and d["roofline_sweep"]["kernel"] == "k_score_point"


def test_particle_filter_leg_from_a_live_run():
    d = run_bench("--legs", "pf", "--no-cpu", "--steps", "4", "--warmup", "1", "--pf-steps", "3")
    pf = d["particle_filter"]
    assert "error" not in pf, pf
    assert pf["unit"] == "particles/s" and pf["ranks"] == 1 and pf["value"] > 1e4 and pf["roofline"]["bytes_per_unit"] == 232
    assert "cfg4" in pf["workload"] and pf["carry_reruns_last_step"] >= 0
