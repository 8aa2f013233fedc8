"""The per-particle-maps filter of tests/test_gpu_particle_maps.py at length, in-tile masks and settle states checked after
every step (and every particle's map and counters against the oracle's): python tools/scratch/tile_stress.py [extra steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as ge  # noqa: E402
from pyoracle import Oracle  # noqa: E402
from test_gpu_particle_maps import run_both  # noqa: E402

pkg = ge.load_package()
extra = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for mode, options in (("fast", ()), ("sorted", ((pkg.OPT_K6_BATCH_FAST, 0),))):
    # (the parameters of test_particle_maps_resampling_shares_then_clones_tiles: resamples, shares tiles, clones them)
    pf, log, _ = run_both(pkg, Oracle(), n=8, seed0=3000, n_steps_extra=extra, options=options, check_masks=True,
                          gp=[0, 0.1, 0, 0.05, 0, 0, 0, 0])
    print(mode, "steps", len(log), "resamplings", sum(1 for r_, _ in log if r_), "cow copies", log[-1][1]["cow_copies"])
