#!/usr/bin/env python3
"""Golden vectors for the on-disk fixture formats and the offline scan-matching tool (N3).

For each case: a map is built by the compiled reference (scan_generate + append_scan), dumped with
GridMap::save_state (oracle/_ref/libslamref.so: ref_map_save_state); the pose / scan / properties
files are written as text; the reference's own tool -- oracle/_ref/sm_runner, compiled from
src/utils/sm_runner.cpp where it lies -- is run on the four files and its stdout recorded.  The
same matcher is also run through the harness for the full-precision result (the tool prints 6
digits).  Everything (file bytes + expected outputs) goes into fixtures.npz; the tests unpack the
files into a temp dir.

    python tests/golden/make_golden_fixtures.py
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

GOLDEN_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(GOLDEN_DIR))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from pyoracle import *  # noqa: E402,F401,F403

BASE_PROPS = """# common part, included by the case files
slam/mapping/grid/type=unbounded_plain
slam/map/meters_per_cell=0.1
slam/scmtch/spe/type=wmpp
this line has no delimiter
"""

BF_PROPS = "slam/scmtch/type=BF\n"

CASES = {
    "hc_mean": dict(cell=REF_CELL_MEAN, weighting=0, kind=SM_HC, params=[6, 0.1, 0.1], oope=OOPE_OBSTACLE,
                    oie=OIE_DISCREPANCY, props="""<common/base.properties>
slam/mapping/grid/area/type=mean_probability
slam/scmtch/type=HC
slam/scmtch/spe/wmpp/weighting/type=even
slam/scmtch/HC/distortion/failed_attempts_limit=6
"""),
    "mc_tbm": dict(cell=REF_CELL_TBM, weighting=1, kind=SM_MC, params=[666666, 0.2, 0.1, 20, 100],
                   oope=OOPE_OBSTACLE, oie=OIE_DISCREPANCY, props="""# MC on a TBM map
<common/base.properties>
slam/mapping/grid/area/type=tbm_consistent
slam/scmtch/type=MC
slam/scmtch/MC/seed=1
slam/scmtch/MC/seed=666666
slam/scmtch/spe/wmpp/weighting/type=viny
"""),
    # common/bf.properties says BF and is merged first, so its slam/scmtch/type wins over this
    # file's (properties_providers.h:88-96) -- the case file asks for HC but the run is BF
    "bf_affine": dict(cell=REF_CELL_AFFINE, weighting=0, kind=SM_BF,
                      params=[-0.2, 0.2, 0.1, -0.2, 0.2, 0.1, -np.deg2rad(2), np.deg2rad(2), np.deg2rad(1)],
                      oope=OOPE_MAX, oie=OIE_OCCUPANCY, skip_rate=1, max_range=12.0,
                      props="""<common/base.properties>
<common/bf.properties>
slam/scmtch/type=HC
slam/mapping/grid/area/type=affine_quality_merge
slam/scmtch/oope/type=max
slam/scmtch/spe/wmpp/weighting/type=even
slam/scmtch/oie/type=occupancy
slam/scmtch/spe/wmpp/sp_skip_rate=1
slam/scmtch/spe/wmpp/sp_max_usable_range=12.0
slam/scmtch/BF/x/from=-0.2
slam/scmtch/BF/x/to=0.2
slam/scmtch/BF/y/from=-0.2
slam/scmtch/BF/y/to=0.2
slam/scmtch/BF/t/from=%.17g
slam/scmtch/BF/t/to=%.17g
""" % (-np.deg2rad(2), np.deg2rad(2))),
}


def main():
    R = Ref()
    R.lib.ref_map_save_state.argtypes = [C.c_void_p, C.c_char_p]
    runner = os.path.join(ROOT, "oracle", "_ref", "sm_runner")
    out = {"common/base.properties": np.frombuffer(BASE_PROPS.encode(), np.uint8),
           "common/bf.properties": np.frombuffer(BF_PROPS.encode(), np.uint8)}
    scale, n = 0.1, 90
    for name, c in CASES.items():
        gt = R.map_create(REF_CELL_MOCK, MAP_UNBOUNDED_PLAIN, n, n, scale, 0.0)
        gt.stamp_text(R.cecum_text(31, 23, 2), (-15, 10))
        gt.stamp_text(R.cecum_text(13, 9, 3), (-6, -4))
        pose = (scale / 2, scale / 2 - 2 * scale, np.deg2rad(90))
        raw = R.scan_generate(gt, pose, 15, 270, 360)
        r, a, o, _ = raw.get()
        m = R.map_create(c["cell"], MAP_UNBOUNDED_PLAIN, n, n, scale)
        for _k in range(4):
            R.append_scan(m, raw, pose, quality=0.9, blur=0.2)
        noisy = np.array([pose[0] + 0.06, pose[1] - 0.05, pose[2] + 0.025])
        with tempfile.TemporaryDirectory() as td:
            os.makedirs(os.path.join(td, "common"))
            open(os.path.join(td, "common", "base.properties"), "w").write(BASE_PROPS)
            open(os.path.join(td, "common", "bf.properties"), "w").write(BF_PROPS)
            open(os.path.join(td, "cfg.properties"), "w").write(c["props"])
            open(os.path.join(td, "p.pose2D"), "w").write("%.17g %.17g %.17g\n" % tuple(noisy))
            with open(os.path.join(td, "s.scan2D"), "w") as f:
                f.write("%d\n" % r.size)
                for k in range(r.size):
                    f.write("%.17g %.17g %d\n" % (r[k], a[k], o[k]))
            R.lib.ref_map_save_state(m.h, os.path.join(td, "m.map").encode())
            txt = subprocess.run([runner, "cfg.properties", "p.pose2D", "m.map", "s.scan2D"], cwd=td,
                                 capture_output=True, text=True)
            if txt.returncode:
                raise SystemExit(txt.stdout + txt.stderr)
            txt = txt.stdout
            files = {k: open(os.path.join(td, k), "rb").read()
                     for k in ("cfg.properties", "p.pose2D", "s.scan2D", "m.map")}
        print(name, "->", txt.strip().splitlines()[-1])
        # full precision through the harness: same map object, scan as the tool rebuilds it
        spe = R.spe_create(c["oope"], c["oie"], c["weighting"], c.get("skip_rate", 0), c.get("max_range", -1.0))
        scan = R.scan_create(r, a, o)
        t = R.process_scan(R.matcher_create(c["kind"], spe, c["params"]), scan, noisy, m)
        md = m.to_data()
        for k, v in files.items():
            out["%s/%s" % (name, k)] = np.frombuffer(v, np.uint8)
        out[name + "/stdout"] = np.frombuffer(txt.encode(), np.uint8)
        out[name + "/prob"] = np.array(t["prob"])
        out[name + "/delta"] = t["delta"]
        out[name + "/n_calls"] = np.array(t["n_calls"])
        out[name + "/payload"] = md.payload
        out[name + "/origin"] = np.array(md.origin)
        out[name + "/scan"] = np.stack([r, a, o.astype(np.float64)])
        out[name + "/pose"] = noisy
    path = os.path.join(GOLDEN_DIR, "fixtures.npz")
    np.savez_compressed(path, **out)
    print("wrote fixtures.npz", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
