#!/usr/bin/env python3
"""The reference's single-hypothesis world loop (tinySLAM / vinySLAM) with and without the drop-in, in ONE process on
the GPU box: oracle/_ref/libslamref_world.so (built here from the reference's headers, oracle/Makefile) runs
init_1h_slam's world and the HBM-resident world (host/slamhip_resident_world.h) over the same synthetic scans, compares
them (trajectory and final map bit for bit) and reports the wall time per scan of each -- the two worlds run one AFTER
the other over the same scans (interleaved, the resident world's queued map update would finish unseen while the
reference computes its next scan), the resident world's time includes its last queued update.
A measurement for DESIGN.md section 6 -- not part of bench.py: the library under oracle/ is test infrastructure.

    python tools/world_loop_timing.py [n_scans] [n_beams]
"""
import ctypes as C
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(root, "oracle", "_ref", "libslamref_world.so"))
lib.refworld_compare_resident.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                          C.POINTER(C.c_double), C.POINTER(C.c_double)]
n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n_beams = int(sys.argv[2]) if len(sys.argv) > 2 else 1080
# the production shape: nobody subscribed to the HIP world's matcher (the adapter then filters the raw scan through
# slamhip_scan_filter_upload); pass a third argument to keep the counting observer attached
if len(sys.argv) <= 3:
    os.environ["REFWORLD_NO_OBSERVER"] = "1"
for preset, pname in ((0, "tinySLAM"), (1, "vinySLAM")):
    for matcher, mname in ((1, "HC"), (0, "MC")):
        poses = (C.c_double * (6 * n_scans))()
        out = (C.c_double * 18)()
        rc = lib.refworld_compare_resident(preset, matcher, n_scans, n_beams, 0, 4.0, poses, out)
        assert rc == 0
        ref_s, hip_s = out[16], out[17]
        print("%s %s, %d beams, %d scans: reference world %.2f ms/scan, resident world %.3f ms/scan (x%.0f); "
              "pose mismatches %d, map mismatches %d of %d cells, scorer calls %d / %d"
              % (pname, mname, n_beams, n_scans - 1, 1e3 * ref_s / (n_scans - 1), 1e3 * hip_s / (n_scans - 1),
                 ref_s / hip_s, out[0], out[3], out[2], out[5], out[6]))
