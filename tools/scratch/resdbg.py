import sys, os, ctypes as C, numpy as np
os.environ["REFWORLD_TRACE"] = "9"
sys.path.insert(0, "tests")
import test_gpu_world as t
L = C.CDLL(t.SO)
r, poses = t.run_resident(L, 1, 1, 1, n_scans=10)
