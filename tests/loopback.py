"""TEST INFRASTRUCTURE: builds tests/native/loopback_transport.cpp (a slamhip_shard_transport whose ranks are threads
of this process) and attaches contexts to in-process groups through slamhip_shard_attach."""
import ctypes as C
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None


class ShardMsg(C.Structure):
    _fields_ = [("peer", C.c_int), ("buf", C.c_void_p), ("bytes", C.c_size_t)]


class ShardTransport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("allgather", C.c_void_p), ("exchange", C.c_void_p), ("destroy", C.c_void_p)]


def lib():
    global _lib
    if _lib is None:
        out = os.path.join(tempfile.mkdtemp(prefix="slamhip_loopback_"), "libslamhip_loopback.so")
        rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
        cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I%s/include" % rocm,
               "-I%s/include" % ROOT, os.path.join(ROOT, "tests", "native", "loopback_transport.cpp"), "-o", out,
               "-L%s/lib" % rocm, "-lamdhip64", "-Wl,-rpath,%s/lib" % rocm, "-lpthread"]
        subprocess.check_call(cmd)
        _lib = C.CDLL(out)
        _lib.loopback_transport_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(ShardTransport)]
        _lib.loopback_transport_create.restype = C.c_int
        _lib.loopback_transport_set_timeout.argtypes = [C.c_char_p, C.c_int]
        _lib.loopback_transport_set_timeout.restype = C.c_int
    return _lib


def set_timeout(name, ms):
    """every wait of group `name` gives up after `ms` (the group is dead from then on)"""
    if lib().loopback_transport_set_timeout(name.encode(), int(ms)):
        raise RuntimeError("loopback_transport_set_timeout failed")


def attach(pkg, ctx, name, rank, world):
    """ctx joins the in-process group `name` as rank `rank` of `world` (the table is copied by the library)."""
    t = ShardTransport()
    rc = lib().loopback_transport_create(name.encode(), rank, world, C.byref(t))
    if rc:
        raise RuntimeError("loopback_transport_create failed: %d" % rc)
    ctx.shard_attach(t, rank, world)
