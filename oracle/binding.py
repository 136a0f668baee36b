"""ctypes binding of the CPU oracle (oracle/mpt_oracle.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (metalpathtracer_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libmpt_oracle.so")

RNG_LITERAL, RNG_PHILOX = 0, 1
BSDF_LAMBERT, BSDF_SCATTER, BSDF_SCATTER_ALL = 0, 1, 2

COUNTER_NAMES = (
    "rays", "node_pops", "aabb_pass", "prim_tests", "sphere_tests", "tri_tests",
    "pushes", "misses", "bounces", "emissive_hits", "depth_exhausted", "paths",
)


class Uniforms(C.Structure):
    """144-byte UniformsData (R/Renderer/Shaders/Structs.h:23-41)."""
    _fields_ = [
        ("primitiveIndex", C.c_int32), ("_p0", C.c_int32 * 3),
        ("cameraPosition", C.c_float * 4),
        ("screenSize", C.c_float * 2), ("_p1", C.c_float * 2),
        ("viewportU", C.c_float * 4),
        ("viewportV", C.c_float * 4),
        ("firstPixelPosition", C.c_float * 4),
        ("randomSeed", C.c_float * 4),
        ("primitiveCount", C.c_uint64),
        ("triangleCount", C.c_uint64),
        ("frameCount", C.c_uint64),
        ("totalPrimitiveCount", C.c_uint64),
    ]


assert C.sizeof(Uniforms) == 144


class RenderParams(C.Structure):
    _fields_ = [
        ("rng_mode", C.c_int32), ("bsdf_mode", C.c_int32), ("max_depth", C.c_int32), ("accumulate", C.c_int32),
        ("sample_begin", C.c_uint32), ("sample_count", C.c_uint32),
        ("seed_lo", C.c_uint32), ("seed_hi", C.c_uint32),
        ("row_begin", C.c_int32), ("row_end", C.c_int32),
    ]


def build(force=False):
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(
            os.path.join(_HERE, "mpt_oracle.cpp")):
        subprocess.check_call(["make", "-C", _HERE, "_build/libmpt_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp, ip, u64p = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint64)
        L.orc_pcg_hash.restype = C.c_uint32
        L.orc_pcg_hash.argtypes = [C.c_uint32]
        L.orc_pcg_float.restype = C.c_float
        L.orc_pcg_float.argtypes = [C.c_uint32]
        L.orc_bitm_random.restype = C.c_uint32
        L.orc_bitm_random.argtypes = [C.POINTER(C.c_uint32)]
        L.orc_host_random_float.restype = C.c_float
        L.orc_host_random_float.argtypes = [C.POINTER(C.c_uint32)]
        L.orc_philox.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.orc_sincos_2pi.argtypes = [C.c_float, fp, fp]
        L.orc_u01.restype = C.c_float
        L.orc_u01.argtypes = [C.c_uint32]
        L.orc_parse_obj_real.restype = C.c_float
        L.orc_parse_obj_real.argtypes = [C.c_char_p]
        L.orc_scene_new.restype = C.c_void_p
        L.orc_scene_free.argtypes = [C.c_void_p]
        L.orc_scene_clear.argtypes = [C.c_void_p]
        L.orc_scene_load_xml.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.orc_scene_add.argtypes = [C.c_void_p, C.c_int, fp, fp, fp, fp]
        L.orc_scene_build_bvh.argtypes = [C.c_void_p]
        for n in ("orc_scene_prim_count", "orc_scene_triangle_count", "orc_scene_node_count"):
            getattr(L, n).restype = C.c_uint64
            getattr(L, n).argtypes = [C.c_void_p]
        L.orc_scene_log.restype = C.c_char_p
        L.orc_scene_log.argtypes = [C.c_void_p]
        L.orc_scene_pack_prims.argtypes = [C.c_void_p, fp]
        L.orc_scene_pack_mats.argtypes = [C.c_void_p, fp]
        L.orc_scene_pack_bvh.argtypes = [C.c_void_p, fp]
        L.orc_scene_pack_prim_idx.argtypes = [C.c_void_p, ip]
        L.orc_viewport.argtypes = [fp, fp, fp, C.c_float, C.c_float, C.c_float, C.POINTER(Uniforms)]
        L.orc_render.argtypes = [C.POINTER(Uniforms), C.POINTER(RenderParams), fp, fp, fp, ip, fp, fp, u64p]
        L.orc_render_mt.argtypes = L.orc_render.argtypes + [C.c_int]
        L.orc_first_hit.argtypes = [fp, fp, fp, fp, ip, fp, ip, fp, ip]
        L.orc_fnv1a64.restype = C.c_uint64
        L.orc_fnv1a64.argtypes = [C.c_void_p, C.c_uint64]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


class OracleScene:
    """Restated Scene (R/Scene/Scene.h) + SceneLoader (R/Scene/SceneLoader.cpp)."""

    def __init__(self):
        self.h = lib().orc_scene_new()

    def __del__(self):
        try:
            lib().orc_scene_free(self.h)
        except Exception:
            pass

    def load_xml(self, path, asset_root=None):
        return lib().orc_scene_load_xml(self.h, path.encode(), asset_root.encode() if asset_root else None)

    def add_sphere(self, center, radius, albedo=(0.8, 0.8, 0.8), mtype=0.0, emission=(0, 0, 0), power=0.0):
        mat = (C.c_float * 8)(*albedo, mtype, *emission, power)
        lib().orc_scene_add(self.h, 0, _f3(center), _f3((radius, 0, 0)), _f3((0, 0, 0)), mat)

    def add_triangle(self, v0, v1, v2, albedo=(0.8, 0.8, 0.8), mtype=0.0, emission=(0, 0, 0), power=0.0):
        mat = (C.c_float * 8)(*albedo, mtype, *emission, power)
        lib().orc_scene_add(self.h, 1, _f3(v0), _f3(v1), _f3(v2), mat)

    def build_bvh(self):
        lib().orc_scene_build_bvh(self.h)

    @property
    def prim_count(self):
        return int(lib().orc_scene_prim_count(self.h))

    @property
    def triangle_count(self):
        return int(lib().orc_scene_triangle_count(self.h))

    @property
    def node_count(self):
        return int(lib().orc_scene_node_count(self.h))

    def buffers(self):
        """(bvh [N,2,4] f32, prims [P,3,4] f32, mats [P,2,4] f32, prim_idx [P] i32) — SURVEY App. D."""
        P, N = self.prim_count, self.node_count
        bvh = np.zeros((N, 2, 4), np.float32)
        prims = np.zeros((P, 3, 4), np.float32)
        mats = np.zeros((P, 2, 4), np.float32)
        idx = np.zeros((P,), np.int32)
        L = lib()
        L.orc_scene_pack_bvh(self.h, _fp(bvh))
        L.orc_scene_pack_prims(self.h, _fp(prims))
        L.orc_scene_pack_mats(self.h, _fp(mats))
        L.orc_scene_pack_prim_idx(self.h, _ip(idx))
        return bvh, prims, mats, idx


def camera_reset():
    """Camera::reset() values (R/Renderer/Camera.h:24-32)."""
    return dict(pos=(0.0, 20.0, 50.0), fwd=(0.0, 0.0, -1.0), up=(0.0, 1.0, 0.0), vfov=60.0)


def make_uniforms(W, H, prim_count, tri_count=0, cam=None, random_seed=(0.0, 0.0, 0.0), frame_count=1):
    cam = cam or camera_reset()
    u = Uniforms()
    lib().orc_viewport(_f3(cam["pos"]), _f3(cam["fwd"]), _f3(cam["up"]), float(cam["vfov"]), float(W), float(H),
                       C.byref(u))
    u.randomSeed[0], u.randomSeed[1], u.randomSeed[2] = [float(x) for x in random_seed]
    u.primitiveCount = prim_count
    u.triangleCount = tri_count
    u.frameCount = frame_count
    return u


def host_seed_sequence(n=3, state=92407235):
    """First n floats of the host randomFloat() stream (R/Renderer/Renderer.cpp:30-41)."""
    st = C.c_uint32(state)
    return [float(lib().orc_host_random_float(C.byref(st))) for _ in range(n)]


def render(u, buffers, rng_mode=RNG_PHILOX, bsdf_mode=BSDF_LAMBERT, max_depth=32, accumulate=1, sample_begin=0,
           sample_count=1, seed=(1, 0), last=None, out=None, threads=1, rows=None):
    """Run the oracle.  Returns (image [H,W,4] f32, counters dict)."""
    bvh, prims, mats, idx = buffers
    W, H = int(u.screenSize[0]), int(u.screenSize[1])
    rp = RenderParams(rng_mode, bsdf_mode, max_depth, accumulate, sample_begin, sample_count, seed[0], seed[1],
                      -1 if rows is None else rows[0], -1 if rows is None else rows[1])
    if out is None:
        out = np.zeros((H, W, 4), np.float32)
    ctr = np.zeros(12, np.uint64)
    lastp = _fp(last) if last is not None else None
    args = (C.byref(u), C.byref(rp), _fp(bvh), _fp(prims), _fp(mats), _ip(idx), lastp, _fp(out),
            ctr.ctypes.data_as(C.POINTER(C.c_uint64)))
    if threads > 1 and rows is None:
        lib().orc_render_mt(*args, int(threads))
    else:
        lib().orc_render(*args)
    return out, dict(zip(COUNTER_NAMES, [int(x) for x in ctr]))


def first_hit(o, d, buffers):
    bvh, prims, _mats, idx = buffers
    t = C.c_float()
    prim = C.c_int32()
    n = (C.c_float * 3)()
    ff = C.c_int32()
    lib().orc_first_hit(_f3(o), _f3(d), _fp(bvh), _fp(prims), _ip(idx), C.byref(t), C.byref(prim), n, C.byref(ff))
    return float(t.value), int(prim.value), (n[0], n[1], n[2]), bool(ff.value)


def fnv1a64(arr):
    a = np.ascontiguousarray(arr)
    return int(lib().orc_fnv1a64(a.ctypes.data_as(C.c_void_p), a.nbytes))
