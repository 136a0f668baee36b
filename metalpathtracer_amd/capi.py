"""ctypes binding of the C ABI declared in include/mpt.h (libmpt_hip.so).

The HIP library is the product; there is no CPU fallback.  Loading fails loudly when the
library has not been built, and creating a context fails loudly when no GPU is present.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MPT_LIB") or os.path.join(_PKG, "lib", "libmpt_hip.so")

RNG_LITERAL, RNG_PHILOX = 0, 1
BSDF_LAMBERT, BSDF_SCATTER, BSDF_SCATTER_ALL = 0, 1, 2
PIPE_WAVEFRONT, PIPE_MEGAKERNEL, PIPE_WAVELOCAL, PIPE_ORDERED, PIPE_AUTO = 0, 1, 2, 3, 4
REFERENCE_ORDER_PIPELINES = (PIPE_WAVEFRONT, PIPE_MEGAKERNEL, PIPE_WAVELOCAL)  # walk the BVH in the reference's own order
DEFAULT_PIPELINE = PIPE_AUTO  # closest-first for big scenes, reference-order wave-local below 8192 primitives
FLAG_COUNT_WORK = 1

STATUS = {0: "MPT_OK", 1: "MPT_ERR_INVALID_ARG", 2: "MPT_ERR_NO_DEVICE", 3: "MPT_ERR_HIP",
          4: "MPT_ERR_BAD_SCENE", 5: "MPT_ERR_NOT_READY", 6: "MPT_ERR_OVERFLOW"}

# every symbol include/mpt.h declares (tests/test_capi_symbols.py checks header <-> library <-> this list)
SYMBOLS = (
    "mpt_create", "mpt_destroy", "mpt_last_error", "mpt_status_string", "mpt_upload_scene", "mpt_set_uniforms",
    "mpt_resize", "mpt_draw", "mpt_render", "mpt_render_async", "mpt_wait", "mpt_async_info", "mpt_sum_buffer", "mpt_set_sum_buffer", "mpt_clear_sum",
    "mpt_read_frame", "mpt_read_sum", "mpt_write_sum", "mpt_get_stats", "mpt_reset_stats", "mpt_stream", "mpt_synchronize",
    "mpt_trace_rays", "mpt_trace_rays_ordered", "mpt_accel_info", "mpt_kat_pcg", "mpt_kat_philox", "mpt_kat_sincos", "mpt_kat_rcp",
    "mpt_build_bvh", "mpt_build_and_upload", "mpt_download_bvh", "mpt_gpu_leaf_max", "mpt_build_info", "mpt_scene_digest", "mpt_comm_unique_id", "mpt_comm_create_all", "mpt_comm_create_rank", "mpt_reduce_sum", "mpt_comm_destroy",
    "mpt_comm_last_error",
)


class MptError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        super().__init__("%s failed: %s%s" % (where, STATUS.get(status, status), (" — " + detail) if detail else ""))


class Uniforms(C.Structure):
    """mpt_uniforms == UniformsData, 144 bytes (R/Renderer/Shaders/Structs.h:23-41)."""
    _fields_ = [
        ("primitiveIndex", C.c_int32), ("_pad0", C.c_int32 * 3),
        ("cameraPosition", C.c_float * 4),
        ("screenSize", C.c_float * 2), ("_pad1", C.c_float * 2),
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
        ("rng_mode", C.c_int32), ("bsdf_mode", C.c_int32), ("max_depth", C.c_int32), ("pipeline", C.c_int32),
        ("sample_begin", C.c_uint32), ("sample_count", C.c_uint32),
        ("seed_lo", C.c_uint32), ("seed_hi", C.c_uint32),
        ("shard_rank", C.c_int32), ("shard_count", C.c_int32),
        ("slots_per_iter", C.c_uint32), ("flags", C.c_uint32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("paths", C.c_uint64), ("rays", C.c_uint64), ("node_visits", C.c_uint64), ("aabb_hits", C.c_uint64),
        ("prim_tests", C.c_uint64), ("iterations", C.c_uint64),
        ("trace_kernel_ms", C.c_double), ("total_ms", C.c_double), ("trace_launches", C.c_uint64),
        ("wave_node_iters", C.c_uint64), ("wave_prim_iters", C.c_uint64), ("wave_leaf_phases", C.c_uint64),
        ("exact_retraces", C.c_uint64), ("tree_parked", C.c_uint64),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


_lib = None


def knob_env():
    """The MPT_* environment knobs that change what the device runs, i.e. part of a measured workload."""
    return {k: v for k, v in sorted(os.environ.items())
            if k.startswith("MPT_") and k != "MPT_LIB" and not k.startswith("MPT_BENCH_") and k != "MPT_CPU_THREADS"}


def build_id():
    """What ties a committed counter profile (profiles/*.json) to the build it was taken from: the sha256 of the library that
    is loaded and the sha256 of the sources it is built from (kernels, host side, ABI header, compiler flags) — the second
    survives a rebuild on another machine."""
    import hashlib
    import re
    root = os.path.dirname(_PKG)
    h = hashlib.sha256()
    csrc = os.path.join(_PKG, "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".h", ".hip")))
    files.append(os.path.join(root, "include", "mpt.h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read())
    m = re.search(r"^HIPFLAGS\s*\?=.*$", open(os.path.join(root, "Makefile")).read(), flags=re.M)
    h.update((m.group(0) if m else "").encode())
    lib = hashlib.sha256(open(LIB_PATH, "rb").read()).hexdigest() if os.path.exists(LIB_PATH) else None
    return {"lib_sha256": lib, "source_sha256": h.hexdigest()}


def load():
    """dlopen libmpt_hip.so and declare prototypes.  Raises if the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libmpt_hip.so is not built (%s). Run `make` or __graft_entry__.build(); "
                          "there is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, fp, ip, up = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint32)
    L.mpt_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.mpt_destroy.argtypes = [vp]
    L.mpt_last_error.argtypes = [vp]
    L.mpt_last_error.restype = C.c_char_p
    L.mpt_status_string.argtypes = [C.c_int]
    L.mpt_status_string.restype = C.c_char_p
    L.mpt_upload_scene.argtypes = [vp, fp, C.c_uint64, fp, fp, ip, C.c_uint64]
    L.mpt_set_uniforms.argtypes = [vp, C.POINTER(Uniforms)]
    L.mpt_resize.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.mpt_draw.argtypes = [vp, C.POINTER(RenderParams)]
    L.mpt_render.argtypes = [vp, C.POINTER(RenderParams)]
    L.mpt_render_async.argtypes = [vp, C.POINTER(RenderParams)]
    L.mpt_wait.argtypes = [vp]
    L.mpt_sum_buffer.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_uint64)]
    L.mpt_set_sum_buffer.argtypes = [vp, vp]
    L.mpt_clear_sum.argtypes = [vp]
    L.mpt_read_frame.argtypes = [vp, fp]
    L.mpt_read_sum.argtypes = [vp, fp]
    L.mpt_write_sum.argtypes = [vp, fp]
    L.mpt_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.mpt_reset_stats.argtypes = [vp]
    L.mpt_stream.argtypes = [vp]
    L.mpt_stream.restype = vp
    L.mpt_synchronize.argtypes = [vp]
    L.mpt_trace_rays.argtypes = [vp, fp, fp, C.c_uint64, fp, ip, fp, ip]
    L.mpt_trace_rays_ordered.argtypes = [vp, fp, fp, C.c_uint64, fp, ip, fp, ip, up]
    L.mpt_accel_info.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.mpt_build_bvh.argtypes = [vp, fp, C.c_uint64, fp, C.c_uint64, C.POINTER(C.c_uint64), ip, C.POINTER(C.c_double)]
    L.mpt_build_and_upload.argtypes = [vp, fp, fp, C.c_uint64, C.POINTER(C.c_double)]
    L.mpt_download_bvh.argtypes = [vp, fp, C.c_uint64, C.POINTER(C.c_uint64), ip]
    L.mpt_gpu_leaf_max.argtypes = [C.c_uint64]
    L.mpt_build_info.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.mpt_scene_digest.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.mpt_comm_unique_id.argtypes = [vp]
    L.mpt_comm_create_all.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(vp)]
    L.mpt_comm_create_rank.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(vp)]
    L.mpt_reduce_sum.argtypes = [vp, C.c_int]
    L.mpt_comm_destroy.argtypes = [vp]
    L.mpt_comm_last_error.argtypes = [vp]
    L.mpt_comm_last_error.restype = C.c_char_p
    L.mpt_kat_pcg.argtypes = [vp, up, C.c_uint64, up, fp]
    L.mpt_kat_philox.argtypes = [vp, up, up, C.c_uint64, up]
    L.mpt_kat_sincos.argtypes = [vp, fp, C.c_uint64, fp, fp]
    L.mpt_kat_rcp.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.mpt_async_info.argtypes = [vp, C.POINTER(C.c_uint64)]
    _lib = L
    return L


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _up(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def gpu_leaf_max(n_prims):
    """mpt_gpu_leaf_max: the leaf limit of the GPU builders for a scene of n_prims primitives (6 below 8192, 2 from there on)."""
    return int(load().mpt_gpu_leaf_max(int(n_prims)))


class Comm:
    """RCCL communicator over the C ABI (mpt_comm_*): `Comm.all([ctx0, ctx1, ...])` for N contexts driven by this
    process, or `Comm.rank(ctx, rank, nranks, id_bytes)` with `Comm.unique_id()` from rank 0 for one process per GPU."""

    def __init__(self, handle, ctxs):
        self.L = load()
        self.h = handle
        self.ctxs = ctxs

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        rc = load().mpt_comm_unique_id(buf)
        if rc:
            raise MptError(rc, "mpt_comm_unique_id")
        return buf.raw

    @classmethod
    def all(cls, ctxs):
        L = load()
        arr = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
        h = C.c_void_p()
        rc = L.mpt_comm_create_all(arr, len(ctxs), C.byref(h))
        if rc:
            raise MptError(rc, "mpt_comm_create_all", (L.mpt_last_error(ctxs[0].h) or b"").decode())
        return cls(h, list(ctxs))

    @classmethod
    def rank(cls, ctx, rank, nranks, id_bytes=None):
        L = load()
        h = C.c_void_p()
        buf = C.create_string_buffer(id_bytes, 128) if id_bytes else None
        rc = L.mpt_comm_create_rank(ctx.h, int(rank), int(nranks), buf, C.byref(h))
        if rc:
            raise MptError(rc, "mpt_comm_create_rank", (L.mpt_last_error(ctx.h) or b"").decode())
        return cls(h, [ctx])

    def reduce_sum(self, root=0):
        rc = self.L.mpt_reduce_sum(self.h, int(root))
        if rc:
            raise MptError(rc, "mpt_reduce_sum", (self.L.mpt_comm_last_error(self.h) or b"").decode())

    def close(self):
        if self.h:
            self.L.mpt_comm_destroy(self.h)
            self.h = None


class Context:
    """Owns one mpt_ctx (one GPU, one stream).  Not thread-safe, like the reference Renderer."""

    def __init__(self, device=0):
        self.L = load()
        self.h = C.c_void_p()
        rc = self.L.mpt_create(int(device), C.byref(self.h))
        if rc:
            raise MptError(rc, "mpt_create", "a MI355X GPU is required; there is no CPU fallback")
        self.width = self.height = 0

    def close(self):
        if getattr(self, "h", None):
            self.L.mpt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, where):
        if rc:
            raise MptError(rc, where, (self.L.mpt_last_error(self.h) or b"").decode())

    def upload_scene(self, bvh, prims, mats, prim_idx):
        bvh = np.ascontiguousarray(bvh, np.float32)
        prims = np.ascontiguousarray(prims, np.float32)
        mats = np.ascontiguousarray(mats, np.float32)
        prim_idx = np.ascontiguousarray(prim_idx, np.int32)
        n_nodes = bvh.size // 8
        n_prims = prims.size // 12
        if mats.size != n_prims * 8 or prim_idx.size != n_prims:
            raise ValueError("scene arrays disagree on the primitive count")
        self._chk(self.L.mpt_upload_scene(self.h, _fp(bvh), n_nodes, _fp(prims), _fp(mats), _ip(prim_idx), n_prims),
                  "mpt_upload_scene")

    def set_uniforms(self, u):
        self._chk(self.L.mpt_set_uniforms(self.h, C.byref(u)), "mpt_set_uniforms")

    def resize(self, w, h):
        self._chk(self.L.mpt_resize(self.h, int(w), int(h)), "mpt_resize")
        self.width, self.height = int(w), int(h)

    @staticmethod
    def params(rng_mode=RNG_PHILOX, bsdf_mode=BSDF_LAMBERT, max_depth=32, pipeline=DEFAULT_PIPELINE, sample_begin=0,
               sample_count=1, seed=(1, 0), shard_rank=0, shard_count=1, slots_per_iter=0, flags=0):
        return RenderParams(rng_mode, bsdf_mode, max_depth, pipeline, sample_begin, sample_count, seed[0], seed[1],
                            shard_rank, shard_count, slots_per_iter, flags)

    def draw(self, **kw):
        p = self.params(**kw)
        self._chk(self.L.mpt_draw(self.h, C.byref(p)), "mpt_draw")

    def render(self, **kw):
        p = self.params(**kw)
        self._chk(self.L.mpt_render(self.h, C.byref(p)), "mpt_render")

    def render_async(self, **kw):
        """Enqueue a render and return; up to two overlap on the device.  wait() collects them (and their stats)."""
        p = self.params(**kw)
        self._chk(self.L.mpt_render_async(self.h, C.byref(p)), "mpt_render_async")

    def wait(self):
        self._chk(self.L.mpt_wait(self.h), "mpt_wait")

    def async_info(self):
        out = (C.c_uint64 * 4)()
        self._chk(self.L.mpt_async_info(self.h, out), "mpt_async_info")
        return dict(zip(("submitted", "gate_resident", "gate_timeout", "call_us_max"), [int(v) for v in out]))

    def clear_sum(self):
        self._chk(self.L.mpt_clear_sum(self.h), "mpt_clear_sum")

    def sum_buffer(self):
        p, n = C.c_void_p(), C.c_uint64()
        self._chk(self.L.mpt_sum_buffer(self.h, C.byref(p), C.byref(n)), "mpt_sum_buffer")
        return p.value, n.value

    def set_sum_buffer(self, device_ptr):
        self._chk(self.L.mpt_set_sum_buffer(self.h, C.c_void_p(device_ptr)), "mpt_set_sum_buffer")

    def read_frame(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        self._chk(self.L.mpt_read_frame(self.h, _fp(out)), "mpt_read_frame")
        return out

    def read_sum(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        self._chk(self.L.mpt_read_sum(self.h, _fp(out)), "mpt_read_sum")
        return out

    def write_sum(self, rgba):
        """Put an HDR sum back (checkpoint / resume): the inverse of read_sum."""
        a = np.ascontiguousarray(rgba, np.float32)
        if a.shape != (self.height, self.width, 4):
            raise ValueError("write_sum: expected an array of shape (%d, %d, 4)" % (self.height, self.width))
        self._chk(self.L.mpt_write_sum(self.h, _fp(a)), "mpt_write_sum")

    def stats(self):
        s = Stats()
        self._chk(self.L.mpt_get_stats(self.h, C.byref(s)), "mpt_get_stats")
        return s.as_dict()

    def reset_stats(self):
        self._chk(self.L.mpt_reset_stats(self.h), "mpt_reset_stats")

    def synchronize(self):
        self._chk(self.L.mpt_synchronize(self.h), "mpt_synchronize")

    def stream(self):
        return self.L.mpt_stream(self.h)

    def trace_rays(self, origins, directions):
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, np.float32).reshape(-1, 3)
        n = o.shape[0]
        t = np.empty(n, np.float32)
        prim = np.empty(n, np.int32)
        nrm = np.empty((n, 3), np.float32)
        front = np.empty(n, np.int32)
        self._chk(self.L.mpt_trace_rays(self.h, _fp(o), _fp(d), n, _fp(t), _ip(prim), _fp(nrm), _ip(front)),
                  "mpt_trace_rays")
        return t, prim, nrm, front

    def trace_rays_ordered(self, origins, directions):
        """Closest hit through the closest-first walk of PIPE_ORDERED; also returns the per-ray re-trace flags."""
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, np.float32).reshape(-1, 3)
        n = o.shape[0]
        t = np.empty(n, np.float32)
        prim = np.empty(n, np.int32)
        nrm = np.empty((n, 3), np.float32)
        front = np.empty(n, np.int32)
        flags = np.empty(n, np.uint32)
        self._chk(self.L.mpt_trace_rays_ordered(self.h, _fp(o), _fp(d), n, _fp(t), _ip(prim), _fp(nrm), _ip(front),
                                                _up(flags)), "mpt_trace_rays_ordered")
        return t, prim, nrm, front, flags

    def build_bvh(self, prims):
        """GPU LBVH over the packed primitive array (12 floats each) -> (bvh [N, 8] f32, prim_idx [P] i32, device ms)."""
        prims = np.ascontiguousarray(prims, np.float32).reshape(-1, 12)
        n = prims.shape[0]
        bvh = np.zeros((2 * n - 1, 8), np.float32)
        idx = np.zeros(n, np.int32)
        nn, ms = C.c_uint64(), C.c_double()
        self._chk(self.L.mpt_build_bvh(self.h, _fp(prims), n, _fp(bvh), 2 * n - 1, C.byref(nn), _ip(idx), C.byref(ms)),
                  "mpt_build_bvh")
        return bvh[: nn.value].copy(), idx, ms.value

    def build_and_upload(self, prims, mats):
        """Build the BVH on the device and make it the scene, without a host round trip.  Returns the device ms."""
        prims = np.ascontiguousarray(prims, np.float32).reshape(-1, 12)
        mats = np.ascontiguousarray(mats, np.float32).reshape(-1, 8)
        assert prims.shape[0] == mats.shape[0]
        ms = C.c_double()
        self._chk(self.L.mpt_build_and_upload(self.h, _fp(prims), _fp(mats), prims.shape[0], C.byref(ms)), "mpt_build_and_upload")
        return ms.value

    def build_info(self):
        """mpt_build_info: what the last scene call left on the device (sizes come from the C API, not from this object)."""
        out = (C.c_uint64 * 8)()
        self._chk(self.L.mpt_build_info(self.h, out), "mpt_build_info")
        keys = ("built_prims", "built_nodes", "built_leaf_max", "auto_ordered_prims", "prims", "threaded_nodes", "materials", "unquantised_nodes")
        return dict(zip(keys, [int(v) for v in out]))

    def download_bvh(self):
        """The tree of the last build_and_upload in the reference's buffer format: (bvh [N, 2, 4] f32, prim_idx [P] i32).
        MPT_ERR_NOT_READY when the scene on the device did not come from build_and_upload."""
        info = self.build_info()
        n, nodes = info["built_prims"], max(1, info["built_nodes"])
        bvh = np.zeros((nodes, 2, 4), np.float32)
        idx = np.zeros(max(1, n), np.int32)
        nn = C.c_uint64()
        self._chk(self.L.mpt_download_bvh(self.h, _fp(bvh), nodes, C.byref(nn), _ip(idx)), "mpt_download_bvh")
        return bvh[: nn.value].copy(), idx[:n]

    def scene_digest(self):
        """16 words: digests of the scene's nine device arrays, then seven counts (include/mpt.h)"""
        out = (C.c_uint64 * 16)()
        self._chk(self.L.mpt_scene_digest(self.h, out), "mpt_scene_digest")
        return [int(v) for v in out]

    def accel_info(self):
        out = (C.c_uint64 * 8)()
        self._chk(self.L.mpt_accel_info(self.h, out), "mpt_accel_info")
        keys = ("ordered_ok", "nodes", "depth", "lds_nodes", "always_spheres", "reference_leaves", "lds_prims", "auto_pipeline")
        return dict(zip(keys, [int(v) for v in out]))

    def kat_pcg(self, seeds):
        s = np.ascontiguousarray(seeds, np.uint32)
        h = np.empty_like(s)
        f = np.empty(s.shape, np.float32)
        self._chk(self.L.mpt_kat_pcg(self.h, _up(s), s.size, _up(h), _fp(f)), "mpt_kat_pcg")
        return h, f

    def kat_philox(self, ctr, key):
        c = np.ascontiguousarray(ctr, np.uint32).reshape(-1, 4)
        k = np.ascontiguousarray(key, np.uint32).reshape(-1, 2)
        o = np.empty_like(c)
        self._chk(self.L.mpt_kat_philox(self.h, _up(c), _up(k), c.shape[0], _up(o)), "mpt_kat_philox")
        return o

    def kat_sincos(self, u):
        u = np.ascontiguousarray(u, np.float32)
        s = np.empty_like(u)
        c = np.empty_like(u)
        self._chk(self.L.mpt_kat_sincos(self.h, _fp(u), u.size, _fp(s), _fp(c)), "mpt_kat_sincos")
        return s, c

    def kat_rcp(self):
        """rcp_chain(x) vs the correctly rounded 1.0f / x over all 2^32 operands, on the device:
        (mismatches inside the range the kernels use the chain in, operands in it, mismatches outside, operands outside)."""
        out = (C.c_uint64 * 4)()
        self._chk(self.L.mpt_kat_rcp(self.h, out), "mpt_kat_rcp")
        return tuple(int(v) for v in out)
