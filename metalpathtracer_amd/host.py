"""Python mirror of the host layer (include/mpt_host.h, libmpt_host.so).

Class and method names follow the reference's C++ interface (R/Scene/Scene.h, R/Scene/SceneLoader.h,
R/Renderer/Renderer.h, R/Renderer/Camera.h) so the tests read like code written against the reference.
All work happens in the C++ library; nothing here computes.
"""
import ctypes as C
import os

import numpy as np

from . import capi

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libmpt_host.so")

BVH_REFERENCE_SWEEP, BVH_BINNED_CENTROID, BVH_GPU_LBVH = 0, 1, 2

SYMBOLS = (
    "mpt_scene_create", "mpt_scene_destroy", "mpt_scene_clear", "mpt_scene_load_xml", "mpt_scene_add_primitive",
    "mpt_scene_build_bvh", "mpt_scene_sort_primitives", "mpt_scene_counts", "mpt_scene_copy_buffers", "mpt_camera_reset_values",
    "mpt_camera_viewport", "mpt_host_random_float", "mpt_renderer_create", "mpt_renderer_destroy",
    "mpt_renderer_drawable_size_will_change", "mpt_renderer_set_params", "mpt_renderer_draw", "mpt_renderer_input",
    "mpt_renderer_read_frame", "mpt_renderer_render_batch", "mpt_renderer_read_sum", "mpt_renderer_clear_sum",
    "mpt_renderer_uniforms", "mpt_renderer_stats", "mpt_renderer_context", "mpt_renderer_scene", "mpt_write_pfm",
    "mpt_write_ppm",
)

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libmpt_host.so is not built (%s). Run `make` or __graft_entry__.build()." % LIB_PATH)
    capi.load()  # libmpt_host.so links libmpt_hip.so
    L = C.CDLL(LIB_PATH)
    vp, fp, ip = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32)
    L.mpt_scene_create.argtypes = [C.POINTER(vp)]
    L.mpt_scene_destroy.argtypes = [vp]
    L.mpt_scene_clear.argtypes = [vp]
    L.mpt_scene_load_xml.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]
    L.mpt_scene_add_primitive.argtypes = [vp, C.c_int, fp, fp, fp, fp]
    L.mpt_scene_build_bvh.argtypes = [vp, C.c_int]
    L.mpt_scene_sort_primitives.argtypes = [vp]
    L.mpt_scene_counts.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                   C.POINTER(C.c_int32)]
    L.mpt_scene_copy_buffers.argtypes = [vp, fp, fp, fp, ip]
    L.mpt_camera_reset_values.argtypes = [fp, fp, fp, fp]
    L.mpt_camera_viewport.argtypes = [fp, fp, fp, C.c_float, C.c_float, C.c_float, C.POINTER(capi.Uniforms)]
    L.mpt_host_random_float.argtypes = [C.POINTER(C.c_uint32)]
    L.mpt_host_random_float.restype = C.c_float
    L.mpt_renderer_create.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.POINTER(vp), C.c_char_p, C.c_size_t]
    L.mpt_renderer_destroy.argtypes = [vp]
    L.mpt_renderer_drawable_size_will_change.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.mpt_renderer_set_params.argtypes = [vp, C.POINTER(capi.RenderParams)]
    L.mpt_renderer_draw.argtypes = [vp]
    L.mpt_renderer_input.argtypes = [vp, fp, fp, C.c_float, C.c_int]
    L.mpt_renderer_read_frame.argtypes = [vp, fp]
    L.mpt_renderer_render_batch.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.mpt_renderer_read_sum.argtypes = [vp, fp]
    L.mpt_renderer_clear_sum.argtypes = [vp]
    L.mpt_renderer_uniforms.argtypes = [vp, C.POINTER(capi.Uniforms)]
    L.mpt_renderer_stats.argtypes = [vp, C.POINTER(capi.Stats)]
    L.mpt_renderer_context.argtypes = [vp]
    L.mpt_renderer_context.restype = vp
    L.mpt_renderer_scene.argtypes = [vp]
    L.mpt_renderer_scene.restype = vp
    L.mpt_write_pfm.argtypes = [C.c_char_p, fp, C.c_uint32, C.c_uint32, C.c_float]
    L.mpt_write_ppm.argtypes = [C.c_char_p, fp, C.c_uint32, C.c_uint32, C.c_float, C.c_float]
    _lib = L
    return L


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


class Scene:
    """class Scene (R/Scene/Scene.h:34-188)."""

    def __init__(self, _borrowed=None):
        self.L = load()
        self._owned = _borrowed is None
        if self._owned:
            self.h = C.c_void_p()
            self.L.mpt_scene_create(C.byref(self.h))
        else:
            self.h = C.c_void_p(_borrowed)

    def __del__(self):
        try:
            if self._owned and self.h:
                self.L.mpt_scene_destroy(self.h)
        except Exception:
            pass

    def clear(self):
        self.L.mpt_scene_clear(self.h)

    def addPrimitive(self, type_, d0, d1, d2, albedo=(0.8, 0.8, 0.8), materialType=0.0, emission=(0, 0, 0),
                     emissionPower=0.0):
        mat = (C.c_float * 8)(*albedo, materialType, *emission, emissionPower)
        rc = self.L.mpt_scene_add_primitive(self.h, int(type_), _f3(d0), _f3(d1), _f3(d2), mat)
        if rc:
            raise ValueError("mpt_scene_add_primitive: %d" % rc)

    def addSphere(self, center, radius, **kw):
        self.addPrimitive(0, center, (radius, 0, 0), (0, 0, 0), **kw)

    def addTriangle(self, v0, v1, v2, **kw):
        self.addPrimitive(1, v0, v1, v2, **kw)

    def buildBVH(self, mode=BVH_REFERENCE_SWEEP):
        rc = self.L.mpt_scene_build_bvh(self.h, int(mode))
        if rc:
            raise ValueError("mpt_scene_build_bvh: %d" % rc)

    def sortPrimitives(self):
        """Scene::sortPrimitives: spheres first, stable — all that mpt_build_and_upload needs of buildBVH."""
        rc = self.L.mpt_scene_sort_primitives(self.h)
        if rc:
            raise ValueError("mpt_scene_sort_primitives: %d" % rc)

    def _counts(self):
        p, t, n, d = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_int32()
        self.L.mpt_scene_counts(self.h, C.byref(p), C.byref(t), C.byref(n), C.byref(d))
        return p.value, t.value, n.value, d.value

    def getPrimitiveCount(self):
        return self._counts()[0]

    def getTriangleCount(self):
        return self._counts()[1]

    def getSphereCount(self):
        c = self._counts()
        return c[0] - c[1]

    def getBVHNodeCount(self):
        return self._counts()[2]

    def getBVHDepth(self):
        return self._counts()[3]

    def packed_primitives(self):
        """(prims [P,3,4], mats [P,2,4]) without a tree: what mpt_build_and_upload takes (createTransformsBuffer /
        createMaterialsBuffer, R/Scene/Scene.h:99-133)."""
        P = self._counts()[0]
        prims = np.zeros((P, 3, 4), np.float32)
        mats = np.zeros((P, 2, 4), np.float32)
        self.L.mpt_scene_copy_buffers(self.h, None, _fp(prims), _fp(mats), None)
        return prims, mats

    def buffers(self):
        """(bvh [N,2,4], prims [P,3,4], mats [P,2,4], prim_idx [P]) — createBVHBuffer / createTransformsBuffer /
        createMaterialsBuffer / createPrimitiveIndexBuffer (R/Scene/Scene.h:99-167)."""
        P, _, N, _ = self._counts()
        bvh = np.zeros((N, 2, 4), np.float32)
        prims = np.zeros((P, 3, 4), np.float32)
        mats = np.zeros((P, 2, 4), np.float32)
        idx = np.zeros((P,), np.int32)
        self.L.mpt_scene_copy_buffers(self.h, _fp(bvh), _fp(prims), _fp(mats), idx.ctypes.data_as(C.POINTER(C.c_int32)))
        return bvh, prims, mats, idx


class SceneLoader:
    """SceneLoader::LoadSceneFromXML (R/Scene/SceneLoader.h:11)."""

    @staticmethod
    def LoadSceneFromXML(path, scene, asset_root=None):
        log = C.create_string_buffer(1 << 16)
        st = load().mpt_scene_load_xml(scene.h, path.encode(), asset_root.encode() if asset_root else None, log,
                                       len(log))
        return st, log.value.decode(errors="replace")


BVH_DEVICE = 3   # make_ready only: build -> render on the device (mpt_build_and_upload); 0..2 are Scene::buildBVH's modes


def make_ready(ctx, scene, bvh=BVH_REFERENCE_SWEEP):
    """The scene ready to render on `ctx` with the tree builder `bvh`: Scene::buildBVH + mpt_upload_scene for the host
    builders (0 reference sweep, 1 binned SAH, 2 GPU tree copied through the host), mpt_build_and_upload for BVH_DEVICE.
    Returns the (bvh, prims, mats, prim_idx) arrays in the reference's buffer format — for BVH_DEVICE the tree comes back
    from the device (mpt_download_bvh): what the oracle walks to render the same image."""
    if bvh == BVH_DEVICE:
        scene.sortPrimitives()                       # spheres first, as every builder does; no host tree is built
        prims, mats = scene.packed_primitives()
        ctx.build_and_upload(prims, mats)
        tree, idx = ctx.download_bvh()
        return tree, prims, mats, idx
    scene.buildBVH(bvh)
    buffers = scene.buffers()
    ctx.upload_scene(*buffers)
    return buffers


def camera_reset():
    """Camera::reset() (R/Renderer/Camera.h:24-32)."""
    pos, fwd, up = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 3)()
    fov = C.c_float()
    load().mpt_camera_reset_values(pos, fwd, up, C.byref(fov))
    return dict(pos=tuple(pos), fwd=tuple(fwd), up=tuple(up), vfov=fov.value)


def make_uniforms(W, H, prim_count, tri_count=0, cam=None, random_seed=(0.0, 0.0, 0.0), frame_count=1):
    """Uniforms as Renderer::recalculateViewport + updateUniforms fill them (R/Renderer/Renderer.cpp:153-182,251-267)."""
    cam = cam or camera_reset()
    u = capi.Uniforms()
    rc = load().mpt_camera_viewport(_f3(cam["pos"]), _f3(cam["fwd"]), _f3(cam["up"]), float(cam["vfov"]), float(W),
                                    float(H), C.byref(u))
    if rc:
        raise ValueError("mpt_camera_viewport: %d" % rc)
    for i in range(3):
        u.randomSeed[i] = float(random_seed[i])
    u.primitiveCount = prim_count
    u.triangleCount = tri_count
    u.frameCount = frame_count
    return u


def host_seed_sequence(n=3, state=92407235):
    st = C.c_uint32(state)
    return [float(load().mpt_host_random_float(C.byref(st))) for _ in range(n)]


class Renderer:
    """class Renderer (R/Renderer/Renderer.h:16-29) over an offscreen target."""

    def __init__(self, device=0, scene_xml=None, asset_root=None):
        self.L = load()
        self.h = C.c_void_p()
        err = C.create_string_buffer(1024)
        rc = self.L.mpt_renderer_create(int(device), scene_xml.encode() if scene_xml else None,
                                        asset_root.encode() if asset_root else None, C.byref(self.h), err, len(err))
        if rc:
            raise capi.MptError(rc, "Renderer", err.value.decode(errors="replace"))
        self.width, self.height = 1280, 720  # R/Renderer/Renderer.cpp:49

    def close(self):
        if getattr(self, "h", None):
            self.L.mpt_renderer_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, where):
        if rc:
            raise capi.MptError(rc, where)

    def drawableSizeWillChange(self, width, height):
        self._chk(self.L.mpt_renderer_drawable_size_will_change(self.h, int(width), int(height)),
                  "drawableSizeWillChange")
        self.width, self.height = int(width), int(height)

    def setRenderParams(self, **kw):
        p = capi.Context.params(**kw)
        self._chk(self.L.mpt_renderer_set_params(self.h, C.byref(p)), "setRenderParams")

    def draw(self):
        self._chk(self.L.mpt_renderer_draw(self.h), "draw")

    def input(self, move=(0, 0, 0), rotate=(0, 0), zoom=0.0, reset=False):
        """InputSystem state for the next draw (movementInput, rotationInput, zoomInput, resetInput)."""
        self._chk(self.L.mpt_renderer_input(self.h, _f3(move), (C.c_float * 2)(*[float(x) for x in rotate]), float(zoom),
                                            int(bool(reset))), "input")

    def readFrame(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        self._chk(self.L.mpt_renderer_read_frame(self.h, _fp(out)), "readFrame")
        return out

    def renderBatch(self, sample_begin, sample_count):
        self._chk(self.L.mpt_renderer_render_batch(self.h, int(sample_begin), int(sample_count)), "renderBatch")

    def readSum(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        self._chk(self.L.mpt_renderer_read_sum(self.h, _fp(out)), "readSum")
        return out

    def clearSum(self):
        self._chk(self.L.mpt_renderer_clear_sum(self.h), "clearSum")

    def uniforms(self):
        u = capi.Uniforms()
        self._chk(self.L.mpt_renderer_uniforms(self.h, C.byref(u)), "uniforms")
        return u

    def stats(self):
        s = capi.Stats()
        self._chk(self.L.mpt_renderer_stats(self.h, C.byref(s)), "stats")
        return s.as_dict()

    def scene(self):
        return Scene(_borrowed=self.L.mpt_renderer_scene(self.h))


def write_pfm(path, rgba, scale=1.0):
    a = np.ascontiguousarray(rgba, np.float32)
    return load().mpt_write_pfm(path.encode(), _fp(a), a.shape[1], a.shape[0], float(scale))


def write_ppm(path, rgba, scale=1.0, gamma=2.2):
    a = np.ascontiguousarray(rgba, np.float32)
    return load().mpt_write_ppm(path.encode(), _fp(a), a.shape[1], a.shape[0], float(scale), float(gamma))
