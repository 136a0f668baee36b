"""The shader clock the trace kernels actually hold (diagnostics build, -DMPT_CLOCK_STAMP): >= 2 s of back-to-back renders,
then delta s_memtime / delta s_memrealtime x 100 MHz summed over all waves (MI355X_MICROARCH.md, DVFS item 6).
usage: MPT_LIB=<libmpt_hip_clock.so> python tools/gpu_clock.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalpathtracer_amd import capi, host
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = capi.load()
buf = (C.c_ulonglong * 2)()
for name, bvh, pipe, tag in (("scene.xml", 3, capi.PIPE_WAVELOCAL, "k_wavelocal"), ("scene.xml", 0, capi.PIPE_WAVELOCAL, "k_wavelocal"),
                             ("scene.xml", 0, capi.PIPE_ORDERED, "k_ordered"), ("bunny20.xml", 3, capi.PIPE_ORDERED, "k_ordered"),
                             ("bunny20.xml", 3, capi.PIPE_WAVELOCAL, "k_wavelocal")):   # (bvh 3: the device build, bench.py's default)
    sc = host.Scene(); st, _ = host.SceneLoader.LoadSceneFromXML(os.path.join(ROOT, "assets", name), sc); assert st == 0
    ctx = capi.Context(0); host.make_ready(ctx, sc, bvh)
    W, H = 1920, 1080
    ctx.resize(W, H); ctx.set_uniforms(host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount()))
    kw = dict(rng_mode=capi.RNG_PHILOX, max_depth=8, sample_count=256, pipeline=pipe)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 2.0:      # warm the chip up under this kernel's load
        ctx.render(**kw)
    L.mpt_debug_clock(buf, 1)
    ms = []
    for k in range(8):
        ctx.render(**kw); ms.append(ctx.stats()["total_ms"])
    L.mpt_debug_clock(buf, 0)
    print("%-12s %-12s %-15s in-kernel clock %.3f GHz  (%.2f ms per 256-spp render)" % (tag, name, ("reference tree", "binned tree", "GPU tree", "device build")[bvh],
                                                                                       buf[0] / buf[1] * 0.1, sum(ms) / len(ms)), flush=True)
    ctx.close()
