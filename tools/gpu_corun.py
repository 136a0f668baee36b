"""What can run BESIDE a resident trace kernel?  One 256-spp render of scene.xml is started with mpt_render_async; 3 ms later small torch
kernels of different shapes are launched on another stream and timed with events (GPU time from launch to completion)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metalpathtracer_amd import capi, host
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sc = host.Scene(); st, _ = host.SceneLoader.LoadSceneFromXML(os.path.join(ROOT, "assets", "scene.xml"), sc); assert st == 0
ctx = capi.Context(0); host.make_ready(ctx, sc, host.BVH_DEVICE)
W, H = 1920, 1080
ctx.resize(W, H); ctx.set_uniforms(host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount()))
kw = dict(rng_mode=capi.RNG_PHILOX, max_depth=8, sample_count=256)
ctx.render(**kw)
side = torch.cuda.Stream()
cases = {"1 element": torch.zeros(1, device="cuda"), "64 K elements": torch.zeros(1 << 16, device="cuda"),
         "16 M elements (64 MB)": torch.zeros(1 << 24, device="cuda"), "256 M elements (1 GB)": torch.zeros(1 << 28, device="cuda")}
for name, x in cases.items():
    with torch.cuda.stream(side):
        x.add_(1.0)
    torch.cuda.synchronize()
    res = []
    for beside in (False, True):
        if beside:
            ctx.render_async(**kw)
            time.sleep(0.003)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        with torch.cuda.stream(side):
            e0.record(); x.add_(1.0); e1.record()
        e1.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        res.append((e0.elapsed_time(e1), wall))
        if beside: ctx.wait()
    print("x.add_(1) on %-24s alone: %.3f ms (wall %.3f)   beside the trace kernel: %.3f ms (wall %.3f)" % (name, res[0][0], res[0][1], res[1][0], res[1][1]), flush=True)
