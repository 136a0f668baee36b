"""Where the time of k_ordered goes (diagnostics build, -DMPT_OT_TIMES): cycles per region summed over all waves, steps and
lanes per step kind.   usage: MPT_LIB=<libmpt_hip_times.so> python tools/gpu_ot_times.py [scene.xml] [spp]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalpathtracer_amd import capi, host
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1] if len(sys.argv) > 1 else "scene.xml"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
W, H = 1920, 1080
shards, rank, bsdf, depth = 1, 0, 0, 8
if name == "config4":   # configs[4]: the 1 M-triangle scene, Scatter.h BSDFs, depth 16, one rank's 1/8 tile shard (tools/config4_scene.py)
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import config4_scene
    xml = config4_scene.write(tempfile.mkdtemp(prefix="mpt_cfg4_"))
    shards, rank, bsdf, depth = 8, 0, 1, 16
else:
    xml = os.path.join(ROOT, "assets", name)
sc = host.Scene(); st, _ = host.SceneLoader.LoadSceneFromXML(xml, sc); assert st == 0
ctx = capi.Context(0); host.make_ready(ctx, sc, int(os.environ.get("BVH", "0")))
ctx.resize(W, H); ctx.set_uniforms(host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount()))
L = capi.load()
buf = (C.c_ulonglong * 40)()
for rep in range(2):
    L.mpt_debug_ot_times(buf, 1)
    ctx.clear_sum(); ctx.reset_stats()
    ctx.render(rng_mode=capi.RNG_PHILOX, max_depth=depth, sample_count=spp, pipeline=capi.PIPE_ORDERED, bsdf_mode=bsdf, shard_rank=rank, shard_count=shards)
    s = ctx.stats()
    L.mpt_debug_ot_times(buf, 0)
v = list(buf)
tot = sum(v[:8])
names = ["select/claim", "fetch (gen / ring load)", "top test", "walk", "final check", "exact walk", "shade", "push"]
print("%s %d spp: %.2f ms, %d rays, retraced %d, parked %d" % (name, spp, s["total_ms"], s["rays"], s["exact_retraces"], s["tree_parked"]))
for n, c in zip(names, v[:8]):
    print("  %-26s %5.1f %%" % (n, 100.0 * c / tot))
kinds = ["ring R (fresh rays)", "ring E (reference order)", "ring M0 (tree walk)", "ring M1", "primary"]
for k in range(5):
    if v[8 + k]:
        print("  steps %-26s %10d  lanes/step %.1f" % (kinds[k], v[8 + k], v[16 + k] / v[8 + k]))
w = v[24:40]
if w[2]:
    print("  walk lane utilisation: node loop %.1f %%, leaf loop %.1f %%" % (100.0 * w[5] / (64.0 * w[2]), 100.0 * w[6] / (64.0 * max(1, w[3]))))
if w[2]:
    print("  walk: node loop %.1f %% of the walk cycles, %.0f cycles per wave trip (%d trips); leaf loop %.0f cycles per wave trip (%d trips); %d rounds"
          % (100.0 * w[0] / max(1, w[0] + w[1]), w[0] / w[2], w[2], w[1] / max(1, w[3]), w[3], w[4]))

if w[7] + w[8]:
    x = v[24 + 7:24 + 11]
    print("  served from LDS: %.1f %% of the node visits (%d of %d), %.1f %% of the primitive records (%d of %d)"
          % (100.0 * x[0] / (x[0] + x[1]), x[0], x[0] + x[1], 100.0 * x[2] / max(1, x[2] + x[3]), x[2], x[2] + x[3]))
if w[11]:
    print("  stack pops: %.2f entries examined per pop (lanes: %d pops); per wave-level call %.2f loop trips (%d calls) = %.2f LDS round trips per node trip + leaf trip"
          % (w[12] / w[11], w[11], w[14] / max(1, w[13]), w[13], w[14] / max(1, w[2] + w[3])))
out = os.environ.get("JSON_OUT")
if out:
    import json
    J = {"what": "k_ordered on %s 1920x1080, %d spp, depth 8: diagnostics build (-DMPT_OT_TIMES), s_memtime around the regions of every step summed over all "
                 "waves (the lane counters of this build perturb the timing: the percentages and the lane utilisation are what it is for)" % (name, spp),
         "build": capi.build_id(), "kernel_ms_instrumented": s["total_ms"], "rays": s["rays"], "exact_retraces": s["exact_retraces"], "tree_parked": s["tree_parked"],
         "cycles_by_region_pct": {n: 100.0 * c / tot for n, c in zip(names, v[:8])},
         "steps": {kinds[k]: {"steps": v[8 + k], "lanes_per_step": v[16 + k] / v[8 + k]} for k in range(5) if v[8 + k]},
         "walk": {"node_loop_lane_utilisation_pct": 100.0 * w[5] / (64.0 * max(1, w[2])), "leaf_loop_lane_utilisation_pct": 100.0 * w[6] / (64.0 * max(1, w[3])),
                  "node_trips": w[2], "leaf_trips": w[3], "rounds": w[4], "node_loop_share_of_walk_pct": 100.0 * w[0] / max(1, w[0] + w[1])}}
    json.dump(J, open(out, "w"), indent=1)
