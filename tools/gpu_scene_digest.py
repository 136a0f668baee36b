"""Digests of every device array mpt_build_and_upload makes (mpt_scene_digest), per scene and builder — what "the same arrays" means when
the builder's code changes.   usage: python tools/gpu_scene_digest.py [--write tests/golden/devbuild_digests.json] [--check FILE] [--time]"""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metalpathtracer_amd import capi, host
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import config4_scene
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def scenes(tmp):
    yield "scene.xml", os.path.join(ROOT, "assets", "scene.xml")
    yield "cornell.xml", os.path.join(ROOT, "assets", "cornell.xml")
    yield "glass.xml", os.path.join(ROOT, "assets", "glass.xml")
    yield "bunny20.xml", os.path.join(ROOT, "assets", "bunny20.xml")
    yield "config4", config4_scene.write(tmp)


def main():
    out = {}
    tmp = tempfile.mkdtemp()
    ctx = capi.Context(0)
    for name, xml in scenes(tmp):
        sc = host.Scene()
        st, log = host.SceneLoader.LoadSceneFromXML(xml, sc, os.path.join(ROOT, "assets"))
        assert st == 0, log
        prims, mats = sc.packed_primitives()
        for builder in ("sah", "ploc", "lbvh"):
            os.environ["MPT_GPU_BUILD"] = builder
            ctx.build_and_upload(prims, mats)
            d = ctx.scene_digest()
            best, ms = 1e9, 1e9
            for k in range(3 if "--time" in sys.argv else 1):
                t0 = time.perf_counter(); m = ctx.build_and_upload(prims, mats); best = min(best, time.perf_counter() - t0); ms = min(ms, m)
            assert ctx.scene_digest() == d, "%s %s: a rebuild gave other arrays" % (name, builder)
            out["%s/%s" % (name, builder)] = ["%016x" % v for v in d]
            print("%-18s %-5s %8d prims  wall %6.2f ms  device %6.2f ms  %s" % (name, builder, int(np.asarray(prims).size) // 12, best * 1e3, ms, " ".join("%016x" % v for v in d[:9])), flush=True)
    os.environ.pop("MPT_GPU_BUILD", None)
    ctx.close()
    if "--write" in sys.argv:
        json.dump(out, open(sys.argv[sys.argv.index("--write") + 1], "w"), indent=1, sort_keys=True)
    if "--check" in sys.argv:
        want = json.load(open(sys.argv[sys.argv.index("--check") + 1]))
        bad = [k for k in want if out.get(k) != want[k]]
        for k in bad:
            names = "nodes prims mats own refleaf refbox always ref_bvh ref_idx n_nodes n_prims n_mats n_own n_leaves n_always depth".split()
            print("DIFFERENT %s: %s" % (k, ", ".join(n for n, a, b in zip(names, out.get(k, [None] * 16), want[k]) if a != b)))
        print("all %d digests as recorded" % len(want) if not bad else "%d of %d differ" % (len(bad), len(want)))
        sys.exit(1 if bad else 0)


main()
