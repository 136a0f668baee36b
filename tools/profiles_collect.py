"""After `gpurun -- 'bash tools/profiles_round.sh r03'`: turns gpurun_out/r03_* into the tracked files under profiles/.
usage: python tools/profiles_collect.py [tag]"""
import json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
notes = {"wl": ("pmc_wavelocal", "scene.xml 1920x1080, 256 spp, depth 8, reference tree: the default bench.py step."),
         "ot": ("pmc_ordered", "scene.xml 1920x1080, 256 spp, depth 8, reference tree, MPT_PIPE_ORDERED."),
         "otb": ("pmc_ordered_bunny20", "bunny x20 (99,362 primitives, host binned-SAH tree) 1920x1080, 64 spp, depth 8, MPT_PIPE_ORDERED."),
         "wlb": ("pmc_wavelocal_bunny20", "bunny x20 (99,362 primitives, host binned-SAH tree) 1920x1080, 64 spp, depth 8, MPT_PIPE_WAVELOCAL.")}
for k, (name, note) in notes.items():
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_finalize.py"), tag + k, os.path.join(P, "%s_%s.json" % (tag, name)), note],
                          stdout=subprocess.DEVNULL)
mem = json.load(open(os.path.join(G, tag + "otb_mem.json")))
mem["note"] = ("memory-pipe counters of ONE k_ordered launch on bunny x20 (64 spp; tools/pmc_mem.sh: texture addresser, vector L1, texture data, L2 requests; "
               "one rocprofv3 --pmc pass per set).  TCP_TOTAL_CACHE_ACCESSES / TCP_TA_TCP_STATE_READ = tag look-ups per wave load instruction.")
json.dump(mem, open(os.path.join(P, tag + "_mem_ordered_bunny20.json"), "w"), indent=1)
for f in ("bench_kernel_stats.csv", "devbuild_kernel_stats.csv", "bench_under_rocprof.json", "bench.json", "step_table.json", "step_table.txt",
          "ot_times_bunny20.json", "ot_times_bunny20.txt", "inkernel_clock.txt", "devbuild.txt", "shard_time.txt", "depth_work.txt", "configs.txt"):
    src = os.path.join(G, "%s_%s" % (tag, f))
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, "%s_%s" % (tag, f)))
    else:
        print("missing", src)
print(sorted(f for f in os.listdir(P) if f.startswith(tag)))
