"""After `gpurun -- 'bash tools/profiles_round.sh r03'`: turns gpurun_out/r03_* into the tracked files under profiles/.
usage: python tools/profiles_collect.py [tag] [--pmc-only]"""
import json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "r05"
pmc_only = "--pmc-only" in sys.argv   # (on the GPU box, between the counter passes and bench.py: the line can then name its profile)
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
notes = {"wl": ("pmc_wavelocal", "scene.xml 1920x1080, 256 spp, depth 8, device-built binned-SAH tree (leaves <= 6): the default bench.py step."),
         "wlref": ("pmc_wavelocal_reference_tree", "scene.xml 1920x1080, 256 spp, depth 8, the REFERENCE's own tree (Scene::buildBVH sweep SAH + mpt_upload_scene): bench.py's extra workload."),
         "cor": ("pmc_wavelocal_cornell", "cornell.xml 1920x1080, 256 spp, depth 8, bench.py's CORNELL_CAM, device-built tree: bench.py's extra workload."),
         "otb": ("pmc_ordered_bunny20", "bunny x20 (99,362 primitives, device-built binned-SAH tree) 1920x1080, 256 spp, depth 8, MPT_PIPE_ORDERED: bench.py's extra workload."),
         "c4": ("pmc_ordered_config4", "configs[4]: 1,000,003 primitives (tools/config4_scene.py), device-built tree, 1920x1080, 4096 spp, depth 16, Scatter.h BSDFs, rank 0's 1/8 tile shard, MPT_PIPE_ORDERED: bench.py's extra workload."),
         "ot": ("pmc_ordered", "scene.xml 1920x1080, 256 spp, depth 8, device-built tree, MPT_PIPE_ORDERED."),
         "wlb": ("pmc_wavelocal_bunny20", "bunny x20 (99,362 primitives, device-built binned-SAH tree) 1920x1080, 64 spp, depth 8, MPT_PIPE_WAVELOCAL.")}
for k, (name, note) in notes.items():
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_finalize.py"), tag + k, os.path.join(P, "%s_%s.json" % (tag, name)), note],
                          stdout=subprocess.DEVNULL)
if pmc_only:
    sys.exit(0)
mem = json.load(open(os.path.join(G, tag + "otb_mem.json")))
mem["note"] = ("memory-pipe counters of ONE k_ordered launch on bunny x20 (64 spp; tools/pmc_mem.sh: texture addresser and texture data unit at two counters a pass, vector L1 and "
               "L2 at four; one rocprofv3 --pmc pass per set, a failed pass fails the script).  TCP_TOTAL_CACHE_ACCESSES / TCP_TA_TCP_STATE_READ = tag look-ups per wave load instruction.")
json.dump(mem, open(os.path.join(P, tag + "_mem_ordered_bunny20.json"), "w"), indent=1)
for f in ("bench_kernel_stats.csv", "devbuild_kernel_stats.csv", "bench_under_rocprof.json", "bench.json", "step_table.json", "step_table.txt",
          "ot_times_bunny20.json", "ot_times_bunny20.txt", "ot_times_config4.txt", "ot_lds_share.txt", "draw_fps.txt", "inkernel_clock.txt", "devbuild.txt", "devbuild_digests.txt", "devbuild_timeline.txt", "shard_time.txt", "depth_work.txt", "configs.txt"):
    src = os.path.join(G, "%s_%s" % (tag, f))
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, "%s_%s" % (tag, f)))
    else:
        print("missing", src)
# the dominant kernel's launches of the profiled bench run, in order, with what each one was: the summary CSV averages the
# scene.xml steps together with the untimed extra workloads (Cornell, bunny x20), this separates them
import csv
trace = os.path.join(G, "bench_trace", "b_kernel_trace.csv")
if os.path.exists(trace):
    rows = [r for r in csv.DictReader(open(trace)) if "k_wavelocal" in r["Kernel_Name"] or "k_ordered" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    line = [l for l in open(os.path.join(G, tag + "_bench_under_rocprof.json")) if l.startswith("{")][-1]
    b = json.loads(line)
    K, Wm = b["steps"], b["warmup"]
    labels = ["lane set-up"] * 2 + ["warm-up"] * Wm + ["timed step"] * K + ["serial render"] * min(3, K) + ["extra: cornell.xml"] * 3 + ["extra: bunny20.xml"] * 3 + ["extra: scene.xml on the reference's tree"] * 3 + ["extra: config4 shard"] * 3
    t0 = int(rows[0]["Start_Timestamp"])
    launches = [{"kernel": r["Kernel_Name"].split("(")[0].replace("void ", ""), "ms": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6,
                 "start_ms": (int(r["Start_Timestamp"]) - t0) / 1e6, "end_ms": (int(r["End_Timestamp"]) - t0) / 1e6,
                 "what": labels[i] if i < len(labels) else "?"} for i, r in enumerate(rows)]
    timed = [x for x in launches if x["what"] == "timed step"]
    first = launches.index(timed[0]) if timed else 0
    ends = [launches[first - 1]["end_ms"]] + [x["end_ms"] for x in timed] if timed and first > 0 else []
    json.dump({"command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps %d --warmup %d" % (K, Wm),
               "note": "ms = End - Start of the kernel trace.  The timed steps are submitted while the launch before them is still running (mpt_render_async: "
                       "a launch is submitted as soon as the one before it is fully resident, DESIGN.md 6), and the trace's Start is the DISPATCH of the "
                       "launch, not the start of its first wave: for those launches End - Start includes the wait behind the resident kernel.  What the "
                       "chip did is the spacing of the ENDs (timed_steps_end_spacing_avg_ms) and the span the kernel stamps itself "
                       "(bench_line_avg_launch_ms: first workgroup's start to last wave's end on the 100 MHz clock); launches issued one at a time "
                       "(set-up, serial renders, extras) read the same in all three.",
               "timed_steps_avg_ms": sum(x["ms"] for x in timed) / max(1, len(timed)),
               "timed_steps_end_spacing_avg_ms": (ends[-1] - ends[0]) / (len(ends) - 1) if len(ends) > 1 else None,
               "bench_line_avg_launch_ms": b["roofline"]["avg_launch_ms"], "bench_line_ms_per_step": b["ms_per_step"],
               "launches": launches}, open(os.path.join(P, tag + "_bench_launches.json"), "w"), indent=1)
print(sorted(f for f in os.listdir(P) if f.startswith(tag)))
