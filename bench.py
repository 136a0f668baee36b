#!/usr/bin/env python3
"""bench.py — headline benchmark: Mrays/s on scene.xml at 1920x1080, depth 8, 256 spp (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over the whole 1920x1080 frame at --spp-per-step samples per pixel (default
256 = the headline config: one step is one complete 1920x1080x256spp render, a single kernel launch per GPU).  Scene
load, BVH build and upload happen before the timed region (inputs resident in HBM); the timed region holds the K render
steps and, for N > 1, the one RCCL reduce of the HDR framebuffer (the K steps accumulate into one HDR sum).  N > 1: one process per GPU, the 8x8 pixel tiles are
interleaved over the ranks (strong scaling: the job is fixed, `value` = all rays of the job / max-over-ranks time).

Rank 0 prints ONE JSON line.  Extra objects:
  roofline      dominant kernel vs the two roofs it can be measured against: vector-ALU instruction issue (wave64
                instructions x 2 cycles on a SIMD-32) and measured HBM traffic; `frac` is the larger of the two and
                cannot exceed 1.  The per-ray instruction / byte figures come from the committed rocprofv3 --pmc passes of
                the same build and workload (profiles/), scaled by the rays of the K timed steps over the WALL time of the
                timed region (the clock `value` is quoted on); per-launch and serial-render figures are secondary fields.
                The SURVEY.md 8(d) algorithmic bytes per ray are kept under `algorithmic` (they price LDS-served BVH
                reads as HBM and so exceed the HBM peak: not a bound).
  cpu_baseline  the CPU oracle (this repo's restatement of the reference algorithm; the reference has no CPU
                path) timed on the host cores on a bounded sample of the same workload (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
N_SIMD = 1024          # 256 CUs x 4 SIMD-32; a wave64 VALU instruction occupies its SIMD for 2 cycles (same guide)
MAX_CLOCK_GHZ = 2.4
N_CU = 256             # one scalar unit per CU: one SALU instruction per cycle, shared by the four SIMDs
B_QUEUE = 168.0        # compulsory wavefront-queue bytes per ray
PIPE_NAMES = {0: "wavefront (global SoA queues)", 1: "megakernel", 2: "wave-local wavefront, reference-order walk",
              3: "wave-local wavefront, closest-first walk of the own 4-wide BVH"}
PIPE_KERNEL = {0: "k_step", 1: "k_megakernel", 2: "k_wavelocal", 3: "k_ordered"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp-per-step", type=int, default=256)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--scene", default=os.path.join(ROOT, "assets", "scene.xml"))
    ap.add_argument("--pipeline", default=os.environ.get("MPT_BENCH_PIPELINE", "default"),
                    choices=["default", "wavefront", "megakernel", "wavelocal", "ordered"])
    ap.add_argument("--bvh", default="device", choices=["reference", "binned", "gpu", "device"],
                    help="tree builder: build -> render on the device (mpt_build_and_upload, the default: binned SAH, what the Renderer's "
                         "throughput mode does), the reference's sweep SAH (the drop-in tree), the host binned SAH, or the GPU builder through "
                         "the host (mpt_build_bvh + mpt_upload_scene)")
    ap.add_argument("--slots", type=int, default=0, help="wavefront width (ray slots per iteration), 0 = default")
    ap.add_argument("--cpu-spp", type=int, default=32, help="samples per pixel of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-workloads", action="store_true",
                    help="skip the untimed extras (Cornell box and bunny x20 at 1920x1080 x 256 spp, one serial render each)")
    return ap.parse_args()


def counter_profile(kernel, workload):
    """Per-launch counters of the dominant kernel from the committed rocprofv3 --pmc passes (tools/pmc_round.sh on
    tools/prof_run.py; PMC cannot be collected from inside this process).  A profile is used only if it was taken on THIS
    workload (scene, size, spp, depth, tree builder, shard, BSDF mode, knobs) with THIS build (library or source sha256,
    capi.build_id): its per-ray counters are then scaled by this run's rays and time (roofline_from_profile).  Returns (per-ray figures or
    None, {"profile_stale": true / "profile_missing": ..., ...}): nothing is guessed and nothing is swallowed."""
    import glob
    from metalpathtracer_amd import capi
    bid = capi.build_id()
    stale = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_*.json")), reverse=True):
        prof = json.load(open(f))          # (a malformed profile is an error, not a silent null)
        wl = prof.get("workload")
        if not wl or kernel not in prof.get("kernel", ""):
            continue
        if any(wl.get(k) != workload[k] for k in ("scene", "width", "height", "spp", "depth", "bvh", "env")):
            continue
        if wl.get("launch", "sync") != workload.get("launch", "sync"):   # mpt_render_async runs another variant of k_wavelocal than mpt_render
            continue
        if wl.get("shards", 1) != workload.get("shards", 1) or wl.get("bsdf", 0) != workload.get("bsdf", 0):   # (a 1/8 shard or Scatter.h BSDFs: other rays)
            continue
        rel = os.path.relpath(f, ROOT)
        if wl.get("source_sha256") != bid["source_sha256"] and wl.get("lib_sha256") != bid["lib_sha256"]:
            stale = stale or {"profile_stale": True, "profile": rel, "profile_source_sha256": wl.get("source_sha256"),
                              "this_source_sha256": bid["source_sha256"]}
            continue
        c, rays, ms = prof["counters"], float(prof["rays_per_launch"]), float(prof["kernel_ms"])
        return {
            "file": rel, "build": {k: wl.get(k) for k in ("lib_sha256", "source_sha256")},
            "valu_per_ray": c["SQ_INSTS_VALU"] / rays, "salu_per_ray": c["SQ_INSTS_SALU"] / rays,
            "lds_per_ray": c["SQ_INSTS_LDS"] / rays,
            "lane_utilisation": c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_INSTS_VALU"]),
            "hbm_bytes_per_ray": (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 / rays,   # gfx950: FETCH_SIZE x2
            "l2_hit_rate": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]),
            "clock_ghz": c["GRBM_GUI_ACTIVE"] / 8.0 / (ms * 1e-3) / 1e9,
            "wave_cycles_split": {k: c[n] / c["SQ_WAVE_CYCLES"] for k, n in
                                  (("issuing", "SQ_ACTIVE_INST_ANY"), ("waitcnt", "SQ_WAIT_ANY"), ("issue_stall", "SQ_WAIT_INST_ANY"))},
            "profiled_kernel_ms": ms,
        }, {}
    return None, stale or {"profile_missing": "no committed counter profile of %s for this workload" % kernel}


def host_cores():
    """Threads the CPU leg may use: the affinity mask, capped by the cgroup CPU quota when there is one."""
    n = max(1, len(os.sched_getaffinity(0)))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    cap = int(os.environ.get("MPT_CPU_THREADS", "0"))
    return min(n, cap) if cap > 0 else n


def cpu_baseline(args, scene_buffers, prim_count, tri_count, reference_buffers=None):
    """Time the CPU oracle on all host cores on a bounded sample (same scene, size, depth, RNG; fewer spp).
    Also returns the per-ray work counters that price the algorithmic bytes per ray (reference traversal)."""
    from oracle import binding as ob
    cores = host_cores()
    u = ob.make_uniforms(args.width, args.height, prim_count, tri_count)
    # sized for roughly 10-30 core-seconds: ~3.4 Mrays/s/core, 1.68 rays/path
    spp = max(1, args.cpu_spp)
    t0 = time.perf_counter()
    _, ct = ob.render(u, scene_buffers, rng_mode=ob.RNG_PHILOX, max_depth=args.depth, accumulate=1, sample_count=spp,
                      seed=(1, 0), threads=cores)
    dt = time.perf_counter() - t0
    n_node = ct["node_pops"] / ct["rays"]
    n_prim = ct["prim_tests"] / ct["rays"]
    h = ct["bounces"] / ct["rays"]
    # one core as well: 1/16 of the sample, at least 1 spp
    spp1 = max(1, spp // 16)
    t0 = time.perf_counter()
    _, c1 = ob.render(u, scene_buffers, rng_mode=ob.RNG_PHILOX, max_depth=args.depth, accumulate=1, sample_count=spp1,
                      seed=(1, 0), threads=1)
    dt1 = time.perf_counter() - t0
    out = dict(value=ct["rays"] / dt / 1e6, unit="Mrays/s", cores=cores, kind="port",
               tree={"reference": "the reference's own tree (Scene::buildBVH sweep SAH, R/Scene/Scene.h:195-317)",
                     "binned": "the host binned-SAH tree", "gpu": "the GPU-built tree (mpt_build_bvh)",
                     "device": "the device-built tree the GPU leg renders (mpt_build_and_upload, read back by mpt_download_bvh) — same arrays, "
                               "same walk order on both sides"}[args.bvh],
               sample="%dx%d, %d spp, depth %d, philox seed (1,0): %d rays in %.2f s on %d threads"
                      % (args.width, args.height, spp, args.depth, ct["rays"], dt, cores),
               single_core={"value": c1["rays"] / dt1 / 1e6, "unit": "Mrays/s",
                            "sample": "%d spp: %d rays in %.2f s on 1 thread" % (spp1, c1["rays"], dt1)})
    if reference_buffers is not None:   # the reference's path on the reference's tree: what "the reference's CPU path" would walk
        t0 = time.perf_counter()
        _, cr = ob.render(u, reference_buffers, rng_mode=ob.RNG_PHILOX, max_depth=args.depth, accumulate=1, sample_count=spp,
                          seed=(1, 0), threads=cores)
        dtr = time.perf_counter() - t0
        out["reference_tree"] = {"value": cr["rays"] / dtr / 1e6, "unit": "Mrays/s", "cores": cores,
                                 "sample": "%d spp: %d rays in %.2f s on %d threads, %.2f node pops and %.2f primitive tests per ray"
                                           % (spp, cr["rays"], dtr, cores, cr["node_pops"] / cr["rays"], cr["prim_tests"] / cr["rays"])}
    return out, (n_node, n_prim, h)


CORNELL_CAM = dict(pos=(0.0, 1.0, 3.4), fwd=(0.0, 0.0, -1.0), up=(0.0, 1.0, 0.0), vfov=40.0)
BVH_CODE = {"reference": 0, "binned": 1, "gpu": 2, "device": 3}


def roofline_from_profile(kernel, workload, rays, seconds):
    """The fractions of one workload: per-ray counters of the committed profile of THIS workload and THIS build (counter_profile)
    x the rays traced in `seconds` of wall time.  Returns (dict, why): dict is None when no such profile is committed."""
    prof, why = counter_profile(kernel, workload)
    if not prof:
        return None, why
    valu_rate = prof["valu_per_ray"] * rays / seconds / 1e9                  # G wave64 instructions per second
    valu_peak = N_SIMD * MAX_CLOCK_GHZ / 2.0                                 # 1024 SIMD-32s, one wave64 instruction per 2 cycles, 2.4 GHz
    hbm_rate = prof["hbm_bytes_per_ray"] * rays / seconds / 1e9
    salu_rate = prof["salu_per_ray"] * rays / seconds / 1e9
    return {"prof": prof, "valu_rate": valu_rate, "valu_peak": valu_peak, "valu_frac": valu_rate / valu_peak,
            "hbm_rate": hbm_rate, "hbm_frac": hbm_rate / HBM_PEAK_GBS, "salu_rate": salu_rate, "salu_frac": salu_rate / (N_CU * MAX_CLOCK_GHZ)}, {}


def extra_workloads(ctx, capi, host, depth):
    """Untimed extras (N = 1), like serial_ms_per_render: ONE serial render each of
      * north_star's "synthetic Cornell-style scene" and bunny x20 (BASELINE.json configs[2]'s scene) at 1920x1080 x 256 spp on the
        tree the Renderer's throughput mode builds for them (device build),
      * scene.xml on the REFERENCE's own tree (Scene::buildBVH sweep SAH + mpt_upload_scene: the drop-in route) at the headline size,
      * configs[4] at its real size: the 1,000,003-primitive scene, 1920x1080 x 4096 spp, depth 16, Scatter.h BSDFs, one GPU's 1/8
        tile shard (1.06 G paths),
    so that the driver's record carries them.  HIP-event time of the whole render, best of two after a warm-up; each with the
    fraction of the vector-issue peak from its OWN committed counter profile (or profile_missing / profile_stale)."""
    import tempfile
    import time
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import config4_scene
    out = []
    tmp = tempfile.mkdtemp(prefix="mpt_cfg4_")
    cases = (("cornell.xml", os.path.join(ROOT, "assets", "cornell.xml"), "device", CORNELL_CAM, 256, depth, capi.BSDF_LAMBERT, 1),
             ("bunny20.xml", os.path.join(ROOT, "assets", "bunny20.xml"), "device", None, 256, depth, capi.BSDF_LAMBERT, 1),
             ("scene.xml", os.path.join(ROOT, "assets", "scene.xml"), "reference", None, 256, depth, capi.BSDF_LAMBERT, 1),
             ("config4", None, "device", None, 4096, 16, capi.BSDF_SCATTER, 8))
    for name, xml, builder, cam, spp, dep, bsdf, shards in cases:
        if xml is None:
            xml = config4_scene.write(tmp)
        sc = host.Scene()
        st, log = host.SceneLoader.LoadSceneFromXML(xml, sc)
        if st != 0:
            out.append({"workload": name, "error": log[-200:]})
            continue
        t0 = time.perf_counter()
        if builder == "device":                      # what the Renderer does in its throughput mode: build -> render on the device
            sc.sortPrimitives()
            prims, mats = sc.packed_primitives()
            t0 = time.perf_counter()
            ctx.build_and_upload(prims, mats)
        else:
            sc.buildBVH(host.BVH_REFERENCE_SWEEP)
            ctx.upload_scene(*sc.buffers())
        setup_ms = (time.perf_counter() - t0) * 1e3
        W, H = 1920, 1080
        ctx.resize(W, H)
        ctx.set_uniforms(host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount(), cam=cam))
        pipe = ctx.accel_info()["auto_pipeline"]
        best = None
        for k in range(3):
            ctx.reset_stats()
            ctx.render(rng_mode=capi.RNG_PHILOX, bsdf_mode=bsdf, max_depth=dep, pipeline=capi.PIPE_AUTO, seed=(1, 0),
                       sample_begin=k * spp, sample_count=spp, shard_rank=0, shard_count=shards)
            s = ctx.stats()
            if k and (best is None or s["total_ms"] < best["total_ms"]):
                best = s
        e = {"workload": "%s %dx%d x %d spp, depth %d, %sone serial render" % (name, W, H, spp, dep, "1/%d tile shard, " % shards if shards > 1 else ""),
             "prims": sc.getPrimitiveCount(), "bvh_builder": builder, "build_and_upload_ms": setup_ms,
             "pipeline": PIPE_NAMES[pipe], "ms_per_render": best["total_ms"],
             "mrays_per_s": best["rays"] / best["total_ms"] / 1e3, "rays": best["rays"], "paths": best["paths"]}
        workload = {"scene": name if name != "config4" else "config4", "width": W, "height": H, "spp": spp, "depth": dep,
                    "bvh": BVH_CODE[builder], "env": capi.knob_env(), "shards": shards, "bsdf": int(bsdf)}
        r, why = roofline_from_profile(PIPE_KERNEL[pipe], workload, best["rays"], best["total_ms"] * 1e-3)
        if r:
            e["roofline"] = {"kernel": PIPE_KERNEL[pipe], "bound": "valu" if r["valu_frac"] >= r["hbm_frac"] else "hbm",
                             "frac": max(r["valu_frac"], r["hbm_frac"]), "valu_frac": r["valu_frac"], "hbm_frac": r["hbm_frac"],
                             "valu_instr_per_ray": r["prof"]["valu_per_ray"], "hbm_bytes_per_ray": r["prof"]["hbm_bytes_per_ray"],
                             "wave_cycles_split": r["prof"]["wave_cycles_split"], "profile": r["prof"]["file"]}
        else:
            e["roofline"] = dict(kernel=PIPE_KERNEL[pipe], frac=None, **why)
        out.append(e)
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
        args.gpus = world

    import numpy as np
    import torch
    from metalpathtracer_amd import capi, distributed as D, host

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the HIP path is the product and there is no CPU fallback")
    backend = os.environ.get("MPT_BENCH_BACKEND", "nccl")   # "gloo" only for functional tests on a 1-GPU box
    if os.environ.get("MPT_BENCH_SHARE_GPU"):               # functional tests: several ranks on one GPU
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    if world > 1:
        if backend == "nccl":
            D.init_from_env(backend="nccl")
        else:
            import torch.distributed as dist0
            dist0.init_process_group(backend=backend, rank=rank, world_size=world)
        import torch.distributed as dist

    # ---- untimed setup: ingest, BVH, upload (the product's own host layer) ----
    sc = host.Scene()
    st, log = host.SceneLoader.LoadSceneFromXML(args.scene, sc)
    if st != 0:
        sys.exit("cannot load %s: %s" % (args.scene, log))
    ctx = capi.Context(local)
    # (device: build -> render on the device, mpt_build_and_upload — what the Renderer's throughput mode does; the tree comes
    #  back in the reference's format for the CPU leg, which walks the same arrays)
    buffers = host.make_ready(ctx, sc, {"reference": host.BVH_REFERENCE_SWEEP, "binned": host.BVH_BINNED_CENTROID, "gpu": host.BVH_GPU_LBVH,
                                        "device": host.BVH_DEVICE}[args.bvh])
    P, T = sc.getPrimitiveCount(), sc.getTriangleCount()
    W, H = args.width, args.height
    ctx.resize(W, H)
    ctx.set_uniforms(host.make_uniforms(W, H, P, T))
    fb = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    ctx.set_sum_buffer(fb.data_ptr())
    pipe = {"default": capi.DEFAULT_PIPELINE, "wavefront": capi.PIPE_WAVEFRONT, "megakernel": capi.PIPE_MEGAKERNEL,
            "wavelocal": capi.PIPE_WAVELOCAL, "ordered": capi.PIPE_ORDERED}[args.pipeline]
    if pipe == capi.PIPE_AUTO:   # what mpt_render resolves AUTO to (include/mpt.h), so that the line names the kernel that ran
        pipe = ctx.accel_info()["auto_pipeline"]
    kw = dict(rng_mode=capi.RNG_PHILOX, bsdf_mode=capi.BSDF_LAMBERT, max_depth=args.depth, pipeline=pipe, seed=(1, 0),
              shard_rank=rank, shard_count=world, slots_per_iter=args.slots)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    spp = args.spp_per_step
    # setup, untimed: both render lanes allocate their workspace (per-path result slots, ray rings) on first use
    for lane in range(2):
        ctx.render_async(sample_begin=0, sample_count=spp, **kw)
    ctx.wait()
    for w in range(args.warmup):
        ctx.render(sample_begin=w * spp, sample_count=spp, **kw)
    if world > 1:  # warm the collective too
        D.reduce_framebuffer(fb.clone())
    fb.zero_()
    ctx.reset_stats()
    kernel_ms = 0.0
    launches = 0
    barrier()
    t0 = time.perf_counter()
    # K steps = K renders of spp samples each, accumulated into one HDR sum.  They are enqueued without a host wait in
    # between (mpt_render_async): the next step's trace kernel fills the compute units that the previous step's tail and
    # resolve leave idle; the sum updates stay in step order, so the image is bit-identical to serial steps.
    for k in range(args.steps):
        ctx.render_async(sample_begin=k * spp, sample_count=spp, **kw)
    ctx.wait()
    s = ctx.stats()
    kernel_ms = s["trace_kernel_ms"]           # HIP events on the lanes' own streams, around every dominant-kernel launch
    launches = s["trace_launches"]
    if world > 1:
        D.reduce_framebuffer(fb)               # ONE reduce(sum) of the HDR framebuffer to rank 0 (RCCL)
    barrier()
    elapsed = time.perf_counter() - t0
    st = ctx.stats()
    rays_local = st["rays"]
    image_mean = (fb[..., :3].double().mean(dim=(0, 1)) / (args.steps * spp)).tolist() if rank == 0 else None
    # untimed extra (N = 1): ONE render at a time, no overlap of consecutive steps — what a single 256-spp frame takes
    serial_ms = []
    if world == 1:
        for k in range(min(3, args.steps)):
            ctx.render(sample_begin=(args.steps + k) * spp, sample_count=spp, **kw)
            serial_ms.append(ctx.stats()["total_ms"])
    serial_ms = sum(serial_ms) / len(serial_ms) if serial_ms else None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        r = torch.tensor([rays_local, st["paths"]], dtype=torch.float64, device="cuda")
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        rays_total, paths_total = int(r[0].item()), int(r[1].item())
    else:
        rays_total, paths_total = rays_local, st["paths"]

    if rank == 0:
        mean = image_mean
        out = {
            "metric": "Mrays/s at 1920x1080x256spp on scene.xml",
            "value": rays_total / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "assets/scene.xml (the reference's bundled scene, mesh path remapped) — no dataset involved",
            "config": {
                "workload": "scene.xml %dx%d x %d spp per step, depth %d (%d steps timed)" % (W, H, spp, args.depth,
                                                                                           args.steps),
                "prims": P, "bvh_nodes": len(buffers[0]), "bvh_builder": args.bvh, "rng": "philox4x32-10 (pixel,sample,bounce)",
                "pipeline": PIPE_NAMES[pipe],
                "parallelism": "8x8-tile interleave over %d rank(s)%s" % (world, " + 1 RCCL reduce(sum) of the HDR framebuffer" if world > 1 else ""),
                "paths": paths_total, "rays": rays_total, "rays_per_path": rays_total / max(1, paths_total),
                "mpaths_per_s": paths_total / elapsed / 1e6,
                "image_mean_rgb": mean,
            },
        }
        n_node, n_prim, h = 7.52, 3.59, 0.405   # SURVEY.md App. C.5 (used only if the CPU leg is skipped)
        if world == 1 and not args.no_cpu_baseline and args.cpu_spp > 0:
            ref_buffers = None
            if args.bvh != "reference":          # a second CPU figure on the reference's own tree (what the reference itself would walk)
                sc_ref = host.Scene()
                st_ref, _ = host.SceneLoader.LoadSceneFromXML(args.scene, sc_ref)
                if st_ref == 0:
                    sc_ref.buildBVH(host.BVH_REFERENCE_SWEEP)
                    ref_buffers = sc_ref.buffers()
            out["cpu_baseline"], (n_node, n_prim, h) = cpu_baseline(args, buffers, P, T, ref_buffers)
        b_ray = B_QUEUE + 32.0 * n_node + 52.0 * n_prim + 32.0 * h
        out["serial_ms_per_render"] = serial_ms
        if world == 1 and not args.no_extra_workloads:
            out["extra_workloads"] = extra_workloads(ctx, capi, host, args.depth)
        out["config"]["serial_mrays_per_s"] = rays_local / args.steps / serial_ms / 1e3 if serial_ms else None
        if launches and kernel_ms > 0:
            rays_per_launch = rays_local / launches
            sec_per_launch = kernel_ms * 1e-3 / launches
            workload = {"scene": os.path.basename(args.scene), "width": W, "height": H, "spp": spp, "depth": args.depth,
                        "bvh": BVH_CODE[args.bvh], "env": capi.knob_env(), "launch": "async"}   # (the timed steps are mpt_render_async)
            rf = {"kernel": PIPE_KERNEL[pipe], "launches": launches, "avg_launch_ms": kernel_ms / launches,
                  "rays_per_launch": rays_per_launch}
            # THE fraction: this rank's rays of the timed region over the WALL time of the timed region (the same clock `value` is
            # quoted on) — what the chip delivered.  `avg_launch_ms` is what the persistent kernel stamped itself (first workgroup's
            # start to last wave's end on the chip's 100 MHz clock: the span rocprofv3's kernel trace reports); the end of one launch
            # overlaps the start of the next, so the sum of the launches is a little more than the region.  Per launch: `per_launch`.
            r, why = roofline_from_profile(PIPE_KERNEL[pipe], workload, rays_local, elapsed)
            if r:
                prof = r["prof"]
                if r["valu_frac"] >= r["hbm_frac"]:
                    rf.update(bound="valu", achieved=r["valu_rate"], peak=r["valu_peak"], unit="G wave64-instr/s", frac=r["valu_frac"])
                else:
                    rf.update(bound="hbm", achieved=r["hbm_rate"], peak=HBM_PEAK_GBS, unit="GB/s", frac=r["hbm_frac"])
                rf["basis"] = ("per-ray counters of the committed profile x the %d rays of the timed region / its %.4f s of wall time (n_gpus = 1: "
                               "the chip's rate; n_gpus > 1: rank 0's share over the job's wall time)" % (rays_local, elapsed))
                rf["traffic"] = prof["hbm_bytes_per_ray"] * rays_per_launch          # HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE)
                rf["valu"] = {"achieved": r["valu_rate"], "peak": r["valu_peak"], "unit": "G wave64-instr/s", "frac": r["valu_frac"],
                              "measured_clock_ghz": prof["clock_ghz"],
                              "measured_clock_note": "GRBM_GUI_ACTIVE / 8 XCDs / kernel time of the PROFILED launch (profiled passes "
                                                     "clock lower than un-profiled ones)",
                              "frac_at_measured_clock": r["valu_rate"] / (N_SIMD * prof["clock_ghz"] / 2.0),
                              "instr_per_ray": prof["valu_per_ray"], "lane_utilisation": prof["lane_utilisation"],
                              "lane_utilisation_note": "exec-mask utilisation (SQ_THREAD_CYCLES_VALU / 64 SQ_INSTS_VALU)",
                              "salu_per_valu": prof["salu_per_ray"] / prof["valu_per_ray"]}
                # the scalar unit: one per CU, one instruction per cycle, shared by the CU's four SIMDs
                rf["scalar"] = {"achieved": r["salu_rate"], "peak": N_CU * MAX_CLOCK_GHZ, "unit": "G instr/s", "frac": r["salu_frac"],
                                "instr_per_ray": prof["salu_per_ray"]}
                rf["hbm"] = {"achieved": r["hbm_rate"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": r["hbm_frac"],
                             "bytes_per_ray": prof["hbm_bytes_per_ray"], "l2_hit_rate": prof["l2_hit_rate"]}
                pl, _ = roofline_from_profile(PIPE_KERNEL[pipe], workload, rays_per_launch, sec_per_launch)
                rf["per_launch"] = {"avg_launch_ms": kernel_ms / launches, "valu_frac": pl["valu_frac"], "hbm_frac": pl["hbm_frac"],
                                    "hbm_gbs": pl["hbm_rate"],
                                    "note": "in-kernel span of each launch (first workgroup's start to last wave's end, 100 MHz clock); the end of "
                                            "one launch overlaps the start of the next and the resolve of the render before runs beside it, so "
                                            "this is not the chip's rate (the timed-region figures above are)"}
                if serial_ms:
                    sr, _ = roofline_from_profile(PIPE_KERNEL[pipe], workload, rays_local / args.steps, serial_ms * 1e-3)
                    rf["serial_render"] = {"ms": serial_ms, "valu_frac": sr["valu_frac"], "hbm_frac": sr["hbm_frac"],
                                           "note": "one render at a time (no overlap of consecutive steps), whole mpt_render incl. resolve"}
                rf["wave_cycles_split"] = prof["wave_cycles_split"]
                rf["profile"] = {"file": prof["file"], "build": prof["build"], "workload": workload}
                rf["source"] = ("%s (rocprofv3 --pmc, one counter set per pass, same build / scene / size / spp / knobs — checked by "
                                "sha256; the profiled launch took %.2f ms)" % (prof["file"], prof["profiled_kernel_ms"]))
            else:
                rf.update(bound="valu", achieved=None, peak=N_SIMD * MAX_CLOCK_GHZ / 2.0, unit="G wave64-instr/s", frac=None, traffic=None,
                          source="no usable counter profile: %s" % json.dumps(why), **why)
            rf["algorithmic"] = {
                "bytes_per_ray": b_ray, "n_node": n_node, "n_prim": n_prim, "h": h,
                "bytes_per_launch": b_ray * rays_per_launch,
                "gbs_if_all_of_it_were_hbm": b_ray * rays_per_launch / sec_per_launch / 1e9,
                "note": "SURVEY 8(d): 168 + 32 n_node + 52 n_prim + 32 h with the reference walk's counts from the CPU oracle "
                        "in this run; BVH reads are served from LDS and primary rays never leave registers, so this figure is "
                        "not HBM traffic and is not used as a bound"}
            out["roofline"] = rf
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
