"""bench.py end to end on the GPU box: the contract line (metric / roofline / cpu_baseline objects) at N = 1 and the
N = 2 launch path (`python -m torch.distributed.run`, one rank per process) on the single GPU of the test box.  The
2-rank run uses gloo for the reduce (two RCCL ranks cannot share one device) and both ranks render on GPU 0; it must
produce the same image mean as the 1-rank run because the tile shards partition the pixels."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
SMALL = ["--steps", "2", "--warmup", "1", "--spp-per-step", "8", "--width", "640", "--height", "360"]


def last_json(text):
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


def test_bench_line_single_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *SMALL, "--cpu-spp", "2"], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["unit"] == "Mrays/s" and d["value"] > 0
    rf = d["roofline"]
    # a bound that binds: vector-ALU issue at the chip's 2.4 GHz or measured HBM traffic, whichever is larger — never above 1
    assert rf["bound"] in ("valu", "hbm") and rf["launches"] == 2 and rf["kernel"] == "k_wavelocal"   # AUTO: < 8192 primitives
    if rf["frac"] is not None:      # a counter profile of THIS workload and THIS build is committed under profiles/
        assert 0.0 < rf["frac"] <= 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
        assert rf["frac"] == max(rf["valu"]["frac"], rf["hbm"]["frac"])
        assert rf["valu"]["peak"] == 1024 * 2.4 / 2 and rf["valu"]["frac_at_measured_clock"] >= rf["valu"]["frac"]
        assert rf["hbm"]["peak"] == 8000.0 and 0.0 < rf["valu"]["lane_utilisation"] <= 1.0
        assert rf["traffic"] > 0 and "profiles/" in rf["source"]
        # THE fraction is achieved-over-the-timed-region: per-ray counters x the rays of the K timed steps / their wall time —
        # the same clock `value` is quoted on — not the (overlapped, longer) HIP-event duration of a single launch
        wall_s = d["ms_per_step"] * d["steps"] * 1e-3
        assert abs(rf["valu"]["frac"] - rf["valu"]["instr_per_ray"] * d["config"]["rays"] / wall_s / 1e9 / rf["valu"]["peak"]) < 1e-6 * rf["valu"]["frac"] + 1e-12
        assert abs(rf["hbm"]["achieved"] - rf["hbm"]["bytes_per_ray"] * d["config"]["rays"] / wall_s / 1e9) < 1e-6 * rf["hbm"]["achieved"] + 1e-9
        assert rf["per_launch"]["avg_launch_ms"] == rf["avg_launch_ms"] and 0.0 < rf["per_launch"]["valu_frac"] <= 1.0
        assert 0.0 < rf["serial_render"]["valu_frac"] <= 1.0
        from metalpathtracer_amd import capi
        bid, pb = capi.build_id(), rf["profile"]["build"]
        assert pb["source_sha256"] == bid["source_sha256"] or pb["lib_sha256"] == bid["lib_sha256"]
    else:                           # ... otherwise the line says why, and prints no fraction it cannot stand behind
        assert rf.get("profile_stale") is True or "profile_missing" in rf
        assert rf["achieved"] is None and rf["traffic"] is None
    assert rf["algorithmic"]["bytes_per_ray"] > 168.0
    assert d["serial_ms_per_render"] > 0 and d["config"]["serial_mrays_per_s"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "spp" in cb["sample"]
    assert "device-built tree" in cb["tree"] and cb["reference_tree"]["value"] > 0      # which tree the CPU leg walked, and the reference's own
    assert d["config"]["paths"] == 640 * 360 * 8 * 2
    # untimed extras, one serial render each: the Cornell-style scene and bunny x20 (device-built trees), scene.xml on the REFERENCE's
    # own tree, and configs[4] at its real size (1/8 tile shard, 4096 spp, depth 16, Scatter.h BSDFs)
    ex = d["extra_workloads"]
    assert [e["workload"].split()[0] for e in ex] == ["cornell.xml", "bunny20.xml", "scene.xml", "config4"]
    assert all(e["ms_per_render"] > 0 and e["mrays_per_s"] > 0 for e in ex)
    assert all(e["paths"] == 1920 * 1080 * 256 for e in ex[:3]) and ex[3]["paths"] == 4050 * 64 * 4096 and ex[3]["prims"] == 1000003
    assert "closest-first" in ex[1]["pipeline"] and "reference-order" in ex[0]["pipeline"] and "closest-first" in ex[3]["pipeline"]
    assert ex[2]["bvh_builder"] == "reference" and "reference-order" in ex[2]["pipeline"]
    for e in ex:                    # each extra names its own counter profile, or says that none of this build is committed
        r = e["roofline"]
        assert (r["frac"] is not None and 0.0 < r["frac"] <= 1.0 and "profiles/" in r["profile"]) or r.get("profile_stale") is True or "profile_missing" in r


def test_bench_two_ranks_share_the_gpu_and_agree_with_one_rank():
    env = dict(os.environ, MPT_BENCH_BACKEND="gloo", MPT_BENCH_SHARE_GPU="1")
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *SMALL, "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29531", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", *SMALL, "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env)
    assert two.returncode == 0, two.stderr[-3000:]
    a, b = last_json(one.stdout), last_json(two.stdout)
    assert b["n_gpus"] == 2 and "reduce" in b["config"]["parallelism"]
    assert a["config"]["paths"] == b["config"]["paths"] and a["config"]["rays"] == b["config"]["rays"]
    assert a["config"]["image_mean_rgb"] == b["config"]["image_mean_rgb"]      # every pixel has exactly one owner
