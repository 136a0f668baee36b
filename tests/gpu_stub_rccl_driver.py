"""Driver of tests/test_gpu_stub_rccl.py: runs in a process of its own, with tests/stub_rccl/_build first on LD_LIBRARY_PATH, so that the
product's dlopen("librccl.so.1") binds the TEST DOUBLE (tests/stub_rccl/stub_rccl.hip) — torch, which brings the real librccl, is never
imported here.  Executes the N > 1 branch of the collective path (mpt_hip.hip: mpt_comm_create_all / mpt_comm_create_rank /
mpt_reduce_sum / comm_abort) on ONE GPU.  Orchestration evidence, not a scaling measurement.  Prints one line per check, "ALL OK" last."""
import ctypes as C
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from metalpathtracer_amd import capi, host

assert "torch" not in sys.modules
stub = C.CDLL("librccl.so.1")                       # the same object the product's dlopen returns
assert hasattr(stub, "stub_rccl_counters"), "the real librccl was found first: LD_LIBRARY_PATH does not start with the stub's directory"


def counters():
    out = (C.c_ulonglong * 16)()
    stub.stub_rccl_counters(out)
    names = ("GetUniqueId", "CommInitAll", "CommInitRank", "CommDestroy", "CommAbort", "Reduce", "GroupStart", "GroupEnd", "completed", "failed", "floats_added",
             "ungrouped_reduce")
    return dict(zip(names, [int(v) for v in out]))


def same(a, b):
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


W, H, SPP = 320, 180, 4
sc = host.Scene(); st, _ = host.SceneLoader.LoadSceneFromXML(os.path.join(ROOT, "assets", "scene.xml"), sc); assert st == 0
sc.buildBVH()
buf = sc.buffers()
u = host.make_uniforms(W, H, sc.getPrimitiveCount(), sc.getTriangleCount())
kw = dict(rng_mode=capi.RNG_PHILOX, max_depth=8, seed=(3, 1), sample_count=SPP)


def ready(w=W, h=H):
    c = capi.Context(0)                              # every rank on GPU 0: what real RCCL refuses and the test double allows
    c.upload_scene(*buf); c.resize(w, h); c.set_uniforms(host.make_uniforms(w, h, sc.getPrimitiveCount(), sc.getTriangleCount())); c.clear_sum()
    return c


whole_ctx = ready()
whole_ctx.render(**kw)
whole = whole_ctx.read_sum()
rays_whole = whole_ctx.stats()["rays"]

# ---- 1. one host thread, N contexts: mpt_comm_create_all (ncclCommInitAll) + one grouped reduce ----------------------------------
for n in (2, 3):
    ctxs = [ready() for _ in range(n)]
    before = counters()
    comm = capi.Comm.all(ctxs)
    for r, c in enumerate(ctxs):
        c.render_async(shard_rank=r, shard_count=n, **kw)      # in flight when the reduce is called: mpt_reduce_sum collects them first
    comm.reduce_sum(0)
    d = {k: counters()[k] - before[k] for k in before}
    assert same(ctxs[0].read_sum(), whole), "rank 0 does not hold the single-GPU image after the reduce (n = %d)" % n
    assert sum(c.stats()["rays"] for c in ctxs) == rays_whole
    assert (d["CommInitAll"], d["GroupStart"], d["GroupEnd"], d["Reduce"], d["completed"], d["failed"], d["ungrouped_reduce"]) == (1, 1, 1, n, 1, 0, 0), d
    assert d["floats_added"] == (n - 1) * W * H * 4
    for r in range(1, n):                                        # the other ranks keep their own shard (a reduce, not an all-reduce)
        own = ctxs[r].read_sum()
        assert not same(own, whole) and float(own[..., 3].max()) > 0
    comm.reduce_sum(0)                                           # a second collective on the same communicator is fine (sum = whole + shards again)
    comm.close()
    assert counters()["CommDestroy"] - before["CommDestroy"] == n
    for c in ctxs:
        c.close()
    print("create_all n=%d: root holds the single-GPU image, every float; calls %s" % (n, d))

# ---- 2. one rank per thread: mpt_comm_unique_id + mpt_comm_create_rank (ncclCommInitRank), ungrouped-by-rank reduces that meet in the clique
ctxs = [ready(), ready()]
uid = capi.Comm.unique_id()
comms = [capi.Comm.rank(ctxs[r], r, 2, uid) for r in range(2)]
before = counters()
errs = [None, None]


def rank_main(r):
    try:
        ctxs[r].render(shard_rank=r, shard_count=2, **kw)
        comms[r].reduce_sum(1)                                   # root = rank 1 this time
    except Exception as e:                                       # noqa: BLE001 (reported below)
        errs[r] = e


ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
[t.start() for t in ts]; [t.join() for t in ts]
assert errs == [None, None], errs
d = {k: counters()[k] - before[k] for k in before}
assert same(ctxs[1].read_sum(), whole) and not same(ctxs[0].read_sum(), whole)
assert (d["GroupStart"], d["GroupEnd"], d["Reduce"], d["completed"], d["failed"]) == (2, 2, 2, 1, 0), d
print("create_rank x 2 threads, root 1: root holds the single-GPU image; calls %s" % d)

# ---- 3. failure legs ----------------------------------------------------------------------------------------------------------------
# (a) a rank that cannot enter the collective aborts its communicator; its peer's ncclReduce FAILS instead of hanging, and aborts too
before = counters()
bad = capi.Context(0)                                            # never sized: no HDR sum to reduce
uid2 = capi.Comm.unique_id()
cb = capi.Comm.rank(bad, 0, 2, uid2)
cg = capi.Comm.rank(ctxs[0], 1, 2, uid2)
res = {}


def peer():
    try:
        cg.reduce_sum(0)
        res["peer"] = "ok"
    except capi.MptError as e:
        res["peer"] = (e.status, str(e))


t = threading.Thread(target=peer); t.start()
try:
    cb.reduce_sum(0)
    res["bad"] = "ok"
except capi.MptError as e:
    res["bad"] = (e.status, str(e))
t.join(30)
assert not t.is_alive(), "the peer of an aborted rank hangs in the collective"
assert res["bad"][0] == 5 and "sized alike" in res["bad"][1], res            # MPT_ERR_NOT_READY
assert res["peer"][0] == 3 and "ncclReduce" in res["peer"][1] and "aborted" in res["peer"][1], res   # MPT_ERR_HIP: the peer aborted
for c in (cb, cg):                                               # both communicators are dead now: the next reduce is refused, nothing is called
    n_before = counters()["Reduce"]
    try:
        c.reduce_sum(0)
        raise AssertionError("a reduce on an aborted communicator went through")
    except capi.MptError as e:
        assert e.status == 5 and "aborted" in str(e), str(e)
    assert counters()["Reduce"] == n_before
d = {k: counters()[k] - before[k] for k in before}
assert d["CommAbort"] == 2 and d["completed"] == 0 and d["failed"] >= 1, d
cb.close(); cg.close(); bad.close()
print("abort leg (create_rank): the failing rank answers NOT_READY, its peer's reduce fails, both communicators refuse further work; calls %s" % d)

# (b) create_all with contexts of different sizes: refused before anything is posted, communicators aborted, next reduce refused
odd = [ready(), ready(W // 2, H // 2)]
before = counters()
comm = capi.Comm.all(odd)
for attempt in range(2):
    try:
        comm.reduce_sum(0)
        raise AssertionError("contexts of different sizes were reduced")
    except capi.MptError as e:
        assert e.status == 5 and ("sized alike" in str(e) if attempt == 0 else "aborted" in str(e)), str(e)
d = {k: counters()[k] - before[k] for k in before}
assert (d["CommAbort"], d["Reduce"], d["GroupStart"]) == (2, 0, 0), d
comm.close()
for c in odd + ctxs + [whole_ctx]:
    c.close()
print("abort leg (create_all, sizes differ): NOT_READY, two ncclCommAbort, no ncclReduce posted, the next reduce refused; calls %s" % d)

# ---- 4. the CLI: mpt_render --devices 0,0 renders two shards on GPU 0 and reduces them; the image file equals the one-GPU file -------
if len(sys.argv) > 1:
    import subprocess
    tmp = sys.argv[1]
    exe = os.path.join(ROOT, "metalpathtracer_amd", "lib", "mpt_render")
    base = [exe, "--scene", os.path.join(ROOT, "assets", "scene.xml"), "--width", "192", "--height", "108", "--spp", "4", "--depth", "8", "--seed", "5"]
    a, b = os.path.join(tmp, "one.pfm"), os.path.join(tmp, "two.pfm")
    for extra, out in (([], a), (["--devices", "0,0"], b)):
        r = subprocess.run(base + extra + ["--out", out], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    assert '"gpus": 2' in r.stdout, r.stdout
    assert open(a, "rb").read() == open(b, "rb").read()
    print("mpt_render --devices 0,0 == mpt_render on one GPU, byte for byte")
print("ALL OK")
