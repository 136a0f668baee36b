// mpt_hip.hip — host side of the C ABI (include/mpt.h) of the MI355X path-tracing hot path: context and resource
// management, conversion of the reference's flat scene arrays into the device layout (threaded breadth-first BVH,
// leaf-ordered primitives, de-duplicated materials), pass scheduling and kernel launches.  The kernels themselves are
// in mpt_kernels.h, their building blocks in mpt_device.h.
//
// One mpt_render = one pass per <= 2^30 paths: k_begin_pass, ONE k_wavelocal launch (default pipeline; the global
// wavefront launches k_step + k_advance per bounce generation instead), k_resolve_sum, one host synchronisation.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see Makefile).  gfx950 only.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <map>
#include <memory>
#include <exception>
#include <string>
#include <vector>

#include "mpt.h"
#include "mpt_accel.h"
#include "mpt_device.h"

#include "mpt_kernels.h"
#include "mpt_lbvh.h"
#include "mpt_devbuild.h"
#include "mpt_ordered.h"

// =====================================================================================================
// host side of the C ABI
// =====================================================================================================
// Everything one in-flight render owns.  A context has two lanes so that consecutive renders overlap: the next
// render's trace kernel fills the CUs that the previous one's tail (and its resolve) leave idle (measured with two
// contexts: +11 % at 1080p x 256 spp, +15 % for a 1/8 tile shard).  Resolves are chained in submission order, so the
// HDR sum is bit-identical to the serial schedule.
// Box-test trips granted per step of ring 0..LEVELS-2 (the last ring runs to completion).  Rounds 1-3 (five rings), scene.xml at
// 1080p x 256 spp: 8/20/50/125 -> 29.3 ms, 8/24/72/inf -> 30.4, 8/16/32/64 -> 30.2, 8/16/32/inf -> 31.8, 8/12/24/48 -> 31.3.
// Round 4 (two rings, mpt_kernels.h): only ring 0's budget is left — 6 / 8 / 12 / 16 / 24: 17.01 / 16.88 / 17.11 / 17.49 / 18.26 ms.
static WaveBudgets default_budgets() {
    WaveBudgets w;
    const uint32_t ladder[] = {8, 20, 50, 125, 300, 700, 1600};
    for (uint32_t k = 0; k < MPT_WL_LEVELS; ++k) {
        w.b[k] = k + 1 < MPT_WL_LEVELS && k < 7 ? ladder[k] : MPT_WL_NO_BUDGET;
        // a step of ring 1.. also ends once fewer than 24 of its 64 lanes are still traversing (the stragglers are
        // parked and regrouped): 29.9 -> 28.5 ms; 16 or 32 lanes are about as good, 8 is not, ring 0 is better without
        w.min_active[k] = k == 0 ? 0 : 24;
    }
    return w;
}

// Node-loop trips a step of tree-walk ring M_k may make before its unfinished rays are parked in M_k+1 (MPT_OT_BUDGETS),
// and the number of lanes that must still be walking for a step to go on (MPT_OT_MIN_ACTIVE).
static OtBudgets default_ot_budgets() {
    OtBudgets b;
    // Round 5: a walk step of ring M0 (and a walk made in place by a top-test step) ends when fewer than 24 of its lanes still walk,
    // or after 32 trips of the node loop; a step of the last ring when fewer than 32 do.  Until round 4 the first rule did not exist
    // (16 trips and out, whoever was still walking): a trip budget cuts a step short while most of its lanes are still busy and sends
    // them through a ring for nothing, a utilisation rule ends it when it has become wasteful.  Same-box sweep (gpurun_out/r05/s10, s11:
    // budgets 16 .. 1000 x minimum lanes 0 .. 40 x 24 / 32), (16; 0, 24) -> (32; 24, 32): 8 / 20 bunnies 43.6 -> 42.8 / 55.1 -> 53.9 ms,
    // the 1 M-triangle shard of configs[4] 117.2 -> 113.2; no trip budget at all (1000; 24, 32) is the same within 0.1 %; a budget
    // WITHOUT the rule is far worse than before (32; 0, 24: 64.9 ms on bunny x20).  Rounds 2-3: 4,10 -> 29.3 / 123 ms, 16,48 -> 27.4 / 111.
    const uint32_t ladder[] = {32, 96, 240, 600, 1500};
    for (uint32_t k = 0; k < MPT_OT_MLEVELS; ++k) {
        b.trips[k] = k + 1 < MPT_OT_MLEVELS && k < 5 ? ladder[k] : 0x7FFFFFFFu;
        b.min_active[k] = k + 1 < MPT_OT_MLEVELS ? 24 : 32;
    }
    b.inplace_min = 48;   // bunny x20 256 spp: never 74.9 ms, 56 72.6, 48 72.8, 40 73.1, 32 73.7, 24 74.4; scene.xml 25.96 / 25.6 / 25.6 / 25.7 / 25.9 / 26.6
    return b;
}

struct Lane {
    hipStream_t stream = nullptr;
    QueueDev q[2] = {};
    uint32_t shard_cap = 0;
    float4* d_slots = nullptr;
    uint64_t slots_cap = 0;
    PassDesc* d_desc = nullptr;
    uint32_t* d_ctr = nullptr;
    uint32_t* h_done = nullptr;  // pinned, device-visible
    PassDesc* h_desc = nullptr;  // pinned copy of the pass descriptor (statistics read-back without a sync copy)
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_resolved = nullptr, ev_traced = nullptr;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending_timed;  // kernel-event pairs to read at collection
    WaveRings ring = {};         // wave-local pipeline: MPT_WL_LEVELS rings of MPT_WL_RING records per wave
    size_t ring_waves = 0;
    OtRings ot_ring = {};        // ... and its rings (MPT_OT_RINGS per wave)
    size_t ot_ring_waves = 0;
    uint32_t announced_id = 0;   // launch id of the lane's last trace kernel that announces its residency (0: none)
    bool in_flight = false;      // enqueued by mpt_render_async, not yet collected
    bool timed = false;
};

// mpt_render_async's submit thread (one per context, started by the first asynchronous render).  What a submission may have to wait
// for — the render lane it is going to use (two renders are in flight), the residency announcement of the trace kernel before it (the
// gate of run_pass, up to 200 ms when a foreign kernel holds the chip) — it waits for HERE, not on the caller's thread: mpt_render_async
// checks its arguments, queues the parameters and returns (round 4's version spun on the caller's thread: VERDICT r4 item 7).  The reference's
// per-frame submit is unfenced in the same way (R/Renderer/Renderer.cpp:253-266,307: commit() and return).  Every other entry point of the
// context first drains the queue (drain_submit), so the context stays single-threaded from the caller's point of view; the first
// failure of a queued render is reported by the call that drains it (mpt_wait, or whatever comes next).
struct Submitter {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv_work, cv_idle;
    std::deque<mpt_render_params> q;
    bool busy = false, stop = false;
    int rc = 0;            // first failure of a queued render since the last drain
    std::string err;
};
#define MPT_ASYNC_QUEUE_MAX 64   // queued renders beyond this make mpt_render_async wait for room (back pressure, not an error)

struct mpt_ctx {
    int device = 0;
    std::unique_ptr<Submitter> sub;      // mpt_render_async's submit thread (none until the first asynchronous render)
    uint64_t async_jobs = 0;             // renders submitted by that thread
    uint64_t gate_resident = 0;          // residency gate: the trace kernel before was resident (or finished) when asked
    uint64_t gate_timeout = 0;           // ... it was not within 200 ms: the event chain instead (a foreign kernel holds the chip)
    uint64_t async_call_us_max = 0;      // longest mpt_render_async call so far, microseconds of host time
    hipStream_t stream = nullptr;   // == lane[0].stream: uploads, clears, read-backs, mpt_draw
    Lane lane[2];
    int next_lane = 0;
    hipEvent_t last_resolved = nullptr;  // last operation on the HDR sum (resolve or clear): the next resolve waits for it
    hipEvent_t last_traced = nullptr;    // behind the last trace kernel enqueued (either lane): the next trace kernel waits for it (lane_order)
    int resolve_wgs_per_cu = 2;          // MPT_RESOLVE_WGS: workgroups of 256 per CU of a resolve that runs beside a trace kernel (0 = as many as pixels / 256)
    bool sync_render = false;            // inside mpt_render / mpt_draw: nothing else is in flight, the resolve may take the whole chip
    uint32_t launch_seq = 0;             // launch ids of the trace kernels (never 0)
    Lane* last_trace_lane = nullptr;     // lane of the last trace kernel enqueued
    int lane_order = 5;                  // MPT_LANE_ORDER: bit 0 = the event chain between trace kernels, bit 2 = the residency gate for overlapping k_wavelocal renders (run_pass), bit 1 = lane 0's stream has the higher priority (no effect measured)
    hipEvent_t ev_sum_op = nullptr;      // recorded behind mpt_clear_sum
    hipDeviceProp_t prop;
    std::string err;
    std::mutex err_mu;                   // writers of `err`: the caller's thread and the submit thread (set_err)
    // scene
    float4* d_nodes = nullptr;
    float4* d_prims = nullptr;
    float4* d_mats = nullptr;
    uint32_t n_nodes = 0, n_prims = 0, n_mats = 0, n_lds_nodes = 0, n_lds_prims = 0;
    bool have_scene = false;
    // the product's own tree (mpt_accel.h) for the closest-first pipeline; it shares d_prims with the threaded tree
    float4* d_acc_nodes = nullptr;
    float4* d_refleaf = nullptr;
    float4* d_refbox = nullptr;       // the box of its reference leaf PER PRIMITIVE, 2 float4 each: the final check of the closest-first walk needs no look-up through the primitive
    float4* d_always = nullptr;
    uint32_t n_acc_nodes = 0, n_always = 0, n_ref_leaves = 0, acc_depth = 0;
    // the two operating points of the closest-first kernel (mpt_ordered.h: k_ordered<.., 5> = 5 x 256 threads per CU, k_ordered<.., 6> = 2 x 768) and
    // what each stages in LDS for the scene at hand (size_lds_images)
    struct OtPoint {
        uint32_t threads, wgs_per_cu, lds_nodes, lds_prims;
    } ot_pt[2] = {{MPT_OT_THREADS, MPT_OT_WGS_PER_CU, 0, 0}, {MPT_OT6_THREADS, (MPT_OT6_WAVES * 256) / MPT_OT6_THREADS, 0, 0}};
    int ot_occ = 0;                  // MPT_OT_OCC: 5 / 6 forces an operating point, 0 = by scene size (ordered_point)
    bool tile_order_forced = false;  // MPT_TILE_ORDER was given
    // mpt_build_and_upload keeps its tree in the reference's buffer format on the device too (mpt_download_bvh)
    float4* d_ref_bvh = nullptr;
    int* d_ref_idx = nullptr;
    uint32_t n_ref_nodes = 0;
    uint32_t built_leaf_max = 0;     // leaf limit of that tree (mpt_build_info)
    uint32_t ot_stack_depth = 8;     // LDS stack entries per lane (MPT_OT_STACK); deeper entries spill to global memory
    mpt_lbvh::ScratchPool build_pool;  // scratch chunks of the GPU builders, kept between builds (<= 2 GiB)
    // The arrays of a scene made by mpt_build_and_upload are views into ONE device allocation (scene_block: nine hipMallocs on the
    // build's critical path were ~0.3 ms of a 4.3 ms build); the block of the scene before is kept as the next build's (spare_block),
    // so a rebuild allocates nothing and a failed build leaves the old scene intact.  Scenes of mpt_upload_scene own their arrays one by one.
    void* scene_block = nullptr;
    void* spare_block = nullptr;
    void* in_block = nullptr;          // device staging of mpt_build_and_upload's input arrays, kept between builds
    size_t in_block_bytes = 0;
    size_t scene_block_bytes = 0, spare_block_bytes = 0;
    OtBudgets ot_budgets = default_ot_budgets();
    float tri_extent = 0.0f, acc_eps_abs = 0.0f, acc_cull_rel = 9.765625e-4f;
    bool acc_ok = false;             // the closest-first pipeline may be used for this scene
    std::string acc_why;
    // uniforms / size
    mpt_uniforms u;
    bool have_uniforms = false;
    uint32_t W = 0, H = 0;
    float4* d_accum[2] = {nullptr, nullptr};
    int cur_target = 0;
    float4* d_sum_own = nullptr;
    float4* d_sum = nullptr;
    // this rank's tile processing order (x | y << 16), rebuilt when size or sharding changes
    uint32_t* d_tile_xy = nullptr;
    uint32_t tile_W = 0, tile_H = 0, tile_rank = 0, tile_nranks = 0, tile_count = 0;
    int tile_mode_built = -1;   // tile order the table was built for (tile_order_of)
    int tile_order_mode = 0;  // 0 row-major top-down (measured best with guided claims: the cheap sky tiles take the
                              // large early claims, the expensive tiles the small late ones), 1 strided, 2 bottom-up
    // literal RNG seeds
    uint32_t* d_pixel_seed = nullptr;
    float seed_rs[3] = {NAN, NAN, NAN};
    uint32_t seed_W = 0, seed_H = 0;
    // launch geometry
    int wg_size = 0;                // 0 = the kernel's own choice (MPT_WG_SIZE overrides, clamped to the kernel's bound)
    const void* occ_fun = nullptr;  // cached occupancy query
    size_t occ_lds = 0;
    int occ_per_cu = 0;
    bool time_kernels = true;
    WaveBudgets budgets = default_budgets();  // box-test loop trips per step of ring 0..3 (measured best
                                                        // ladder; an unlimited ring-3 budget leaves ring 4 unused)
    // guided path-id claims: max(wl_min, remaining / (wl_div * waves)).  wl_div, measured again after the kernels got faster
    // (serial 256-spp render of scene.xml, two runs): 16 -> 23.4-23.8 ms, 24 22.5, 32 22.3-22.4, 48 22.3-22.5, 64 22.3-22.4,
    // 128 22.5, 256 22.8; glass.xml, bunny x20, 32-spp and 640x360 renders move by +-1 % between 16 and 64
    // wl_min = 0: by the size of the pass (run_pass) — the smallest claim is 64 path ids (one step) when a wave's share of the pass is
    // small (a 1/8 shard: 11 k paths per wave) and up to 256 when it is large (a full 1080p x 256 spp pass: 86 k): scene.xml with 64 /
    // 128 / 256 / 512 / 1024 -> 16.79-16.86 / 16.70 / 16.62-16.65 / 16.65 / 17.87 ms, the 1/8 shard's serial step 2.54 / - / 2.71 / 3.36
    uint32_t wl_min = 0, wl_div = 32;
    uint32_t wl_block = MPT_WL_BLOCK;  // path ids a wave claims per atomic (multiple of 64)
    int wgs_per_cu = 0;  // 0 = as many as the occupancy query admits
    size_t lds_budget = 78 * 1024;  // per workgroup; two workgroups per CU share the 160 KiB
    mpt_stats stats = {};
};

// kernel variants: COUNT (work counters) x ALL_LDS (the whole BVH fits the LDS budget)
static const void* step_kernel(bool count, bool all_lds) {
    if (count) return all_lds ? (const void*)k_step<true, true> : (const void*)k_step<true, false>;
    return all_lds ? (const void*)k_step<false, true> : (const void*)k_step<false, false>;
}
static const void* mega_kernel(bool count, bool all_lds) {
    if (count) return all_lds ? (const void*)k_megakernel<true, true> : (const void*)k_megakernel<true, false>;
    return all_lds ? (const void*)k_megakernel<false, true> : (const void*)k_megakernel<false, false>;
}

static const void* ordered_kernel(bool count, bool all_lds, bool six = false) {
    if (six && !count) return all_lds ? (const void*)k_ordered<false, true, MPT_OT6_WAVES> : (const void*)k_ordered<false, false, MPT_OT6_WAVES>;
    if (count) return all_lds ? (const void*)k_ordered<true, true> : (const void*)k_ordered<true, false>;
    return all_lds ? (const void*)k_ordered<false, true> : (const void*)k_ordered<false, false>;
}

static const void* wavelocal_kernel(bool count, bool all_lds, bool corun) {
    if (corun && !count) return all_lds ? (const void*)k_wavelocal_corun<true> : (const void*)k_wavelocal_corun<false>;
    if (count) return all_lds ? (const void*)k_wavelocal<true, true> : (const void*)k_wavelocal<true, false>;
    return all_lds ? (const void*)k_wavelocal<false, true> : (const void*)k_wavelocal<false, false>;
}

#define MPT_LDS_MATS MPT_LDS_MATS_N               // materials staged in LDS (32 B each; mpt_kernels.h)
#define MPT_LDS_EXTRA (MPT_LDS_CFG_F4 * 16 + MPT_LDS_MATS * 32)  // material table + descriptor copy (k_step) / configuration block (k_wavelocal, k_ordered) behind the scene image

static void set_err(mpt_ctx* ctx, const std::string& msg);   // ctx->err = msg under ctx->err_mu (the submit thread may fail at the same moment)
#define HIPCHK(call)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            set_err(ctx, std::string(#call) + ": " + hipGetErrorString(e_));                \
            return MPT_ERR_HIP;                                                             \
        }                                                                                   \
    } while (0)

static void set_err(mpt_ctx* ctx, const std::string& msg) {
    std::lock_guard<std::mutex> lk(ctx->err_mu);
    ctx->err = msg;
}
static int fail(mpt_ctx* ctx, int code, const std::string& msg) {
    if (ctx) set_err(ctx, msg);
    return code;
}
static int wait_impl(mpt_ctx* ctx);  // collects the renders still in flight (defined with the render entry points)
static int drain_submit(mpt_ctx* ctx);   // waits until the renders queued by mpt_render_async have been submitted (struct Submitter)
static void stop_submit(mpt_ctx* ctx);

extern "C" const char* mpt_status_string(int s) {
    switch (s) {
        case MPT_OK: return "ok";
        case MPT_ERR_INVALID_ARG: return "invalid argument";
        case MPT_ERR_NO_DEVICE: return "no HIP device";
        case MPT_ERR_HIP: return "HIP runtime error";
        case MPT_ERR_BAD_SCENE: return "malformed scene arrays";
        case MPT_ERR_NOT_READY: return "scene, uniforms or size not set";
        case MPT_ERR_OVERFLOW: return "ray queue overflow";
    }
    return "unknown status";
}
extern "C" const char* mpt_last_error(const mpt_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

static int create_impl(int device_ordinal, mpt_ctx** out) {
    if (!out) return MPT_ERR_INVALID_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return MPT_ERR_NO_DEVICE;
    if (device_ordinal < 0 || device_ordinal >= n) return MPT_ERR_NO_DEVICE;
    mpt_ctx* ctx = new mpt_ctx();
    ctx->device = device_ordinal;
    auto bail = [&](int code) {
        mpt_destroy(ctx);
        return code;
    };
    if (hipSetDevice(device_ordinal) != hipSuccess) return bail(MPT_ERR_HIP);
    if (hipGetDeviceProperties(&ctx->prop, device_ordinal) != hipSuccess) return bail(MPT_ERR_HIP);
    if (const char* lo = getenv("MPT_LANE_ORDER")) ctx->lane_order = atoi(lo);
    if (const char* rw = getenv("MPT_RESOLVE_WGS")) ctx->resolve_wgs_per_cu = std::max(0, atoi(rw));
    int prio_lo = 0, prio_hi = 0;   // (numerically lower = higher priority)
    if (hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) != hipSuccess) prio_lo = prio_hi = 0;
    for (Lane& L : ctx->lane) {  // small per-lane state now; slots, rings and queues are allocated on first use
        if (ctx->lane_order & 2) {
            if (hipStreamCreateWithPriority(&L.stream, hipStreamNonBlocking, &L == &ctx->lane[0] ? prio_hi : prio_lo) != hipSuccess) return bail(MPT_ERR_HIP);
        } else if (hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking) != hipSuccess) return bail(MPT_ERR_HIP);
        if (hipMalloc(&L.d_desc, sizeof(PassDesc)) != hipSuccess) return bail(MPT_ERR_HIP);
        if (hipMemset(L.d_desc, 0, sizeof(PassDesc)) != hipSuccess) return bail(MPT_ERR_HIP);
        if (hipMalloc(&L.d_ctr, MPT_CTR_WORDS * 4) != hipSuccess) return bail(MPT_ERR_HIP);
        if (hipMemset(L.d_ctr, 0, MPT_CTR_WORDS * 4) != hipSuccess) return bail(MPT_ERR_HIP);
        if (hipHostMalloc((void**)&L.h_done, 64, hipHostMallocMapped) != hipSuccess) return bail(MPT_ERR_HIP);
        if (hipHostMalloc((void**)&L.h_desc, sizeof(PassDesc), hipHostMallocDefault) != hipSuccess) return bail(MPT_ERR_HIP);
        memset((void*)L.h_done, 0, 64);   // (every word: the residency gate polls word MPT_HOST_RESIDENT against small launch ids — ADVICE r4)
        if (hipEventCreate(&L.ev0) != hipSuccess || hipEventCreate(&L.ev1) != hipSuccess ||
            hipEventCreate(&L.ev_resolved) != hipSuccess || hipEventCreateWithFlags(&L.ev_traced, hipEventDisableTiming) != hipSuccess)
            return bail(MPT_ERR_HIP);
    }
    ctx->stream = ctx->lane[0].stream;
    if (hipEventCreate(&ctx->ev_sum_op) != hipSuccess) return bail(MPT_ERR_HIP);
    const char* e;
    if ((e = getenv("MPT_WG_SIZE"))) ctx->wg_size = atoi(e);
    if ((e = getenv("MPT_WGS_PER_CU"))) ctx->wgs_per_cu = atoi(e);
    if ((e = getenv("MPT_LDS_BYTES"))) ctx->lds_budget = (size_t)atol(e);
    ctx->time_kernels = !((e = getenv("MPT_NO_KERNEL_EVENTS")) && atoi(e));
    if ((e = getenv("MPT_WL_BLOCK")) && atoi(e) >= 64) ctx->wl_block = (uint32_t)atoi(e) & ~63u;
    if ((e = getenv("MPT_BUDGETS"))) {  // e.g. "8,20,50,125": box-test trips per step of ring 0, 1, ... (the last ring has none)
        unsigned prev = 4;
        const char* q = e;
        for (uint32_t k = 0; k + 1 < MPT_WL_LEVELS; ++k) {
            unsigned v = 0;
            int used = 0;
            if (q && sscanf(q, "%u%n", &v, &used) == 1) {
                q += used;
                if (*q == ',') ++q;
            } else {
                v = prev * 2;
                q = nullptr;
            }
            ctx->budgets.b[k] = v < 1 ? 1 : v;
            prev = ctx->budgets.b[k];
        }
    }
    if ((e = getenv("MPT_MIN_ACTIVE"))) {  // "a,b,c,d,e": lanes that must still be traversing for a step of ring k to go on
        const char* q = e;
        unsigned v = 0;
        for (uint32_t k = 0; k < MPT_WL_LEVELS; ++k) {
            int used = 0;
            if (q && sscanf(q, "%u%n", &v, &used) == 1) {
                q += used;
                if (*q == ',') ++q;
            } else {
                q = nullptr;
            }
            ctx->budgets.min_active[k] = v > 64 ? 64 : v;
        }
    }
    if ((e = getenv("MPT_LIGHT_BUDGET")) && atoi(e) >= 1) {  // one number: geometric ladder b, 2b, 4b, 8b
        uint32_t b = (uint32_t)atoi(e);
        for (uint32_t k = 0; k + 1 < MPT_WL_LEVELS; ++k) ctx->budgets.b[k] = b > (1u << 28) ? b : std::min<uint64_t>((uint64_t)b << k, MPT_WL_NO_BUDGET);
    }
    if ((e = getenv("MPT_WL_MIN")) && atoi(e) >= 64) ctx->wl_min = (uint32_t)atoi(e) & ~63u;
    if ((e = getenv("MPT_TILE_ORDER"))) {
        ctx->tile_order_mode = atoi(e);
        ctx->tile_order_forced = true;
    }
    if ((e = getenv("MPT_OT_OCC")) && (atoi(e) == 5 || atoi(e) == 6)) ctx->ot_occ = atoi(e);
    if ((e = getenv("MPT_WL_DIV")) && atoi(e) >= 1) ctx->wl_div = (uint32_t)atoi(e);
    if ((e = getenv("MPT_OT_STACK")) && atoi(e) >= 2 && atoi(e) <= (int)MPT_OT_PARK) ctx->ot_stack_depth = (uint32_t)atoi(e);
    for (int which = 0; which < 2; ++which)
        if ((e = getenv(which ? "MPT_OT_MIN_ACTIVE" : "MPT_OT_BUDGETS"))) {  // "a,b,c": one number per tree-walk ring
            const char* q = e;
            for (uint32_t k = 0; k < MPT_OT_MLEVELS && q; ++k) {
                unsigned v = 0;
                int used = 0;
                if (sscanf(q, "%u%n", &v, &used) != 1) break;
                q += used;
                if (*q == ',') ++q;
                if (which) ctx->ot_budgets.min_active[k] = v > 64 ? 64 : v;
                else if (k + 1 < MPT_OT_MLEVELS) ctx->ot_budgets.trips[k] = v < 1 ? 1 : v;
            }
        }
    if ((e = getenv("MPT_OT_INPLACE")) && atoi(e) >= 1) ctx->ot_budgets.inplace_min = (uint32_t)atoi(e);
    if ((e = getenv("MPT_OT_CULL_REL")) && atof(e) > 0.0) ctx->acc_cull_rel = (float)atof(e);
    if (ctx->wg_size < 64 || ctx->wg_size > 1024 || (ctx->wg_size & 63)) ctx->wg_size = 0;
    if (ctx->lds_budget > 160 * 1024) ctx->lds_budget = 160 * 1024;
    // allow the full 160 KiB of dynamic LDS
    for (int c = 0; c < 2; ++c)
        for (int a = 0; a < 2; ++a) {
            hipFuncSetAttribute(step_kernel(c, a), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute(mega_kernel(c, a), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute(wavelocal_kernel(c, a, false), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (!c) hipFuncSetAttribute(wavelocal_kernel(false, a, true), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        }
    for (int c = 0; c < 2; ++c)
        for (int a = 0; a < 2; ++a) {
            hipFuncSetAttribute(ordered_kernel(c, a), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (!c) hipFuncSetAttribute(ordered_kernel(false, a, true), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        }
    hipFuncSetAttribute((const void*)k_trace_rays_ordered, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k_trace_rays, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    *out = ctx;
    return MPT_OK;
}

// Lets go of the scene's device arrays: one by one (mpt_upload_scene's), or the block they are views into (mpt_build_and_upload's), which
// becomes the spare block of the next build if it is not larger than 1 GiB and larger than the spare there is.
static void free_scene_buffers(mpt_ctx* ctx, bool keep_spare) {
    if (ctx->scene_block) {
        // (the larger of the two is the one worth keeping: a spare that is too small for the scenes being built is never taken)
        if (keep_spare && ctx->scene_block_bytes <= ((size_t)1 << 30) && (!ctx->spare_block || ctx->scene_block_bytes > ctx->spare_block_bytes)) {
            hipFree(ctx->spare_block);
            ctx->spare_block = ctx->scene_block;
            ctx->spare_block_bytes = ctx->scene_block_bytes;
        } else {
            hipFree(ctx->scene_block);
        }
        ctx->scene_block = nullptr;
        ctx->scene_block_bytes = 0;
    } else {
        hipFree(ctx->d_nodes);
        hipFree(ctx->d_prims);
        hipFree(ctx->d_mats);
        hipFree(ctx->d_acc_nodes);
        hipFree(ctx->d_refbox);
        hipFree(ctx->d_refleaf);
        hipFree(ctx->d_always);
        hipFree(ctx->d_ref_bvh);
        hipFree(ctx->d_ref_idx);
    }
    ctx->d_nodes = ctx->d_prims = ctx->d_mats = ctx->d_acc_nodes = ctx->d_refbox = ctx->d_refleaf = ctx->d_always = ctx->d_ref_bvh = nullptr;
    ctx->d_ref_idx = nullptr;
}

static void free_ot_rings(OtRings& r) {
    hipFree(r.base);
    r = OtRings{};
}
static void free_queues(Lane& L) {
    for (int i = 0; i < 2; ++i) {
        hipFree(L.q[i].od);
        hipFree(L.q[i].dt);
        hipFree(L.q[i].tl);
        hipFree(L.q[i].ia);
        L.q[i] = QueueDev{};
    }
    L.shard_cap = 0;
}

extern "C" int mpt_destroy(mpt_ctx* ctx) {
    if (!ctx) return MPT_ERR_INVALID_ARG;
    stop_submit(ctx);   // (renders still queued are submitted, then the thread ends)
    hipSetDevice(ctx->device);
    for (Lane& L : ctx->lane)
        if (L.stream) hipStreamSynchronize(L.stream);
    free_scene_buffers(ctx, false);
    hipFree(ctx->spare_block);
    hipFree(ctx->in_block);
    hipFree(ctx->d_accum[0]);
    hipFree(ctx->d_accum[1]);
    hipFree(ctx->d_sum_own);
    hipFree(ctx->d_pixel_seed);
    hipFree(ctx->d_tile_xy);
    for (Lane& L : ctx->lane) {
        hipFree(L.d_slots);
        hipFree(L.d_desc);
        hipFree(L.d_ctr);
        hipFree(L.ring.base);
        free_ot_rings(L.ot_ring);
        free_queues(L);
        if (L.h_done) hipHostFree(L.h_done);
        if (L.h_desc) hipHostFree(L.h_desc);
        if (L.ev0) hipEventDestroy(L.ev0);
        if (L.ev1) hipEventDestroy(L.ev1);
        if (L.ev_resolved) hipEventDestroy(L.ev_resolved);
        if (L.ev_traced) hipEventDestroy(L.ev_traced);
        for (auto e : L.ev_pool) hipEventDestroy(e);
        if (L.stream) hipStreamDestroy(L.stream);
    }
    if (ctx->ev_sum_op) hipEventDestroy(ctx->ev_sum_op);
    delete ctx;
    return MPT_OK;
}

// ---- scene upload: reference arrays -> threaded, breadth-first, leaf-ordered device layout ----------------
namespace {
struct HostNode {
    float bmin[3], bmax[3];
    int leftFirst, count;
};
static inline int bits_to_int(float f) {
    int i;
    memcpy(&i, &f, 4);
    return i;
}
static inline float int_to_bits(int i) {
    float f;
    memcpy(&f, &i, 4);
    return f;
}
}  // namespace

// What each kernel stages in LDS, from the sizes of the uploaded scene.
static void size_lds_images(mpt_ctx* ctx) {
    // LDS image of the reference-order kernels = top of the tree + primitives of the shallowest leaves.  When the whole
    // tree fits, the rest of the budget goes to primitives; otherwise 6 KiB are reserved for them (the leaves next to the
    // root are visited by almost every ray: on scene.xml the three spheres take 65 % of all primitive tests).
    {
        const size_t budget = ctx->lds_budget > MPT_LDS_EXTRA ? ctx->lds_budget - MPT_LDS_EXTRA : 0;
        const size_t all_nodes = (size_t)ctx->n_nodes * 32, all_prims = (size_t)ctx->n_prims * 48;
        size_t prim_bytes = all_nodes <= budget ? std::min(all_prims, budget - all_nodes)
                                                : std::min<size_t>(all_prims, std::min<size_t>(6 * 1024, budget / 4));
        prim_bytes -= prim_bytes % 48;
        ctx->n_lds_prims = (uint32_t)(prim_bytes / 48);
        ctx->n_lds_nodes = (uint32_t)std::min<size_t>(ctx->n_nodes, (budget - prim_bytes) / 32);
    }
    // LDS image of the closest-first kernel at each of its operating points (wgs_per_cu workgroups of `threads` share a CU's 160 KiB):
    // the stacks, then as many own nodes as fit (breadth-first prefix), the always list, and primitives with what is left.
    for (mpt_ctx::OtPoint& pt : ctx->ot_pt) {
        const size_t stacks = (size_t)pt.threads * ctx->ot_stack_depth * 8;
        // (two workgroups per CU get 78 KiB each, not 80: the allocation granule must leave both room)
        const size_t avail = pt.wgs_per_cu == 1 ? 160 * 1024 : 156 * 1024 / pt.wgs_per_cu, fixed = MPT_LDS_EXTRA + stacks + (size_t)ctx->n_always * 80;
        const size_t total = avail > fixed + 112 ? avail - fixed : 112;
        const size_t all_nodes = (size_t)ctx->n_acc_nodes * 112, all_prims = (size_t)ctx->n_prims * 48;
        size_t prim_bytes = all_nodes <= total ? std::min(all_prims, total - all_nodes)
                                               : std::min<size_t>(all_prims, std::min<size_t>(4 * 1024, total / 4));
        if (const char* e = getenv("MPT_OT_LDS_PRIMS")) prim_bytes = std::min<size_t>(prim_bytes, (size_t)atoi(e) * 48);
        prim_bytes -= prim_bytes % 48;
        pt.lds_prims = (uint32_t)(prim_bytes / 48);
        pt.lds_nodes = (uint32_t)std::min<size_t>(ctx->n_acc_nodes, (total - prim_bytes) / 112);
    }
}
// Which operating point of the closest-first kernel renders this scene: the six-wave one up to MPT_OT6_MAX_PRIMS primitives (the
// sixth wave pays while a compute die's L2 can mostly hold the tree: 40 k / 99 k / 397 k / 596 k / 795 k primitives -4.4 / -4.0 / -2.7 /
// -1.5 / -0.7 % with the striped tile order, 994 k +1.7 %, the 1 M-triangle height fields of configs[4] +1.5 .. +7 % — mpt_ordered.h,
// gpurun_out/r05/s7, s9, s12), the five-wave one beyond and for counted renders.
#define MPT_OT6_MAX_PRIMS 800000u
static int ordered_point(const mpt_ctx* ctx, bool count) {
    if (count) return 0;
    if (ctx->ot_occ == 5) return 0;
    if (ctx->ot_occ == 6) return 1;
    return ctx->n_prims <= MPT_OT6_MAX_PRIMS ? 1 : 0;
}

// the per-primitive reference-leaf boxes of an uploaded scene (k_prim_refbox, mpt_devbuild.h)
static int make_prim_refbox(mpt_ctx* ctx) {
    hipFree(ctx->d_refbox);
    ctx->d_refbox = nullptr;
    if (ctx->n_prims == 0 || ctx->d_refleaf == nullptr || ctx->n_ref_leaves == 0) return MPT_OK;
    HIPCHK(hipMalloc(&ctx->d_refbox, (size_t)ctx->n_prims * 32));
    hipLaunchKernelGGL(mpt_devbuild::k_prim_refbox, dim3((ctx->n_prims + 255u) / 256u), dim3(256), 0, ctx->stream, (const float4*)ctx->d_prims,
                       (const float4*)ctx->d_refleaf, ctx->n_prims, ctx->d_refbox);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return MPT_OK;
}
static int upload_scene_impl(mpt_ctx* ctx, const float* bvh, uint64_t n_nodes, const float* prims,
                                const float* mats, const int32_t* prim_idx, uint64_t n_prims) {
    if (!ctx) return MPT_ERR_INVALID_ARG;
    if (!bvh || !prims || !mats || !prim_idx || n_nodes == 0 || n_prims == 0)
        return fail(ctx, MPT_ERR_INVALID_ARG, "null or empty scene array");
    if (n_nodes >= (1ull << 30) || n_prims >= (1ull << 27))
        return fail(ctx, MPT_ERR_BAD_SCENE, "scene too large for the 27-bit leaf encoding");
    {
        int wrc = wait_impl(ctx);  // renders in flight still read the old scene
        if (wrc) return wrc;
    }
    const uint32_t N = (uint32_t)n_nodes, P = (uint32_t)n_prims;

    // 1. walk the reference tree in ITS visit order (root; right subtree; left subtree —
    //    PathTracing.h:188-193 pushes left then right, so right pops first) and record the order.
    std::vector<uint32_t> order;   // visit position -> reference node
    std::vector<uint32_t> sub_end; // visit position -> visit position just past its subtree
    order.reserve(N);
    std::vector<uint8_t> seen(N, 0);
    {
        // iterative DFS with explicit post-processing for subtree ends
        struct Fr {
            uint32_t node, pos;
            int stage;
        };
        std::vector<Fr> st;
        st.push_back({0u, 0u, 0});
        sub_end.assign(N, 0);
        while (!st.empty()) {
            Fr& f = st.back();
            if (f.stage == 0) {
                if (f.node >= N || seen[f.node]) return fail(ctx, MPT_ERR_BAD_SCENE, "BVH is not a tree (cycle or index out of range)");
                seen[f.node] = 1;
                f.pos = (uint32_t)order.size();
                order.push_back(f.node);
                int count = bits_to_int(bvh[8 * (size_t)f.node + 7]);
                if (count > 0) {
                    int first = bits_to_int(bvh[8 * (size_t)f.node + 3]);
                    if (first < 0 || (uint64_t)first + (uint64_t)count > n_prims)
                        return fail(ctx, MPT_ERR_BAD_SCENE, "leaf primitive range out of bounds");
                    sub_end[f.pos] = f.pos + 1;
                    st.pop_back();
                } else {
                    f.stage = 1;
                    uint32_t right = (uint32_t)(-(long long)count);
                    if (count == 0) return fail(ctx, MPT_ERR_BAD_SCENE, "internal node with right child 0");
                    st.push_back({right, 0u, 0});
                }
            } else if (f.stage == 1) {
                f.stage = 2;
                int left = bits_to_int(bvh[8 * (size_t)f.node + 3]);
                if (left < 0) return fail(ctx, MPT_ERR_BAD_SCENE, "negative left child");
                st.push_back({(uint32_t)left, 0u, 0});
            } else {
                sub_end[f.pos] = (uint32_t)order.size();
                st.pop_back();
            }
        }
    }
    const uint32_t NV = (uint32_t)order.size();  // reachable nodes

    // 1b. every child box must lie inside its parent's box: then the reference's walk equals a scan over the leaves in
    //     visit order (mpt_ordered.h) and the closest-first pipeline may be used; otherwise only the reference-order ones.
    bool nested = true;
    for (uint32_t pos = 0; pos < NV && nested; ++pos) {
        const float* n = bvh + 8 * (size_t)order[pos];
        if (bits_to_int(n[7]) > 0) continue;
        const uint32_t kids[2] = {pos + 1, pos + 1 < NV ? sub_end[pos + 1] : NV};
        for (uint32_t k : kids) {
            if (k >= NV) continue;
            const float* c = bvh + 8 * (size_t)order[k];
            for (int a = 0; a < 3; ++a)
                if (!(c[a] >= n[a] && c[4 + a] <= n[4 + a])) nested = false;
        }
    }

    // 2. leaves: gather primitives into leaf order; split leaves of more than 16 primitives into a chain.
    //    Device node list in VISIT order first (dn), then permuted breadth-first.
    struct DNode {
        float bmin[3], bmax[3];
        bool leaf;
        uint32_t first, count;   // leaf
        uint32_t hit, miss;      // links as indices into dn (visit order); N_total = end
        uint32_t ref_leaf;       // leaf: number of the reference leaf it came from (visit order)
    };
    std::vector<DNode> dn;
    dn.reserve(NV + 16);
    std::vector<uint32_t> pos_to_dn(NV + 1, 0);  // visit position -> dn index of its (first) node
    std::vector<float> dprims;
    dprims.reserve((size_t)P * 12);
    std::vector<float> mat_table;
    std::vector<float> refleaf;  // 8 floats per reference leaf: (bmin, 0) (bmax, 0)
    std::map<std::vector<uint32_t>, uint32_t> mat_lookup;
    auto mat_index = [&](uint32_t pid) -> uint32_t {
        std::vector<uint32_t> key(8);
        memcpy(key.data(), mats + 8 * (size_t)pid, 32);
        auto it = mat_lookup.find(key);
        if (it != mat_lookup.end()) return it->second;
        uint32_t id = (uint32_t)(mat_table.size() / 8);
        mat_table.insert(mat_table.end(), mats + 8 * (size_t)pid, mats + 8 * (size_t)pid + 8);
        mat_lookup.emplace(std::move(key), id);
        return id;
    };
    uint32_t n_spheres = 0;
    float tri_extent = 0.0f;  // largest |coordinate| of a triangle vertex
    // first pass: create dn entries; a long leaf becomes ceil(count/16) chained nodes
    for (uint32_t pos = 0; pos < NV; ++pos) {
        const float* n = bvh + 8 * (size_t)order[pos];
        int count = bits_to_int(n[7]);
        pos_to_dn[pos] = (uint32_t)dn.size();
        DNode d;
        memcpy(d.bmin, n, 12);
        memcpy(d.bmax, n + 4, 12);
        d.hit = d.miss = 0;
        d.ref_leaf = 0;
        if (count > 0) {
            int first = bits_to_int(n[3]);
            d.leaf = true;
            d.ref_leaf = (uint32_t)(refleaf.size() / 8);
            const float rl[8] = {n[0], n[1], n[2], 0.0f, n[4], n[5], n[6], 0.0f};
            refleaf.insert(refleaf.end(), rl, rl + 8);
            for (int k0 = 0; k0 < count; k0 += 16) {
                d.first = (uint32_t)(dprims.size() / 12);
                d.count = (uint32_t)std::min(16, count - k0);
                for (uint32_t k = 0; k < d.count; ++k) {
                    int32_t pid = prim_idx[first + k0 + (int)k];
                    if (pid < 0 || (uint64_t)pid >= n_prims) return fail(ctx, MPT_ERR_BAD_SCENE, "primitive index out of range");
                    const float* p = prims + 12 * (size_t)pid;
                    float rec[12];
                    int type = (int)p[3];
                    uint32_t m = mat_index((uint32_t)pid);
                    if (type == 1) {  // triangle: v0, e1 = v1 - v0, e2 = v2 - v0 (PathTracing.h:149-150)
                        rec[0] = p[0]; rec[1] = p[1]; rec[2] = p[2];
                        rec[4] = p[4] - p[0]; rec[5] = p[5] - p[1]; rec[6] = p[6] - p[2];
                        rec[8] = p[8] - p[0]; rec[9] = p[9] - p[1]; rec[10] = p[10] - p[2];
                        for (int q = 0; q < 11; ++q)
                            if ((q & 3) != 3 && std::isfinite(p[q])) tri_extent = std::max(tri_extent, fabsf(p[q]));
                    } else {  // sphere; anything that is neither is never hit (PathTracing.h:120,143) — kept as a sphere of radius NaN
                        rec[0] = p[0]; rec[1] = p[1]; rec[2] = p[2];
                        rec[4] = type == 0 ? p[4] : NAN; rec[5] = 0; rec[6] = 0;
                        rec[8] = 0; rec[9] = 0; rec[10] = 0;
                        n_spheres++;
                    }
                    rec[3] = int_to_bits((int)((d.ref_leaf << 1) | (type == 1 ? 1u : 0u)));
                    rec[7] = int_to_bits((int)m);
                    rec[11] = int_to_bits(pid);
                    dprims.insert(dprims.end(), rec, rec + 12);
                }
                dn.push_back(d);
            }
        } else {
            d.leaf = false;
            d.first = d.count = 0;
            dn.push_back(d);
        }
    }
    pos_to_dn[NV] = (uint32_t)dn.size();
    const uint32_t ND = (uint32_t)dn.size();
    // second pass: links.  In visit order the node after position pos is pos+1; a missed internal node
    // skips to sub_end[pos].
    for (uint32_t pos = 0; pos < NV; ++pos) {
        uint32_t a = pos_to_dn[pos], b = pos_to_dn[pos + 1];
        if (dn[a].leaf) {
            for (uint32_t k = a; k < b; ++k) dn[k].hit = dn[k].miss = k + 1;  // chain, then the next position
        } else {
            dn[a].hit = a + 1;
            dn[a].miss = pos_to_dn[sub_end[pos]];
        }
    }

    // 3. the product's own tree over the leaves (mpt_accel.h): spheres go to the always list (their t has no usable
    //    error bound: the r = 10^4 ground sphere), every leaf that holds triangles becomes an item whose box contains
    //    the reference leaf box — or, for the triangles that share a leaf with a sphere, a tight box of their own.
    const bool use_always = n_spheres <= MPT_ACCEL_MAX_ALWAYS;
    const float pad = std::max(tri_extent, 1e-6f) * 6.103515625e-05f;  // 2^-14: covers the rcp / fma slab arithmetic
    std::vector<mpt_accel::Item> items;
    std::vector<uint32_t> item_dn;  // item -> dn index
    std::vector<uint32_t> sphere_only_dn;
    std::vector<float> always;
    for (uint32_t i = 0; i < ND; ++i) {
        const DNode& d = dn[i];
        if (!d.leaf) continue;
        mpt_accel::Box tb = mpt_accel::empty_box();
        uint32_t ntri = 0, nsph = 0;
        for (uint32_t k = 0; k < d.count; ++k) {
            const float* r = dprims.data() + (size_t)(d.first + k) * 12;
            if (bits_to_int(r[3]) & 1) {
                ntri++;
                const float v[3][3] = {{r[0], r[1], r[2]}, {r[0] + r[4], r[1] + r[5], r[2] + r[6]}, {r[0] + r[8], r[1] + r[9], r[2] + r[10]}};
                for (auto& q : v) {
                    mpt_accel::Box c;
                    memcpy(c.lo, q, 12);
                    memcpy(c.hi, q, 12);
                    mpt_accel::grow(tb, c);
                }
            } else {
                nsph++;
            }
        }
        if (ntri == 0 && use_always) {
            sphere_only_dn.push_back(i);
            continue;
        }
        mpt_accel::Item it;
        memcpy(it.box.lo, d.bmin, 12);
        memcpy(it.box.hi, d.bmax, 12);
        if (nsph != 0 && use_always) {  // a sphere's leaf-mates: v0 + e is not exactly the vertex, hence the generous margin
            float ext = 0.0f;
            for (int a = 0; a < 3; ++a) ext = std::max(ext, tb.hi[a] - tb.lo[a]);
            for (int a = 0; a < 3; ++a) {
                it.box.lo[a] = std::max(d.bmin[a], tb.lo[a] - 0.05f * ext - pad);
                it.box.hi[a] = std::min(d.bmax[a], tb.hi[a] + 0.05f * ext + pad);
            }
        }
        for (int a = 0; a < 3; ++a) {
            it.box.lo[a] -= pad;
            it.box.hi[a] += pad;
        }
        it.count = d.count;
        it.key = i;
        items.push_back(it);
        item_dn.push_back(i);
    }
    const mpt_accel::Topology topo = mpt_accel::build_topology(items);

    // 4. ONE primitive array for both structures: leaves without an item (spheres only) first — the reference-order
    //    kernels test them for almost every ray, so they belong to the LDS-staged prefix — then the items in the
    //    breadth-first order of the own tree (shallow leaves first).
    std::vector<uint32_t> first_of(items.size(), 0);
    {
        std::vector<float> ordered(dprims.size());
        uint32_t at = 0;
        auto place = [&](DNode& d) {
            memcpy(ordered.data() + (size_t)at * 12, dprims.data() + (size_t)d.first * 12, (size_t)d.count * 48);
            d.first = at;
            at += d.count;
        };
        for (uint32_t i : sphere_only_dn) place(dn[i]);
        for (uint32_t it : topo.item_order) {
            place(dn[item_dn[it]]);
            first_of[it] = dn[item_dn[it]].first;
        }
        dprims.swap(ordered);
    }
    const std::vector<float> acc_nodes = mpt_accel::emit(topo, items, first_of);
    if (use_always)
        for (size_t i = 0; i < dprims.size() / 12; ++i) {
            const float* r = dprims.data() + i * 12;
            if (bits_to_int(r[3]) & 1) continue;
            float rec[20];
            memcpy(rec, r, 48);
            const uint32_t k = (uint32_t)(always.size() / 20), leaf = (uint32_t)bits_to_int(r[3]) >> 1;
            rec[5] = int_to_bits((int)i);  // own position in the primitive array = what the walk reports as the winner
            rec[6] = int_to_bits((int)k);
            memcpy(rec + 12, refleaf.data() + 8 * (size_t)leaf, 32);  // the box of its reference leaf, for the final check
            always.insert(always.end(), rec, rec + 20);
            dprims[i * 12 + 6] = int_to_bits((int)k);  // the primitive record points back at its always-list entry
        }

    // 5. reference-order structure: breadth-first permutation of the threaded nodes (top of the tree first -> LDS).
    std::vector<uint32_t> depth(ND, 0);
    {
        // depth by walking positions: children of internal at pos are pos+1 (right) and sub_end[pos+1] (left)
        std::vector<uint32_t> pdepth(NV, 0);
        for (uint32_t pos = 0; pos < NV; ++pos) {
            uint32_t a = pos_to_dn[pos];
            if (!dn[a].leaf) {
                uint32_t r = pos + 1;
                if (r < NV) {
                    pdepth[r] = pdepth[pos] + 1;
                    uint32_t l = sub_end[r];
                    if (l < NV && l < sub_end[pos]) pdepth[l] = pdepth[pos] + 1;
                }
            }
            for (uint32_t k = a; k < pos_to_dn[pos + 1]; ++k) depth[k] = pdepth[pos];
        }
    }
    std::vector<uint32_t> perm(ND);  // new index -> dn index
    for (uint32_t i = 0; i < ND; ++i) perm[i] = i;
    std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return depth[a] < depth[b]; });
    // the root must stay at index 0 (depth 0, unique) — guaranteed by the stable sort
    std::vector<uint32_t> inv(ND + 1);
    for (uint32_t i = 0; i < ND; ++i) inv[perm[i]] = i;
    inv[ND] = ND;  // terminator
    std::vector<float> dnodes((size_t)ND * 8);
    for (uint32_t i = 0; i < ND; ++i) {
        const DNode& d = dn[perm[i]];
        float* o = dnodes.data() + 8 * (size_t)i;
        memcpy(o, d.bmin, 12);
        memcpy(o + 4, d.bmax, 12);
        if (d.leaf) {  // hit link = "hold this leaf" (MPT_NODE_HOLD | first << 4 | count - 1), miss link = the next node either way
            o[3] = int_to_bits((int)(0x80000000u | (d.first * 16u + (d.count - 1u))));
            o[7] = int_to_bits((int)inv[d.hit]);
        } else {
            o[3] = int_to_bits((int)inv[d.hit]);
            o[7] = int_to_bits((int)inv[d.miss]);
        }
    }

    free_scene_buffers(ctx, true);
    ctx->n_ref_nodes = 0;
    ctx->built_leaf_max = 0;
    ctx->have_scene = false;
    auto up = [&](float4** dst, const std::vector<float>& v, size_t min_floats) -> hipError_t {
        hipError_t e = hipMalloc(dst, std::max(v.size(), min_floats) * 4);
        if (e != hipSuccess || v.empty()) return e;
        return hipMemcpy(*dst, v.data(), v.size() * 4, hipMemcpyHostToDevice);
    };
    HIPCHK(up(&ctx->d_nodes, dnodes, 8));
    HIPCHK(up(&ctx->d_prims, dprims, 12));
    HIPCHK(up(&ctx->d_mats, mat_table, 8));
    HIPCHK(up(&ctx->d_acc_nodes, acc_nodes, MPT_ACCEL_NODE_FLOATS));
    HIPCHK(up(&ctx->d_refleaf, refleaf, 8));
    HIPCHK(up(&ctx->d_always, always, 20));
    ctx->n_nodes = ND;
    ctx->n_prims = (uint32_t)(dprims.size() / 12);
    ctx->n_mats = (uint32_t)(mat_table.size() / 8);
    ctx->n_acc_nodes = (uint32_t)(acc_nodes.size() / MPT_ACCEL_NODE_FLOATS);
    ctx->n_always = (uint32_t)(always.size() / 20);
    ctx->n_ref_leaves = (uint32_t)(refleaf.size() / 8);
    {
        int rrc = make_prim_refbox(ctx);
        if (rrc) return rrc;
    }
    ctx->acc_depth = topo.depth;
    ctx->tri_extent = tri_extent;
    ctx->acc_eps_abs = tri_extent * 3.814697265625e-06f;  // 2^-18 of the largest triangle coordinate
    ctx->acc_ok = nested && use_always;
    ctx->acc_why = !nested ? "a child box is not inside its parent's box" : !use_always ? "more than 16 spheres" : "";
    size_lds_images(ctx);
    ctx->have_scene = true;
    return MPT_OK;
}

static int set_uniforms_impl(mpt_ctx* ctx, const mpt_uniforms* u) {
    if (!ctx || !u) return MPT_ERR_INVALID_ARG;
    ctx->u = *u;
    ctx->have_uniforms = true;
    return MPT_OK;
}

static int resize_impl(mpt_ctx* ctx, uint32_t width, uint32_t height) {
    if (!ctx || width == 0 || height == 0 || (uint64_t)width * height >= (1ull << 31))
        return fail(ctx, MPT_ERR_INVALID_ARG, "bad size");
    {
        int wrc = wait_impl(ctx);
        if (wrc) return wrc;
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 2; ++i) {
        hipFree(ctx->d_accum[i]);
        ctx->d_accum[i] = nullptr;
    }
    hipFree(ctx->d_sum_own);
    ctx->d_sum_own = nullptr;
    size_t bytes = (size_t)width * height * 16;
    for (int i = 0; i < 2; ++i) {
        HIPCHK(hipMalloc(&ctx->d_accum[i], bytes));
        HIPCHK(hipMemset(ctx->d_accum[i], 0, bytes));
    }
    HIPCHK(hipMalloc(&ctx->d_sum_own, bytes));
    HIPCHK(hipMemset(ctx->d_sum_own, 0, bytes));
    ctx->d_sum = ctx->d_sum_own;
    ctx->W = width;
    ctx->H = height;
    ctx->cur_target = 0;
    return MPT_OK;
}

extern "C" int mpt_sum_buffer(mpt_ctx* ctx, void** p, uint64_t* bytes) {
    if (!ctx || !p) return MPT_ERR_INVALID_ARG;
    if (!ctx->d_sum) return fail(ctx, MPT_ERR_NOT_READY, "mpt_resize not called");
    *p = ctx->d_sum;
    if (bytes) *bytes = (uint64_t)ctx->W * ctx->H * 16;
    return MPT_OK;
}
extern "C" int mpt_set_sum_buffer(mpt_ctx* ctx, void* p) {
    if (!ctx) return MPT_ERR_INVALID_ARG;
    int wrc = drain_submit(ctx);
    if (wrc) return wrc;
    wrc = wait_impl(ctx);
    if (wrc) return wrc;
    ctx->d_sum = p ? (float4*)p : ctx->d_sum_own;
    return MPT_OK;
}
static int clear_sum_impl(mpt_ctx* ctx) {
    if (!ctx) return MPT_ERR_INVALID_ARG;
    if (!ctx->d_sum) return fail(ctx, MPT_ERR_NOT_READY, "mpt_resize not called");
    int wrc = wait_impl(ctx);
    if (wrc) return wrc;
    HIPCHK(hipMemsetAsync(ctx->d_sum, 0, (size_t)ctx->W * ctx->H * 16, ctx->stream));
    HIPCHK(hipEventRecord(ctx->ev_sum_op, ctx->stream));  // the next resolve (on either lane) is ordered behind the clear
    ctx->last_resolved = ctx->ev_sum_op;
    return MPT_OK;
}
extern "C" void* mpt_stream(mpt_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }
extern "C" int mpt_synchronize(mpt_ctx* ctx) {
    if (!ctx) return MPT_ERR_INVALID_ARG;
    int wrc = drain_submit(ctx);
    if (wrc) return wrc;
    wrc = wait_impl(ctx);
    if (wrc) return wrc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return MPT_OK;
}
static int read_frame_impl(mpt_ctx* ctx, float* out) {
    if (!ctx || !out) return MPT_ERR_INVALID_ARG;
    if (!ctx->d_accum[0]) return fail(ctx, MPT_ERR_NOT_READY, "mpt_resize not called");
    int wrc = wait_impl(ctx);
    if (wrc) return wrc;
    HIPCHK(hipMemcpyAsync(out, ctx->d_accum[ctx->cur_target], (size_t)ctx->W * ctx->H * 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return MPT_OK;
}
static int read_sum_impl(mpt_ctx* ctx, float* out) {
    if (!ctx || !out) return MPT_ERR_INVALID_ARG;
    if (!ctx->d_sum) return fail(ctx, MPT_ERR_NOT_READY, "mpt_resize not called");
    int wrc = wait_impl(ctx);
    if (wrc) return wrc;
    HIPCHK(hipMemcpyAsync(out, ctx->d_sum, (size_t)ctx->W * ctx->H * 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return MPT_OK;
}
extern "C" int mpt_get_stats(mpt_ctx* ctx, mpt_stats* out) {
    if (!ctx || !out) return MPT_ERR_INVALID_ARG;
    drain_submit(ctx);   // (a failure of a queued render stays for the next call that can report it: this one returns numbers)
    *out = ctx->stats;
    return MPT_OK;
}
extern "C" int mpt_reset_stats(mpt_ctx* ctx) {
    if (!ctx) return MPT_ERR_INVALID_ARG;
    drain_submit(ctx);
    ctx->stats = mpt_stats{};
    return MPT_OK;
}

// ---- pass machinery -------------------------------------------------------------------------------------------
static SceneDev scene_dev(const mpt_ctx* ctx) {
    SceneDev s;
    s.nodes = ctx->d_nodes;
    s.prims = ctx->d_prims;
    s.mats = ctx->d_mats;
    s.n_nodes = ctx->n_nodes;
    s.n_lds_nodes = ctx->n_lds_nodes;
    s.n_lds_prims = ctx->n_lds_prims;
    s.n_lds_mats = std::min<uint32_t>(ctx->n_mats, MPT_LDS_MATS);
    s.n_prims = ctx->n_prims;
    s.n_mats = ctx->n_mats;
    s.lds_prim_off = 2u * s.n_lds_nodes;
    s.lds_mat_off = s.lds_prim_off + 3u * s.n_lds_prims;
    return s;
}

// The closest-first kernel's view: own nodes / always list / primitives / materials / stacks in LDS; the threaded
// reference-order nodes stay in global memory (ring E and the test hook walk them from there).
static size_t ordered_views(const mpt_ctx* ctx, int point, uint32_t stack_depth, SceneDev& s, AccelDev& a) {
    const mpt_ctx::OtPoint& pt = ctx->ot_pt[point];
    const uint32_t threads = pt.threads;
    s = scene_dev(ctx);
    s.n_lds_nodes = 0;
    s.n_lds_prims = pt.lds_prims;
    a.nodes = ctx->d_acc_nodes;
    a.refleaf = ctx->d_refleaf;
    a.refbox = ctx->d_refbox;
    a.always = ctx->d_always;
    a.n_nodes = ctx->n_acc_nodes;
    a.n_lds_nodes = pt.lds_nodes;
    a.n_always = ctx->n_always;
    a.lds_always_off = 7u * a.n_lds_nodes;
    s.lds_prim_off = a.lds_always_off + 5u * a.n_always;
    s.lds_mat_off = s.lds_prim_off + 3u * s.n_lds_prims;
    const uint32_t image4 = mpt_lds_image_end_f4(s.lds_mat_off);   // (material table + the kernel's configuration block: the one definition of mpt_kernels.h)
    a.lds_stack_off = image4 * 16u;
    a.stack_depth = stack_depth;
    a.eps_abs = ctx->acc_eps_abs;
    a.cull_rel = ctx->acc_cull_rel;
    a.o_limit = ctx->tri_extent > 0.0f ? 64.0f * ctx->tri_extent : INFINITY;  // no triangles: no tree, nothing to bound
    return (size_t)a.lds_stack_off + (size_t)threads * stack_depth * 8u;
}
// The regions of k_ordered's LDS image in the order the kernel addresses them: nodes, always list, primitives, materials, configuration
// block, per-lane stacks.  None may overlap the next and the last must end inside the launch's allocation (the check that would have
// caught round 4's fault at the first launch instead of on the GPU: see mpt_lds_cfg_off_f4).
static bool ordered_layout_ok(const SceneDev& s, const AccelDev& a, uint32_t threads, size_t lds_bytes) {
    const uint64_t nodes_end = 7ull * a.n_lds_nodes, always_end = (uint64_t)a.lds_always_off + 5ull * a.n_always,
                   prims_end = (uint64_t)s.lds_prim_off + 3ull * s.n_lds_prims, mats_end = (uint64_t)s.lds_mat_off + 2ull * s.n_lds_mats,
                   cfg_begin = mpt_lds_cfg_off_f4(s.lds_mat_off), cfg_end = mpt_lds_image_end_f4(s.lds_mat_off),
                   stack_end = (uint64_t)a.lds_stack_off + (uint64_t)threads * a.stack_depth * 8u;
    return nodes_end <= a.lds_always_off && always_end <= s.lds_prim_off && prims_end <= s.lds_mat_off && mats_end <= cfg_begin &&
           cfg_end * 16u <= a.lds_stack_off && (a.lds_stack_off & 7u) == 0u && stack_end <= lds_bytes && a.stack_depth >= 2u && a.stack_depth <= MPT_OT_PARK &&
           (uint64_t)a.n_nodes < (1ull << 24) /* MPT_OT_SIGNSEL: node offsets by a 24-bit multiply */;
}
// Fragment.metal:29 + Random.h:32-35: per-pixel u32 seed of the literal RNG.  The float sin-hash is
// chaotic in the last ulp of sin(), so it is evaluated once on the host with the
// C library's sinf and uploaded; everything downstream is integer hashing on the device.
static int ensure_pixel_seeds(mpt_ctx* ctx) {
    const float* rs = ctx->u.randomSeed;
    if (ctx->d_pixel_seed && ctx->seed_W == ctx->W && ctx->seed_H == ctx->H &&
        memcmp(ctx->seed_rs, rs, 12) == 0)
        return MPT_OK;
    std::vector<uint32_t> seeds((size_t)ctx->W * ctx->H);
    const float W = ctx->u.screenSize[0], H = ctx->u.screenSize[1];
    for (uint32_t y = 0; y < ctx->H; ++y)
        for (uint32_t x = 0; x < ctx->W; ++x) {
            float uvx = ((float)x + 0.5f) / W, uvy = ((float)y + 0.5f) / H;
            float v = sinf(uvx * rs[0] + uvy * rs[1]) * rs[2];
            v = v - floorf(v);
            seeds[(size_t)y * ctx->W + x] = (uint32_t)(v * 4294967296.0f);
        }
    hipFree(ctx->d_pixel_seed);
    ctx->d_pixel_seed = nullptr;
    HIPCHK(hipMalloc(&ctx->d_pixel_seed, seeds.size() * 4));
    HIPCHK(hipMemcpy(ctx->d_pixel_seed, seeds.data(), seeds.size() * 4, hipMemcpyHostToDevice));
    memcpy(ctx->seed_rs, rs, 12);
    ctx->seed_W = ctx->W;
    ctx->seed_H = ctx->H;
    return MPT_OK;
}

// Tiles t with t % nranks == rank, in the order the pass walks them.  Strided order: local index k -> local tile
// (k * P) mod n with P ~ 0.618 n coprime to n, which spreads consecutive path-id windows over the whole image.
static int ensure_tile_order(mpt_ctx* ctx, uint32_t rank, uint32_t nranks, int mode, uint32_t& n_local) {
    const uint32_t tiles_x = (ctx->W + 7) / 8, tiles_y = (ctx->H + 7) / 8, tiles = tiles_x * tiles_y;
    n_local = tiles > rank ? (tiles - rank + nranks - 1) / nranks : 0;
    if (ctx->d_tile_xy && ctx->tile_W == ctx->W && ctx->tile_H == ctx->H && ctx->tile_rank == rank &&
        ctx->tile_nranks == nranks && ctx->tile_mode_built == mode)
        return MPT_OK;
    std::vector<uint32_t> xy(std::max<uint32_t>(n_local, 1));
    uint64_t P = 1;
    if (mode == 1 && n_local > 2) {
        P = (uint64_t)(0.6180339887 * n_local) | 1ull;
        auto gcd = [](uint64_t a, uint64_t b) {
            while (b) {
                uint64_t t = a % b;
                a = b;
                b = t;
            }
            return a;
        };
        while (gcd(P, n_local) != 1) P += 2;
    }
    for (uint32_t k = 0; k < n_local; ++k) {
        uint32_t tl = (uint32_t)(((uint64_t)k * P) % n_local);
        if (mode == 2) tl = n_local - 1u - k;  // bottom-up row-major
        const uint32_t T = tl * nranks + rank;
        xy[k] = (T % tiles_x) | ((T / tiles_x) << 16);
    }
    if (mode == 3 && n_local >= MPT_NGROUP) {
        // XCD stripes (round 5): claim range g — the workgroups with blockIdx % 8 == g, one XCD under round-robin placement, with an L2
        // of its own — owns the tiles at positions k % 8 == g of this table (range_chunk_to_path_chunk).  Row-major order gives every
        // range every 8th tile of every row: all eight L2s see the whole scene.  Here the rank's tiles are cut into eight vertical
        // stripes of equal tile count (sky, ground and geometry in each: balanced) and range g gets stripe g, top-down: the rays of one
        // XCD start in one part of the scene and its L2 holds that part of the tree.  Ranges that run dry steal from the next, as ever.
        std::vector<uint32_t> col(xy);   // the rank's tiles, column-major
        std::sort(col.begin(), col.end(), [](uint32_t a, uint32_t b) {
            const uint32_t ax = a & 0xFFFFu, bx = b & 0xFFFFu;
            return ax != bx ? ax < bx : (a >> 16) < (b >> 16);
        });
        size_t at = 0;
        for (uint32_t g = 0; g < MPT_NGROUP; ++g) {
            const uint32_t n_g = (n_local - g + MPT_NGROUP - 1u) / MPT_NGROUP;   // = range_paths' tile count of range g
            std::vector<uint32_t> stripe(col.begin() + at, col.begin() + at + n_g);
            at += n_g;
            std::sort(stripe.begin(), stripe.end(), [](uint32_t a, uint32_t b) {   // top-down, left to right inside the stripe
                const uint32_t ay = a >> 16, by = b >> 16;
                return ay != by ? ay < by : (a & 0xFFFFu) < (b & 0xFFFFu);
            });
            for (uint32_t j = 0; j < n_g; ++j) xy[(size_t)j * MPT_NGROUP + g] = stripe[j];
        }
    }
    hipFree(ctx->d_tile_xy);
    ctx->d_tile_xy = nullptr;
    HIPCHK(hipMalloc(&ctx->d_tile_xy, xy.size() * 4));
    HIPCHK(hipMemcpy(ctx->d_tile_xy, xy.data(), xy.size() * 4, hipMemcpyHostToDevice));
    ctx->tile_W = ctx->W;
    ctx->tile_H = ctx->H;
    ctx->tile_rank = rank;
    ctx->tile_nranks = nranks;
    ctx->tile_mode_built = mode;
    ctx->tile_count = n_local;
    return MPT_OK;
}

// The global ray queues exist only for the wavefront pipeline; the other two never touch them.
static int ensure_workspace(mpt_ctx* ctx, Lane& L, uint32_t slots_items, uint64_t pass_paths, bool need_queues) {
    uint32_t cap = ((slots_items + MPT_NSHARD - 1) / MPT_NSHARD + 2) * 64u;
    if (need_queues && cap > L.shard_cap) {
        free_queues(L);
        size_t n = (size_t)cap * MPT_NSHARD;
        for (int i = 0; i < 2; ++i) {
            HIPCHK(hipMalloc(&L.q[i].od, n * 16));
            HIPCHK(hipMalloc(&L.q[i].dt, n * 16));
            HIPCHK(hipMalloc(&L.q[i].tl, n * 16));
            HIPCHK(hipMalloc(&L.q[i].ia, n * 8));
        }
        L.shard_cap = cap;
    }
    if (pass_paths > L.slots_cap) {
        hipFree(L.d_slots);
        L.d_slots = nullptr;
        L.slots_cap = 0;
        HIPCHK(hipMalloc(&L.d_slots, pass_paths * 16));
        L.slots_cap = pass_paths;
    }
    return MPT_OK;
}

static int check_ready(mpt_ctx* ctx, const mpt_render_params* p) {
    if (!ctx || !p) return MPT_ERR_INVALID_ARG;
    if (!ctx->have_scene || !ctx->have_uniforms || !ctx->W) return fail(ctx, MPT_ERR_NOT_READY, "scene, uniforms or size not set");
    if ((uint32_t)ctx->u.screenSize[0] != ctx->W || (uint32_t)ctx->u.screenSize[1] != ctx->H)
        return fail(ctx, MPT_ERR_INVALID_ARG, "uniforms.screenSize does not match mpt_resize");
    if (p->rng_mode < 0 || p->rng_mode > 1 || p->bsdf_mode < 0 || p->bsdf_mode > 2 || p->max_depth < 1 ||
        p->max_depth > 31 + 1 || p->pipeline < 0 || p->pipeline > 4 || p->shard_count < 1 || p->shard_rank < 0 ||
        p->shard_rank >= p->shard_count || (uint64_t)p->sample_begin + p->sample_count > (1ull << 27))
        return fail(ctx, MPT_ERR_INVALID_ARG, "bad render params");
    return MPT_OK;
}

// Paths one pass may hold.  The global-wavefront queue packs path | bounce << 27 into one word; the wave-local
// rings and the megakernel carry full 32-bit path ids.  Per-path result slots cost 16 B each: cap a pass at 2^30
// paths (16 GiB of slots) — a whole 1920x1080x256spp render (531 M paths, 8.5 GB) is ONE pass and ONE launch.
static inline uint64_t pass_path_limit(int pipeline) {
    return pipeline == MPT_PIPE_WAVEFRONT ? (1ull << 27) : (1ull << 30);
}

static inline bool count_flag(const mpt_render_params* p) { return (p->flags & MPT_FLAG_COUNT_WORK) != 0; }

// the closest-first pipeline needs nested boxes and few spheres (mpt_upload_scene); otherwise the reference-order
// wave-local pipeline renders the same image
// MPT_PIPE_AUTO never picks the closest-first pipeline for the literal RNG: that mode exists to reproduce the reference's
// frames, arithmetic artefacts included, and only the reference's own visit order does that for every ray (include/mpt.h).
static int resolve_pipeline(const mpt_ctx* ctx, int pipeline, int rng_mode = MPT_RNG_PHILOX) {
    if (pipeline == MPT_PIPE_AUTO)
        pipeline = ctx->n_prims >= MPT_AUTO_ORDERED_PRIMS && rng_mode != MPT_RNG_LITERAL ? MPT_PIPE_ORDERED : MPT_PIPE_WAVELOCAL;
    if (pipeline == MPT_PIPE_ORDERED && !ctx->acc_ok) pipeline = MPT_PIPE_WAVELOCAL;
    return pipeline;
}

// The order in which a pass walks the rank's tiles: what MPT_TILE_ORDER says, else row-major — or, for the six-wave operating point of
// the closest-first kernel, one vertical stripe of the image per claim range = per XCD (ensure_tile_order, mode 3: measured together).
static int tile_order_of(const mpt_ctx* ctx, const mpt_render_params* p) {
    if (ctx->tile_order_forced) return ctx->tile_order_mode;
    return resolve_pipeline(ctx, p->pipeline, p->rng_mode) == MPT_PIPE_ORDERED && ordered_point(ctx, count_flag(p)) == 1 ? 3 : 0;
}

// Runs one pass of S samples/pixel over this rank's tiles; leaves the per-path results in d_slots.
static int run_pass(mpt_ctx* ctx, Lane& L, const mpt_render_params* p, uint32_t sample_begin, uint32_t S, PassParams& pp,
                    uint32_t& n_local_tiles, bool time_kernels) {
    const uint32_t tiles_x = (ctx->W + 7) / 8;
    const uint32_t nr = (uint32_t)p->shard_count, rk = (uint32_t)p->shard_rank;
    const int tile_mode = tile_order_of(ctx, p);
    {
        int trc = ensure_tile_order(ctx, rk, nr, tile_mode, n_local_tiles);
        if (trc) return trc;
    }
    const uint64_t pass_paths = (uint64_t)n_local_tiles * S * 64ull;
    if (pass_paths >= pass_path_limit(p->pipeline)) return fail(ctx, MPT_ERR_INVALID_ARG, "pass too large (internal)");
    uint32_t slots = p->slots_per_iter ? p->slots_per_iter : (16u << 20);
    uint32_t slots_items = std::max<uint32_t>(64, (slots + 63) / 64);
    if ((uint64_t)slots_items * 64 > pass_paths + 64) slots_items = (uint32_t)((pass_paths + 63) / 64);
    if (slots_items < 8) slots_items = 8;
    int rc = ensure_workspace(ctx, L, slots_items, std::max<uint64_t>(pass_paths, 64), p->pipeline == MPT_PIPE_WAVEFRONT);
    if (rc) return rc;
    if (p->rng_mode == MPT_RNG_LITERAL && (rc = ensure_pixel_seeds(ctx))) return rc;

    const int pipeline = resolve_pipeline(ctx, p->pipeline, p->rng_mode);
    pp.scene = scene_dev(ctx);
    pp.q[0] = L.q[0];
    pp.q[1] = L.q[1];
    pp.shard_cap = L.shard_cap;
    pp.desc = L.d_desc;
    pp.ctr = L.d_ctr;
    pp.slots = L.d_slots;
    pp.pixel_seed = ctx->d_pixel_seed;
    const mpt_uniforms& u = ctx->u;
    pp.cam = F3{u.cameraPosition[0], u.cameraPosition[1], u.cameraPosition[2]};
    pp.first = F3{u.firstPixelPosition[0], u.firstPixelPosition[1], u.firstPixelPosition[2]};
    pp.vu = F3{u.viewportU[0], u.viewportU[1], u.viewportU[2]};
    pp.vv = F3{u.viewportV[0], u.viewportV[1], u.viewportV[2]};
    pp.W = u.screenSize[0];
    pp.H = u.screenSize[1];
    pp.width = ctx->W;
    pp.height = ctx->H;
    pp.tiles_x = tiles_x;
    pp.S = S;
    pp.s_shift = (S & (S - 1)) == 0 ? (uint32_t)__builtin_ctz(S) : 0xFFu;
    pp.tile_xy = ctx->d_tile_xy;
    pp.sample_begin = sample_begin;
    pp.rank = rk;
    pp.nranks = nr;
    {   // path -> pixel by arithmetic when the tiles are walked in row-major order: T / tiles_x = umulhi(T, ceil(2^32 / tiles_x)) is exact
        // while T * tiles_x < 2^32 (error term T * (tiles_x - 1) / (tiles_x * 2^32) < 1 / tiles_x); otherwise the table
        const uint64_t tiles = (uint64_t)tiles_x * ((ctx->H + 7) / 8);
        pp.tile_magic = 0u;
        if (tile_mode == 0 && tiles_x >= 2u && tiles * tiles_x < (1ull << 32) && getenv("MPT_TILE_TABLE") == nullptr)
            pp.tile_magic = (uint32_t)(((1ull << 32) + tiles_x - 1u) / tiles_x);
    }
    pp.sp.rng_mode = p->rng_mode;
    pp.sp.bsdf_mode = p->bsdf_mode;
    pp.sp.max_depth = p->max_depth;
    pp.sp.seed_lo = p->seed_lo;
    pp.sp.seed_hi = p->seed_hi;
    pp.sp.primitive_count = (uint32_t)std::min<uint64_t>(u.primitiveCount, 0xFFFFFFFFull);
    uint32_t* dev_done = nullptr;
    HIPCHK(hipHostGetDevicePointer((void**)&dev_done, L.h_done, 0));
    pp.host_done = dev_done;

    if (pass_paths == 0) return MPT_OK;
    // test mode: poison the per-path result slots so that a path that is lost shows up as NaN in the image
    if (count_flag(p)) HIPCHK(hipMemsetAsync(L.d_slots, 0xFF, pass_paths * 16, L.stream));

    size_t lds = (size_t)ctx->n_lds_nodes * 32 + (size_t)ctx->n_lds_prims * 48 + MPT_LDS_EXTRA;
    int per_cu = 0;
    const bool all_lds = ctx->n_lds_nodes == ctx->n_nodes;
    // a render that overlaps others (mpt_render_async) runs the variant of the wave-local kernel that leaves room on the CU for the
    // resolve of the render before it (k_wavelocal_corun); only then may the end of one pass overlap the start of the next (below)
    const bool corun = (ctx->lane_order & 4) && !ctx->sync_render && pipeline == MPT_PIPE_WAVELOCAL && !count_flag(p);
    const int ot_point = pipeline == MPT_PIPE_ORDERED ? ordered_point(ctx, count_flag(p)) : 0;
    const void* kfun = pipeline == MPT_PIPE_MEGAKERNEL ? mega_kernel(count_flag(p), all_lds)
                       : pipeline == MPT_PIPE_WAVELOCAL ? wavelocal_kernel(count_flag(p), all_lds, corun)
                       : pipeline == MPT_PIPE_ORDERED  ? ordered_kernel(count_flag(p), ctx->ot_pt[ot_point].lds_nodes == ctx->n_acc_nodes, ot_point == 1)
                                                        : step_kernel(count_flag(p), all_lds);
    // workgroup size = the kernel's launch bound (mpt_kernels.h: 768 for the wave-local kernel; mpt_ordered.h: 256 or 768 by operating point)
    const int wg_max = pipeline == MPT_PIPE_WAVELOCAL ? MPT_WL_THREADS(all_lds) : pipeline == MPT_PIPE_ORDERED ? (int)ctx->ot_pt[ot_point].threads : 1024;
    const int wg = pipeline == MPT_PIPE_ORDERED ? wg_max : ctx->wg_size > 0 && ctx->wg_size <= wg_max ? ctx->wg_size : wg_max;
    AccelDev accel = {};
    if (pipeline == MPT_PIPE_ORDERED) {
        lds = ordered_views(ctx, ot_point, ctx->ot_stack_depth, pp.scene, accel);
        if (!ordered_layout_ok(pp.scene, accel, (uint32_t)wg, lds)) return fail(ctx, MPT_ERR_INVALID_ARG, "LDS layout of the closest-first kernel overlaps (internal)");
    }
    if (ctx->occ_fun == kfun && ctx->occ_lds == lds) {
        per_cu = ctx->occ_per_cu;
    } else {
        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfun, wg, lds));
        ctx->occ_fun = kfun;
        ctx->occ_lds = lds;
        ctx->occ_per_cu = per_cu;
    }
    if (per_cu < 1) return fail(ctx, MPT_ERR_HIP, "kernel does not fit on a CU");
    if (getenv("MPT_DEBUG_LAUNCH")) fprintf(stderr, "[mpt] pipeline %d: %d workgroups of %d per CU, %zu B of LDS each\n", pipeline, per_cu, wg, lds);
    if (ctx->wgs_per_cu > 0 && per_cu > ctx->wgs_per_cu) per_cu = ctx->wgs_per_cu;
    const int grid = ctx->prop.multiProcessorCount * per_cu;
    hipStream_t st = L.stream;
    *L.h_done = 0;
    // Trace kernels are persistent and sized for the whole chip: two of them must never be DISPATCHED side by side (each would hold
    // half of the workgroup slots for its whole life — measured: 21.7 ms per 256-spp step instead of 16.8), and which of two pending
    // launches the hardware starts is not ours to choose.  Two ways to keep them apart:
    //  * the chain: a trace kernel waits for the event behind the one enqueued before it, on whichever lane — strictly one after the
    //    other; what the second lane overlaps is the resolve of the render before, the statistics copy, the host;
    //  * the gate (renders of k_wavelocal_corun): this launch is SUBMITTED only once the trace kernel before it has announced that
    //    all its workgroups are resident (announce_resident; the host waits here, at most for the render that is running) — from then
    //    on this kernel's workgroups can only take the places the others give up as they finish: the end of one pass overlaps the
    //    start of the next, and the resolve runs in the room the variant leaves.  Without the announcement within 200 ms (a kernel
    //    that does not announce, a foreign kernel holding the chip): the chain.
    bool need_chain = (ctx->lane_order & 1) != 0;
    if (corun && ctx->last_trace_lane && ctx->last_trace_lane != &L) {
        Lane& O = *ctx->last_trace_lane;
        bool resident = false;
        if (O.announced_id != 0u) {
            const volatile uint32_t* flag = (const volatile uint32_t*)O.h_done + MPT_HOST_RESIDENT;
            const auto t0 = std::chrono::steady_clock::now();
            for (;;) {
                if (*flag == O.announced_id || hipEventQuery(O.ev_traced) == hipSuccess) {   // resident (or even finished)
                    resident = true;
                    break;
                }
                if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) break;
                std::this_thread::yield();
            }
            (resident ? ctx->gate_resident : ctx->gate_timeout)++;
        }
        need_chain = !resident;
    }
    if (need_chain && ctx->last_traced && ctx->last_traced != L.ev_traced) HIPCHK(hipStreamWaitEvent(st, ctx->last_traced, 0));
    pp.launch_id = 0u;
    if (pipeline == MPT_PIPE_WAVELOCAL || pipeline == MPT_PIPE_ORDERED) {
        if (++ctx->launch_seq == 0u) ctx->launch_seq = 1u;
        pp.launch_id = ctx->launch_seq;
    }
    L.announced_id = pp.launch_id;
    hipLaunchKernelGGL(k_begin_pass, dim3(1), dim3(64), 0, st, L.d_desc, L.d_ctr, (uint32_t)pass_paths, slots_items,
                       (volatile uint32_t*)dev_done, pipeline == MPT_PIPE_WAVEFRONT ? 0 : 1);
    if (pipeline != MPT_PIPE_WAVEFRONT) {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (time_kernels) {  // events from the context's pool; they are read after the ONE sync of mpt_render / mpt_draw
            while (L.ev_pool.size() < L.ev_used + 2) {
                hipEvent_t e;
                HIPCHK(hipEventCreate(&e));
                L.ev_pool.push_back(e);
            }
            e0 = L.ev_pool[L.ev_used++];
            e1 = L.ev_pool[L.ev_used++];
            HIPCHK(hipEventRecord(e0, st));
        }
        if (pipeline == MPT_PIPE_WAVELOCAL || pipeline == MPT_PIPE_ORDERED) {
            const size_t waves = (size_t)grid * (wg / 64);
            if (pipeline == MPT_PIPE_WAVELOCAL && waves > L.ring_waves) {
                WaveRings& r = L.ring;
                hipFree(r.base);
                r = WaveRings{};
                L.ring_waves = 0;
                const size_t n = waves * MPT_WL_LEVELS * MPT_WL_RING;
                if (n >= (1ull << 31)) return fail(ctx, MPT_ERR_INVALID_ARG, "too many waves for the ring index (internal)");
                HIPCHK(hipMalloc(&r.base, 5 * n * 16 + waves * 128 + 1024));  // five arrays + room for the MPT_DEBUG_WAVE_TIMES records
                r.n = (uint32_t)n;
                L.ring_waves = waves;
            }
            uint32_t wl_block = ctx->wl_block, wl_min = ctx->wl_min, wl_div = ctx->wl_div;
            if (wl_min == 0u) {   // (a wave's share of the pass / 16 Ki path ids, in steps of 64, between 64 and 256)
                const uint64_t share = pass_paths / std::max<size_t>(1, waves);
                wl_min = 64u * (uint32_t)std::min<uint64_t>(4, std::max<uint64_t>(1, share >> 14));
            }
            if (pipeline == MPT_PIPE_ORDERED) {
                if (waves > L.ot_ring_waves) {
                    free_ot_rings(L.ot_ring);
                    L.ot_ring_waves = 0;
                    const size_t n = waves * MPT_OT_RINGS * MPT_WL_RING;
                    if (n >= (1ull << 31)) return fail(ctx, MPT_ERR_INVALID_ARG, "too many waves for the ring index (internal)");
                    OtRings& r = L.ot_ring;
                    HIPCHK(hipMalloc(&r.base, MPT_OT_RING_ARRAYS * n * 16));
                    r.n = (uint32_t)n;
                    L.ot_ring_waves = waves;
                }
                void* args[] = {(void*)&pp, (void*)&accel, (void*)&L.ot_ring, (void*)&ctx->ot_budgets, (void*)&wl_block, (void*)&wl_min, (void*)&wl_div};
                HIPCHK(hipLaunchKernel(kfun, dim3(grid), dim3(wg), args, lds, st));
            } else {
                void* args[] = {(void*)&pp, (void*)&L.ring, (void*)&ctx->budgets, (void*)&wl_block, (void*)&wl_min, (void*)&wl_div};
                HIPCHK(hipLaunchKernel(kfun, dim3(grid), dim3(wg), args, lds, st));
            }
        } else {
            void* args[] = {(void*)&pp};
            HIPCHK(hipLaunchKernel(kfun, dim3(grid), dim3(wg), args, lds, st));
        }
        if (time_kernels) {
            HIPCHK(hipEventRecord(e1, st));
            L.pending_timed.emplace_back(e0, e1);
        }
        if (ctx->lane_order & 5) {   // (always, unless MPT_LANE_ORDER=0)
            HIPCHK(hipEventRecord(L.ev_traced, st));
            ctx->last_traced = L.ev_traced;
            ctx->last_trace_lane = &L;
        }
        ctx->stats.iterations += 1;
        return MPT_OK;
    }
    // wavefront: enqueue iterations in batches; the device publishes `done` to pinned host memory
    std::vector<hipEvent_t>& pool = L.ev_pool;
    size_t& ev_used = L.ev_used;
    auto get_event = [&]() -> hipEvent_t {
        if (ev_used == pool.size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            pool.push_back(e);
        }
        return pool[ev_used++];
    };
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timed;
    uint32_t parity = 0;
    uint64_t launched = 0;
    // Expected number of iterations: every slot of an iteration retires one ray; a pass traces about
    // 1.7 rays per path here, and the queue drains over at most max_depth further iterations.  All of them
    // are enqueued without a host round trip (an iteration that finds no work exits at once); the device
    // publishes `done` to pinned host memory and the host only tops up if the estimate was short.
    const uint64_t est = (pass_paths * 7 / 4) / ((uint64_t)slots_items * 64) + (uint64_t)p->max_depth + 2;
    const uint64_t hard_cap = est * 8 + 64;
    uint64_t batch = est;
    for (;;) {
        for (uint64_t b = 0; b < batch; ++b) {
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (time_kernels) {
                e0 = get_event();
                e1 = get_event();
                if (!e0 || !e1) return fail(ctx, MPT_ERR_HIP, "hipEventCreate failed");
                HIPCHK(hipEventRecord(e0, st));
            }
            {
                void* args[] = {(void*)&pp, (void*)&parity};
                HIPCHK(hipLaunchKernel(kfun, dim3(grid), dim3(wg), args, lds, st));
            }
            if (time_kernels) {
                HIPCHK(hipEventRecord(e1, st));
                timed.emplace_back(e0, e1);
            }
            hipLaunchKernelGGL(k_advance, dim3(1), dim3(64), 0, st, L.d_desc, L.d_ctr, (volatile uint32_t*)dev_done);
            parity ^= 1u;
            launched++;
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(st));
        uint32_t flag = *(volatile uint32_t*)L.h_done;
        if (flag & 2u) return fail(ctx, MPT_ERR_OVERFLOW, "ray queue overflow");
        if (flag & 1u) break;
        if (launched > hard_cap) return fail(ctx, MPT_ERR_HIP, "pass did not drain (internal)");
        batch = 4;
    }
    {
        PassDesc hd;
        HIPCHK(hipMemcpy(&hd, L.d_desc, sizeof hd, hipMemcpyDeviceToHost));
        // only the iterations that had work count as launches of the dominant kernel
        const uint64_t real = hd.iterations;
        for (size_t i = 0; i < timed.size() && i < real; ++i) {
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, timed[i].first, timed[i].second));
            ctx->stats.trace_kernel_ms += ms;
            if (getenv("MPT_DEBUG_ITERS")) fprintf(stderr, "[mpt] iter %zu: %.3f ms\n", i, ms);
        }
        if (time_kernels) ctx->stats.trace_launches += real;
        ctx->stats.iterations += real;
    }
    return MPT_OK;
}

// Statistics: the pass descriptor of a lane is copied to pinned memory behind the lane's last kernel
// (enqueue_stats_copy); collect_lane waits for the lane ONCE and reads the copy and the kernel-event pairs.
static int enqueue_stats_copy(mpt_ctx* ctx, Lane& L) {
    HIPCHK(hipMemcpyAsync(L.h_desc, L.d_desc, sizeof(PassDesc), hipMemcpyDeviceToHost, L.stream));
    HIPCHK(hipMemsetAsync(&L.d_desc->paths, 0, MPT_DESC_COUNTERS * sizeof(unsigned long long), L.stream));
    HIPCHK(hipMemsetAsync(&L.d_desc->overflow, 0, 4, L.stream));
    return MPT_OK;
}

static int collect_lane(mpt_ctx* ctx, Lane& L) {
    HIPCHK(hipStreamSynchronize(L.stream));
    L.in_flight = false;
    const PassDesc& hd = *L.h_desc;
    // duration of the trace kernels: what the persistent kernels stamped themselves (first workgroup's start to last wave's end on the
    // 100 MHz clock, note_wave_exit) — the event pairs of overlapping renders include the time a launch waits behind the running kernel
    float ev_ms = 0;
    for (auto& pr : L.pending_timed) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, pr.first, pr.second));
        ev_ms += ms;
        ctx->stats.trace_launches += 1;
    }
    ctx->stats.trace_kernel_ms += hd.trace_ticks != 0 && !L.pending_timed.empty() ? (double)hd.trace_ticks * 1e-5 : (double)ev_ms;
    L.pending_timed.clear();
    L.ev_used = 0;
    if (L.timed) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, L.ev0, L.ev1));
        ctx->stats.total_ms += ms;
        L.timed = false;
    }
    if (hd.overflow) return fail(ctx, MPT_ERR_OVERFLOW, "ray ring overflow");
    ctx->stats.paths += hd.paths;
    ctx->stats.rays += hd.rays;
    ctx->stats.node_visits += hd.node_visits;
    ctx->stats.aabb_hits += hd.aabb_hits;
    ctx->stats.prim_tests += hd.prim_tests;
    ctx->stats.wave_node_iters += hd.node_iters;
    ctx->stats.wave_prim_iters += hd.prim_iters;
    ctx->stats.wave_leaf_phases += hd.leaf_phases;
    ctx->stats.exact_retraces += hd.flagged;
    ctx->stats.tree_parked += hd.parked;
    return MPT_OK;
}

// Waits for every render still in flight (oldest first) and folds its statistics into ctx->stats.
static int wait_impl(mpt_ctx* ctx) {
    if (!ctx) return MPT_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    int first_rc = MPT_OK;
    for (int k = 0; k < 2; ++k) {
        Lane& L = ctx->lane[(ctx->next_lane + k) & 1];  // next_lane is the older of the two
        if (!L.in_flight) continue;
        int rc = collect_lane(ctx, L);
        if (rc && !first_rc) first_rc = rc;
    }
    return first_rc;
}

// Enqueues one render (all its passes, each followed by its resolve into the HDR sum) on the next lane and returns.
// At most two renders are in flight: the lane is collected first if it is still busy.
static int render_async_impl(mpt_ctx* ctx, const mpt_render_params* p) {
    int rc = check_ready(ctx, p);
    if (rc) return rc;
    HIPCHK(hipSetDevice(ctx->device));
    // a render with another size / sharding rebuilds the shared tile table: nothing may be in flight then
    if (ctx->d_tile_xy && (ctx->tile_W != ctx->W || ctx->tile_H != ctx->H || ctx->tile_rank != (uint32_t)p->shard_rank ||
                           ctx->tile_nranks != (uint32_t)p->shard_count || ctx->tile_mode_built != tile_order_of(ctx, p)) &&
        (rc = wait_impl(ctx)))
        return rc;
    Lane& L = ctx->lane[ctx->next_lane];
    if (L.in_flight && (rc = collect_lane(ctx, L))) return rc;
    HIPCHK(hipEventRecord(L.ev0, L.stream));
    ctx->next_lane ^= 1;
    const uint32_t tiles = ((ctx->W + 7) / 8) * ((ctx->H + 7) / 8);
    const uint32_t local_tiles = (tiles + p->shard_count - 1) / p->shard_count;
    uint32_t s_max = (uint32_t)std::max<uint64_t>(
        1, (pass_path_limit(p->pipeline) - 64) / ((uint64_t)std::max(1u, local_tiles) * 64ull));
    if (p->pipeline == MPT_PIPE_WAVEFRONT && s_max > 64) s_max = 64;
    if (s_max > 0x07FFFFFFu) s_max = 0x07FFFFFFu;
    {   // keep the samples per pass a power of two when possible (division-free path -> pixel)
        uint32_t pow2 = 1;
        while (pow2 * 2 <= s_max) pow2 *= 2;
        if (p->sample_count >= pow2) s_max = pow2;
    }
    const char* e = getenv("MPT_PASS_SPP");
    if (e && atoi(e) > 0) s_max = std::min<uint32_t>(s_max, (uint32_t)atoi(e));
    // a render that fails half way: drain what it has enqueued, forget its event pairs and give the lane back, so that
    // the next render starts from a clean lane (the HDR sum may hold some of this render's passes: the caller clears it)
    auto abandon = [&](int code) {
        // (the statistics copy that normally clears them never runs for an abandoned render)
        hipMemsetAsync(&L.d_desc->paths, 0, MPT_DESC_COUNTERS * sizeof(unsigned long long), L.stream);
        hipMemsetAsync(&L.d_desc->overflow, 0, 4, L.stream);
        hipStreamSynchronize(L.stream);
        L.pending_timed.clear();
        L.ev_used = 0;
        L.timed = false;
        L.in_flight = false;
        ctx->next_lane ^= 1;
        return code;
    };
#define HIPCHK_AB(call)                                                                     \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            set_err(ctx, std::string(#call) + ": " + hipGetErrorString(e_));                \
            return abandon(MPT_ERR_HIP);                                                    \
        }                                                                                   \
    } while (0)
    uint32_t done = 0;
    while (done < p->sample_count) {
        uint32_t S = std::min(s_max, p->sample_count - done);
        PassParams pp;
        uint32_t nlt = 0;
        rc = run_pass(ctx, L, p, p->sample_begin + done, S, pp, nlt, ctx->time_kernels);
        if (rc) return abandon(rc);
        if (nlt) {
            // sum[p] += pass total must happen in submission order on both lanes (float addition does not commute
            // bit for bit): this resolve waits for the previous one, wherever it ran
            if (ctx->last_resolved) HIPCHK_AB(hipStreamWaitEvent(L.stream, ctx->last_resolved, 0));
            // beside the other lane's trace kernel: a footprint that fits next to it on every CU (k_resolve_sum); alone: the whole chip
            const uint32_t threads = nlt * 64u, wide = (threads + 255u) / 256u;
            const uint32_t narrow = (uint32_t)ctx->prop.multiProcessorCount * (uint32_t)ctx->resolve_wgs_per_cu;
            const uint32_t rgrid = !ctx->sync_render && ctx->resolve_wgs_per_cu > 0 ? std::min(wide, narrow) : wide;
            hipLaunchKernelGGL(k_resolve_sum, dim3(rgrid), dim3(256), 0, L.stream, pp, ctx->d_sum, nlt);
            HIPCHK_AB(hipGetLastError());
            HIPCHK_AB(hipEventRecord(L.ev_resolved, L.stream));
            ctx->last_resolved = L.ev_resolved;
        }
        done += S;
        // the next pass reuses the lane's slot buffer and descriptor: in-stream order is enough, no host sync here
    }
    HIPCHK_AB(hipEventRecord(L.ev1, L.stream));
    L.timed = true;
    if ((rc = enqueue_stats_copy(ctx, L))) return abandon(rc);
    L.in_flight = true;
    return MPT_OK;
#undef HIPCHK_AB
}

// "Nothing else is in flight and nothing will be submitted behind this render": the trace kernel may be the variant that fills the
// scalar register file and the resolve may take the whole chip.  Set for the duration of mpt_render / mpt_draw; a guard, so that no
// exit path (an exception turned into a status by guarded()) leaves it set.
namespace {
struct SyncRenderScope {
    mpt_ctx* ctx;
    explicit SyncRenderScope(mpt_ctx* c) : ctx(c) { ctx->sync_render = true; }
    ~SyncRenderScope() { ctx->sync_render = false; }
    SyncRenderScope(const SyncRenderScope&) = delete;
    SyncRenderScope& operator=(const SyncRenderScope&) = delete;
};
}  // namespace
static int render_impl(mpt_ctx* ctx, const mpt_render_params* p) {
    int rc = wait_impl(ctx);
    if (rc) return rc;
    ctx->stats.trace_kernel_ms = 0;   // the synchronous call reports its own timings
    ctx->stats.trace_launches = 0;
    ctx->stats.total_ms = 0;
    ctx->next_lane = 0;  // nothing is in flight: serial renders stay on lane 0 (the second lane allocates only if used)
    {
        SyncRenderScope sync(ctx);
        rc = render_async_impl(ctx, p);
    }
    if (rc) return rc;
    return wait_impl(ctx);
}

static int draw_impl(mpt_ctx* ctx, const mpt_render_params* p) {
    int rc = check_ready(ctx, p);
    if (rc) return rc;
    if ((rc = wait_impl(ctx))) return rc;
    SyncRenderScope sync(ctx);   // (the frame protocol is synchronous: one frame at a time, collected before the call returns)
    Lane& L = ctx->lane[0];
    ctx->cur_target ^= 1;  // std::swap(_accumulationTargets[0], [1]) — Renderer.cpp:278
    PassParams pp;
    uint32_t nlt = 0;
    rc = run_pass(ctx, L, p, p->sample_begin, 1, pp, nlt, false);
    if (rc) return rc;
    if (nlt) {
        uint32_t threads = nlt * 64u;
        hipLaunchKernelGGL(k_resolve_frame, dim3((threads + 255) / 256), dim3(256), 0, L.stream, pp,
                           (const float4*)ctx->d_accum[ctx->cur_target ^ 1], ctx->d_accum[ctx->cur_target], nlt,
                           (unsigned long long)ctx->u.frameCount);
        HIPCHK(hipGetLastError());
    }
    if ((rc = enqueue_stats_copy(ctx, L))) return rc;
    return collect_lane(ctx, L);
}

// ---- unit-test entry points -----------------------------------------------------------------------------------
namespace {
struct DevBuf {  // temporary device buffer, freed on every exit path
    void* p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p) { o.p = nullptr; }
    ~DevBuf() { if (p) hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
};
}  // namespace
static int trace_rays_impl(mpt_ctx* ctx, const float* o, const float* d, uint64_t n, float* t_out,
                              int32_t* prim_out, float* normal_out, int32_t* front_out) {
    if (!ctx || !o || !d || !t_out || !prim_out || !normal_out || !front_out || n == 0 || n > (1ull << 30))
        return fail(ctx, MPT_ERR_INVALID_ARG, "bad argument");
    if (!ctx->have_scene) return fail(ctx, MPT_ERR_NOT_READY, "no scene");
    HIPCHK(hipSetDevice(ctx->device));
    DevBuf d_o, d_d, d_t, d_n, d_p, d_f;
    HIPCHK(d_o.alloc(n * 12));
    HIPCHK(d_d.alloc(n * 12));
    HIPCHK(d_t.alloc(n * 4));
    HIPCHK(d_n.alloc(n * 12));
    HIPCHK(d_p.alloc(n * 4));
    HIPCHK(d_f.alloc(n * 4));
    HIPCHK(hipMemcpy(d_o.p, o, n * 12, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_d.p, d, n * 12, hipMemcpyHostToDevice));
    SceneDev sc = scene_dev(ctx);
    hipLaunchKernelGGL(k_trace_rays, dim3((uint32_t)((n + 255) / 256)), dim3(256), (size_t)ctx->n_lds_nodes * 32 + (size_t)ctx->n_lds_prims * 48 + MPT_LDS_EXTRA,
                       ctx->stream, sc, (const float*)d_o.p, (const float*)d_d.p, (uint32_t)n, (float*)d_t.p, (int*)d_p.p, (float*)d_n.p, (int*)d_f.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(t_out, d_t.p, n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(prim_out, d_p.p, n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(normal_out, d_n.p, n * 12, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(front_out, d_f.p, n * 4, hipMemcpyDeviceToHost));
    return MPT_OK;
}

static int trace_rays_ordered_impl(mpt_ctx* ctx, const float* o, const float* d, uint64_t n, float* t_out, int32_t* prim_out,
                                   float* normal_out, int32_t* front_out, uint32_t* flags_out) {
    if (!ctx || !o || !d || !t_out || !prim_out || !normal_out || !front_out || !flags_out || n == 0 || n > (1ull << 22))
        return fail(ctx, MPT_ERR_INVALID_ARG, "bad argument (at most 2^22 rays per call)");
    if (!ctx->have_scene) return fail(ctx, MPT_ERR_NOT_READY, "no scene");
    if (!ctx->acc_ok) return fail(ctx, MPT_ERR_BAD_SCENE, "closest-first walk unavailable for this scene: " + ctx->acc_why);
    HIPCHK(hipSetDevice(ctx->device));
    DevBuf d_o, d_d, d_t, d_n, d_p, d_f, d_g;
    const uint32_t blocks = (uint32_t)((n + 255) / 256);
    HIPCHK(d_o.alloc(n * 12));
    HIPCHK(d_d.alloc(n * 12));
    HIPCHK(d_t.alloc(n * 4));
    HIPCHK(d_n.alloc(n * 12));
    HIPCHK(d_p.alloc(n * 4));
    HIPCHK(d_f.alloc(n * 4));
    HIPCHK(d_g.alloc(n * 4));
    HIPCHK(hipMemcpy(d_o.p, o, n * 12, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_d.p, d, n * 12, hipMemcpyHostToDevice));
    SceneDev sc;
    AccelDev ac;
    const size_t lds = ordered_views(ctx, 0, ctx->ot_stack_depth, sc, ac);   // (the five-wave operating point's image: workgroups of 256)
    if (!ordered_layout_ok(sc, ac, 256u, lds)) return fail(ctx, MPT_ERR_INVALID_ARG, "LDS layout of the closest-first kernel overlaps (internal)");
    hipLaunchKernelGGL(k_trace_rays_ordered, dim3(blocks), dim3(256), lds, ctx->stream, sc, ac, (const float*)d_o.p, (const float*)d_d.p,
                       (uint32_t)n, (float*)d_t.p, (int*)d_p.p, (float*)d_n.p, (int*)d_f.p, (uint32_t*)d_g.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(t_out, d_t.p, n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(prim_out, d_p.p, n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(normal_out, d_n.p, n * 12, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(front_out, d_f.p, n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(flags_out, d_g.p, n * 4, hipMemcpyDeviceToHost));
    return MPT_OK;
}

template <typename F>
static int kat_run(mpt_ctx* ctx, const void* const* in, const size_t* in_bytes, int n_in, void* const* out,
                   const size_t* out_bytes, int n_out, F launch) {
    HIPCHK(hipSetDevice(ctx->device));
    std::vector<DevBuf> bi(n_in), bo(n_out);
    std::vector<void*> di(n_in), dout(n_out);
    for (int i = 0; i < n_in; ++i) {
        HIPCHK(bi[i].alloc(in_bytes[i]));
        di[i] = bi[i].p;
        HIPCHK(hipMemcpy(di[i], in[i], in_bytes[i], hipMemcpyHostToDevice));
    }
    for (int i = 0; i < n_out; ++i) {
        HIPCHK(bo[i].alloc(out_bytes[i]));
        dout[i] = bo[i].p;
    }
    launch(di, dout);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < n_out; ++i) HIPCHK(hipMemcpy(out[i], dout[i], out_bytes[i], hipMemcpyDeviceToHost));
    return MPT_OK;
}

static int kat_pcg_impl(mpt_ctx* ctx, const uint32_t* seeds, uint64_t n, uint32_t* h, float* f) {
    if (!ctx || !seeds || !h || !f || n == 0 || n > (1u << 28)) return MPT_ERR_INVALID_ARG;
    const void* in[] = {seeds};
    size_t ib[] = {n * 4};
    void* out[] = {h, f};
    size_t ob[] = {n * 4, n * 4};
    return kat_run(ctx, in, ib, 1, out, ob, 2, [&](std::vector<void*>& di, std::vector<void*>& dout) {
        hipLaunchKernelGGL(k_kat_pcg, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const uint32_t*)di[0], (uint32_t)n, (uint32_t*)dout[0], (float*)dout[1]);
    });
}
static int kat_philox_impl(mpt_ctx* ctx, const uint32_t* c, const uint32_t* k, uint64_t n, uint32_t* o) {
    if (!ctx || !c || !k || !o || n == 0 || n > (1u << 26)) return MPT_ERR_INVALID_ARG;
    const void* in[] = {c, k};
    size_t ib[] = {n * 16, n * 8};
    void* out[] = {o};
    size_t ob[] = {n * 16};
    return kat_run(ctx, in, ib, 2, out, ob, 1, [&](std::vector<void*>& di, std::vector<void*>& dout) {
        hipLaunchKernelGGL(k_kat_philox, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const uint32_t*)di[0], (const uint32_t*)di[1], (uint32_t)n, (uint32_t*)dout[0]);
    });
}
static int kat_sincos_impl(mpt_ctx* ctx, const float* u, uint64_t n, float* s, float* c) {
    if (!ctx || !u || !s || !c || n == 0 || n > (1u << 28)) return MPT_ERR_INVALID_ARG;
    const void* in[] = {u};
    size_t ib[] = {n * 4};
    void* out[] = {s, c};
    size_t ob[] = {n * 4, n * 4};
    return kat_run(ctx, in, ib, 1, out, ob, 2, [&](std::vector<void*>& di, std::vector<void*>& dout) {
        hipLaunchKernelGGL(k_kat_sincos, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const float*)di[0], (uint32_t)n, (float*)dout[0], (float*)dout[1]);
    });
}
static int kat_rcp_impl(mpt_ctx* ctx, uint64_t* out4) {
    if (!ctx || !out4) return MPT_ERR_INVALID_ARG;
    const uint64_t zero[4] = {0, 0, 0, 0};
    const void* in[] = {zero};
    size_t ib[] = {sizeof zero};
    void* out[] = {out4};
    size_t ob[] = {sizeof zero};
    return kat_run(ctx, in, ib, 1, out, ob, 1, [&](std::vector<void*>& di, std::vector<void*>& dout) {
        (void)hipMemcpyAsync(dout[0], di[0], sizeof zero, hipMemcpyDeviceToDevice, ctx->stream);   // (an error shows in hipGetLastError of kat_run)
        hipLaunchKernelGGL(k_kat_rcp, dim3(65536), dim3(256), 0, ctx->stream, (unsigned long long*)dout[0]);
    });
}

// The binary tree of the GPU builders (mpt_lbvh.h).  Default: top-down binned SAH over the primitives (as good a tree as the host's
// binned builder).  MPT_GPU_BUILD = ploc: the clustering pass (25 % faster to build, renders 1-7 % slower); lbvh: the plain Karras tree.
static int gpu_builder() {
    const char* e = getenv("MPT_GPU_BUILD");
    if (e && strcmp(e, "lbvh") == 0) return mpt_lbvh::BUILDER_KARRAS;
    if (e && strcmp(e, "ploc") == 0) return mpt_lbvh::BUILDER_PLOC;
    return mpt_lbvh::BUILDER_SAH;
}

// Primitives per leaf of the GPU builders (MPT_LBVH_LEAF = 1..8 overrides).  Scenes that MPT_PIPE_AUTO renders with the
// reference-order kernel (fewer than MPT_AUTO_ORDERED_PRIMS primitives: the tree and a prefix of the primitives live in LDS
// there) get leaves of <= 6 — scene.xml, ms per 256 spp with 2 / 4 / 5 / 6 / 7 / 8: 20.8 / 19.3 / 19.2 / 18.6 / 18.9 / 19.1,
// against 21.3 on the reference's own tree; the closest-first kernel tests every primitive of a leaf it enters and wants them
// small (bunny x20: 12.5 Grays/s with 4, 13.2 with 2).
static int gpu_leaf_max(uint64_t n_prims) {
    if (const char* lm = getenv("MPT_LBVH_LEAF")) return std::min(std::max(atoi(lm), 1), (int)MPT_LBVH_LEAF_MAX);
    return n_prims < MPT_AUTO_ORDERED_PRIMS ? 6 : 2;
}

// ---- build -> render without the host (mpt_devbuild.h) -------------------------------------------------------------------------
static int build_and_upload_impl(mpt_ctx* ctx, const float* prims, const float* mats, uint64_t n_prims, double* device_ms_out) {
    if (!ctx) return MPT_ERR_INVALID_ARG;
    if (!prims || !mats || n_prims == 0) return fail(ctx, MPT_ERR_INVALID_ARG, "null or empty primitive / material array");
    if (n_prims >= (1ull << 27)) return fail(ctx, MPT_ERR_BAD_SCENE, "scene too large for the 27-bit leaf encoding");
    {
        int wrc = wait_impl(ctx);  // renders in flight still read the old scene
        if (wrc) return wrc;
    }
    HIPCHK(hipSetDevice(ctx->device));
    const uint32_t n = (uint32_t)n_prims;
    // (whether the always list can be used decides the own boxes of the leaves that hold spheres.  The default builder counts the spheres on
    //  the device, in a kernel it runs anyway; a pass of the host over the 48 MB of 1 M primitives was 0.4 ms of the call)
    uint32_t n_spheres = 0xFFFFFFFFu;
    if (!(gpu_builder() == mpt_lbvh::BUILDER_SAH && n > 2)) {
        n_spheres = 0;
        for (uint32_t i = 0; i < n; ++i) n_spheres += (int)prims[12 * (size_t)i + 3] != 1 ? 1u : 0u;
    }
    // the staging buffer of the inputs is kept between builds (<= 1 GiB): two hipMallocs and two hipFrees of 80 MB were 0.3 ms of the call
    const size_t in_bytes = (size_t)n * 80;
    if (ctx->in_block_bytes < in_bytes) {
        hipFree(ctx->in_block);
        ctx->in_block = nullptr;
        ctx->in_block_bytes = 0;
        HIPCHK(hipMalloc(&ctx->in_block, in_bytes));
        ctx->in_block_bytes = in_bytes;
    }
    struct InBuf {
        void* p;
    } d_p{ctx->in_block}, d_m{(char*)ctx->in_block + (size_t)n * 48};
    struct InRelease {   // (a buffer too large to keep goes when the call ends)
        mpt_ctx* c;
        ~InRelease() {
            if (c->in_block_bytes > ((size_t)1 << 30)) {
                hipFree(c->in_block);
                c->in_block = nullptr;
                c->in_block_bytes = 0;
            }
        }
    } in_release{ctx};
    HIPCHK(hipMemcpyAsync(d_p.p, prims, (size_t)n * 48, hipMemcpyHostToDevice, ctx->stream));
    const int leaf_max = gpu_leaf_max(n);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIPCHK(hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) {
        hipEventDestroy(e0);
        return fail(ctx, MPT_ERR_HIP, "hipEventCreate failed");
    }
    hipEventRecord(e0, ctx->stream);
    mpt_devbuild::Built b;
    // (the materials are copied by the build itself, on its second stream, beside the tree build: mpt_devbuild.h build_pass)
    hipError_t e = mpt_devbuild::build(ctx->stream, (float4*)d_p.p, (float4*)d_m.p, mats, n, leaf_max, gpu_builder(), n_spheres, b, &ctx->build_pool, &ctx->spare_block,
                                       &ctx->spare_block_bytes);
    if (e == hipSuccess) e = hipEventRecord(e1, ctx->stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.0f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    if (e != hipSuccess) {
        b.release();
        return fail(ctx, MPT_ERR_HIP, std::string("GPU BVH build: ") + hipGetErrorString(e));
    }
    if (n_spheres == 0xFFFFFFFFu) n_spheres = b.n_spheres;
    if (b.n_spheres != n_spheres) {
        b.release();
        return fail(ctx, MPT_ERR_HIP, "GPU BVH build: sphere count mismatch (internal)");
    }
    free_scene_buffers(ctx, true);   // (the old scene's block becomes the next build's)
    ctx->scene_block = b.block;
    ctx->scene_block_bytes = b.block_bytes;
    ctx->d_nodes = b.nodes;
    ctx->d_prims = b.prims;
    ctx->d_mats = b.mats;
    ctx->d_acc_nodes = b.acc_nodes;
    ctx->n_acc_nodes = b.n_acc_nodes;
    ctx->d_refleaf = b.refleaf;
    ctx->d_refbox = b.refbox;
    ctx->d_always = b.always;
    ctx->d_ref_bvh = b.ref_bvh;
    ctx->d_ref_idx = b.ref_idx;
    ctx->n_ref_nodes = b.n_nodes;
    ctx->built_leaf_max = (uint32_t)leaf_max;
    ctx->n_nodes = b.n_nodes;
    ctx->n_prims = b.n_prims;
    ctx->n_mats = b.n_mats;
    ctx->n_acc_nodes = b.n_acc_nodes;
    ctx->n_always = b.n_always;
    ctx->n_ref_leaves = b.n_ref_leaves;
    ctx->acc_depth = b.acc_depth;
    ctx->tri_extent = b.tri_extent;
    ctx->acc_eps_abs = b.tri_extent * 3.814697265625e-06f;  // 2^-18 of the largest triangle coordinate
    ctx->acc_ok = n_spheres <= MPT_ACCEL_MAX_ALWAYS;        // (parent boxes are unions of child boxes: nested by construction)
    ctx->acc_why = ctx->acc_ok ? "" : "more than 16 spheres";
    size_lds_images(ctx);
    ctx->have_scene = true;
    if (device_ms_out) *device_ms_out = ms;
    return MPT_OK;
}

static int download_bvh_impl(mpt_ctx* ctx, float* bvh_out, uint64_t cap_nodes, uint64_t* n_nodes_out, int32_t* prim_idx_out) {
    if (!ctx || !bvh_out || !n_nodes_out || !prim_idx_out) return fail(ctx, MPT_ERR_INVALID_ARG, "bad argument");
    if (!ctx->have_scene || !ctx->d_ref_bvh) return fail(ctx, MPT_ERR_NOT_READY, "the scene was not built by mpt_build_and_upload");
    if (cap_nodes < ctx->n_ref_nodes) return fail(ctx, MPT_ERR_INVALID_ARG, "bvh_out too small");
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemcpy(bvh_out, ctx->d_ref_bvh, (size_t)ctx->n_ref_nodes * 32, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(prim_idx_out, ctx->d_ref_idx, (size_t)ctx->n_prims * 4, hipMemcpyDeviceToHost));
    *n_nodes_out = ctx->n_ref_nodes;
    return MPT_OK;
}

// ---- mpt_render_async's submit thread (struct Submitter) -------------------------------------------------------------------------
static void submit_thread_main(mpt_ctx* ctx) {
    Submitter& S = *ctx->sub;
    std::unique_lock<std::mutex> lk(S.mu);
    for (;;) {
        S.cv_work.wait(lk, [&] { return S.stop || !S.q.empty(); });
        if (S.q.empty()) return;   // (stop, and nothing left to submit)
        const mpt_render_params p = S.q.front();
        S.q.pop_front();
        S.busy = true;
        lk.unlock();
        int rc = MPT_ERR_HIP;
        try {
            rc = render_async_impl(ctx, &p);
            ctx->async_jobs++;
        } catch (const std::exception& e) {
            try {
                set_err(ctx, std::string("host exception: ") + e.what());
            } catch (...) {
            }
        } catch (...) {
        }
        lk.lock();
        if (rc != MPT_OK && S.rc == MPT_OK) {
            S.rc = rc;
            S.err = ctx->err;
        }
        S.busy = false;
        S.cv_idle.notify_all();
    }
}
// Everything queued has been submitted (and the thread is idle): the context is the caller's alone again.  Returns the first failure of
// a queued render, once.  Called by every entry point except mpt_render_async.
static int drain_submit(mpt_ctx* ctx) {
    if (!ctx || !ctx->sub) return MPT_OK;
    Submitter& S = *ctx->sub;
    if (std::this_thread::get_id() == S.th.get_id()) return MPT_OK;   // (the submit thread itself, inside render_async_impl)
    std::unique_lock<std::mutex> lk(S.mu);
    S.cv_idle.wait(lk, [&] { return S.q.empty() && !S.busy; });
    const int rc = S.rc;
    if (rc != MPT_OK) {
        set_err(ctx, S.err);
        S.rc = MPT_OK;
        S.err.clear();
    }
    return rc;
}
static void stop_submit(mpt_ctx* ctx) {
    if (!ctx || !ctx->sub) return;
    {
        std::lock_guard<std::mutex> lk(ctx->sub->mu);
        ctx->sub->stop = true;
    }
    ctx->sub->cv_work.notify_all();
    if (ctx->sub->th.joinable()) ctx->sub->th.join();   // (queued renders are submitted first: the thread leaves on an empty queue)
    ctx->sub.reset();
}
static int enqueue_render(mpt_ctx* ctx, const mpt_render_params* p) {
    const auto t0 = std::chrono::steady_clock::now();
    int rc = check_ready(ctx, p);   // argument and state errors are the caller's, at once (the state they depend on only changes in calls that drain)
    if (rc) return rc;
    if (!ctx->sub) {
        ctx->sub.reset(new Submitter());
        ctx->sub->th = std::thread(submit_thread_main, ctx);
    }
    Submitter& S = *ctx->sub;
    {
        std::unique_lock<std::mutex> lk(S.mu);
        S.cv_idle.wait(lk, [&] { return S.q.size() < MPT_ASYNC_QUEUE_MAX; });
        S.q.push_back(*p);
    }
    S.cv_work.notify_one();
    const uint64_t us = (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
    ctx->async_call_us_max = std::max(ctx->async_call_us_max, us);
    return MPT_OK;
}

// ---- exception barrier: nothing thrown by the host-side containers may cross the C ABI (include/mpt.h) --------
// (and the join point of the submit thread: every guarded entry point first waits until the renders queued by mpt_render_async have
//  been submitted — DRAIN = false for mpt_render_async itself)
template <bool DRAIN = true, class F>
static int guarded(mpt_ctx* ctx, F&& body) noexcept {
    try {
        if (DRAIN) {
            const int d = drain_submit(ctx);
            if (d) return d;
        }
        return body();
    } catch (const std::exception& e) {
        if (ctx) {
            try {
                set_err(ctx, std::string("host exception: ") + e.what());
            } catch (...) {
            }
        }
        return MPT_ERR_HIP;
    } catch (...) {
        return MPT_ERR_HIP;
    }
}

extern "C" int mpt_create(int device_ordinal, mpt_ctx** out) {
    return guarded(nullptr, [&] { return create_impl(device_ordinal, out); });
}

extern "C" int mpt_upload_scene(mpt_ctx* ctx, const float* bvh, uint64_t n_nodes, const float* prims, const float* mats, const int32_t* prim_idx, uint64_t n_prims) {
    return guarded(ctx, [&] { return upload_scene_impl(ctx, bvh, n_nodes, prims, mats, prim_idx, n_prims); });
}

extern "C" int mpt_set_uniforms(mpt_ctx* ctx, const mpt_uniforms* u) {
    return guarded(ctx, [&] { return set_uniforms_impl(ctx, u); });
}

extern "C" int mpt_resize(mpt_ctx* ctx, uint32_t width, uint32_t height) {
    return guarded(ctx, [&] { return resize_impl(ctx, width, height); });
}

extern "C" int mpt_clear_sum(mpt_ctx* ctx) {
    return guarded(ctx, [&] { return clear_sum_impl(ctx); });
}

extern "C" int mpt_read_frame(mpt_ctx* ctx, float* out) {
    return guarded(ctx, [&] { return read_frame_impl(ctx, out); });
}

extern "C" int mpt_read_sum(mpt_ctx* ctx, float* out) {
    return guarded(ctx, [&] { return read_sum_impl(ctx, out); });
}

// The inverse: the HDR sum of an earlier run goes back in (checkpoint / resume of the accumulation: with the sample index the
// caller continues from, the resumed render is bit-identical to an uninterrupted one — sum[p] += sample_s in sample order either way).
static int write_sum_impl(mpt_ctx* ctx, const float* in) {
    if (!ctx || !in) return MPT_ERR_INVALID_ARG;
    if (!ctx->d_sum) return fail(ctx, MPT_ERR_NOT_READY, "mpt_resize not called");
    int wrc = wait_impl(ctx);
    if (wrc) return wrc;
    HIPCHK(hipMemcpyAsync(ctx->d_sum, in, (size_t)ctx->W * ctx->H * 16, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipEventRecord(ctx->ev_sum_op, ctx->stream));   // the next resolve (on either lane) is ordered behind the upload
    ctx->last_resolved = ctx->ev_sum_op;
    HIPCHK(hipStreamSynchronize(ctx->stream));             // (the host array may be freed on return)
    return MPT_OK;
}
extern "C" int mpt_write_sum(mpt_ctx* ctx, const float* rgba_host) {
    return guarded(ctx, [&] { return write_sum_impl(ctx, rgba_host); });
}

extern "C" int mpt_render(mpt_ctx* ctx, const mpt_render_params* p) {
    return guarded(ctx, [&] { return render_impl(ctx, p); });
}
extern "C" int mpt_render_async(mpt_ctx* ctx, const mpt_render_params* p) {
    return guarded<false>(ctx, [&] { return enqueue_render(ctx, p); });
}
extern "C" int mpt_async_info(mpt_ctx* ctx, uint64_t out[4]) {
    if (!ctx || !out) return MPT_ERR_INVALID_ARG;
    return guarded(ctx, [&] {
        const uint64_t v[4] = {ctx->async_jobs, ctx->gate_resident, ctx->gate_timeout, ctx->async_call_us_max};
        memcpy(out, v, sizeof v);
        return (int)MPT_OK;
    });
}
extern "C" int mpt_wait(mpt_ctx* ctx) {
    return guarded(ctx, [&] { return wait_impl(ctx); });
}

extern "C" int mpt_draw(mpt_ctx* ctx, const mpt_render_params* p) {
    return guarded(ctx, [&] { return draw_impl(ctx, p); });
}

extern "C" int mpt_trace_rays(mpt_ctx* ctx, const float* o, const float* d, uint64_t n, float* t_out, int32_t* prim_out, float* normal_out, int32_t* front_out) {
    return guarded(ctx, [&] { return trace_rays_impl(ctx, o, d, n, t_out, prim_out, normal_out, front_out); });
}

extern "C" int mpt_trace_rays_ordered(mpt_ctx* ctx, const float* o, const float* d, uint64_t n, float* t_out, int32_t* prim_out, float* normal_out, int32_t* front_out, uint32_t* flags_out) {
    return guarded(ctx, [&] { return trace_rays_ordered_impl(ctx, o, d, n, t_out, prim_out, normal_out, front_out, flags_out); });
}

extern "C" int mpt_accel_info(mpt_ctx* ctx, uint64_t out[8]) {
    if (!ctx || !out) return MPT_ERR_INVALID_ARG;
    if (!ctx->have_scene) return fail(ctx, MPT_ERR_NOT_READY, "no scene");
    const uint64_t v[8] = {ctx->acc_ok ? 1u : 0u, ctx->n_acc_nodes, ctx->acc_depth, ctx->ot_pt[ordered_point(ctx, false)].lds_nodes, ctx->n_always, ctx->n_ref_leaves, ctx->ot_pt[ordered_point(ctx, false)].lds_prims,
                           (uint64_t)resolve_pipeline(ctx, MPT_PIPE_AUTO)};
    memcpy(out, v, sizeof v);
    return MPT_OK;
}

extern "C" int mpt_gpu_leaf_max(uint64_t n_prims) { return gpu_leaf_max(n_prims); }

// Position-sensitive 64-bit digest of a device array of 32-bit words: sum over i of splitmix64(i << 32 | word[i]) (a commutative sum, so
// the order in which the waves add is free).  What the tests compare two builds of a scene by, array by array (mpt_scene_digest).
__global__ void k_digest(const uint32_t* w, uint64_t n_words, unsigned long long* out) {
    unsigned long long acc = 0ull;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * blockDim.x) {
        unsigned long long z = (i << 32 | w[i]) + 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        acc += z ^ (z >> 31);
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63u) == 0) atomicAdd(out, acc);
}
extern "C" int mpt_scene_digest(mpt_ctx* ctx, uint64_t out[16]) {
    return guarded(ctx, [&]() -> int {
        if (!ctx || !out) return MPT_ERR_INVALID_ARG;
        if (!ctx->have_scene) return fail(ctx, MPT_ERR_NOT_READY, "no scene");
        HIPCHK(hipSetDevice(ctx->device));
        const bool built = ctx->d_ref_bvh != nullptr;
        struct Arr { const void* p; uint64_t words; } a[9] = {
            {ctx->d_nodes, (uint64_t)ctx->n_nodes * 8}, {ctx->d_prims, (uint64_t)ctx->n_prims * 12}, {ctx->d_mats, (uint64_t)ctx->n_mats * 8},
            {ctx->d_acc_nodes, (uint64_t)ctx->n_acc_nodes * MPT_OT_NODE_STRIDE * 4}, {ctx->d_refleaf, (uint64_t)ctx->n_ref_leaves * 8},
            {ctx->d_refbox, ctx->d_refbox ? (uint64_t)ctx->n_prims * 8 : 0u}, {ctx->d_always, (uint64_t)ctx->n_always * 20},
            {ctx->d_ref_bvh, built ? (uint64_t)ctx->n_ref_nodes * 8 : 0u}, {ctx->d_ref_idx, built ? (uint64_t)ctx->n_prims : 0u}};
        DevBuf d;
        HIPCHK(d.alloc(16 * 8));
        HIPCHK(hipMemsetAsync(d.p, 0, 16 * 8, ctx->stream));
        for (int k = 0; k < 9; ++k)
            if (a[k].p && a[k].words) {
                const uint32_t g = (uint32_t)std::min<uint64_t>((a[k].words + 255) / 256, 2048);
                hipLaunchKernelGGL(k_digest, dim3(g), dim3(256), 0, ctx->stream, (const uint32_t*)a[k].p, a[k].words, (unsigned long long*)d.p + k);
            }
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(out, d.p, 16 * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        const uint64_t counts[7] = {ctx->n_nodes, ctx->n_prims, ctx->n_mats, ctx->n_acc_nodes, ctx->n_ref_leaves, ctx->n_always, ctx->acc_depth};
        for (int k = 0; k < 7; ++k) out[9 + k] = counts[k];
        return MPT_OK;
    });
}

extern "C" int mpt_build_info(mpt_ctx* ctx, uint64_t out[8]) {
    if (!ctx || !out) return MPT_ERR_INVALID_ARG;
    if (!ctx->have_scene) return fail(ctx, MPT_ERR_NOT_READY, "no scene");
    const bool built = ctx->d_ref_bvh != nullptr;
    const uint64_t v[8] = {built ? ctx->n_prims : 0u, built ? ctx->n_ref_nodes : 0u, built ? ctx->built_leaf_max : 0u, MPT_AUTO_ORDERED_PRIMS,
                           ctx->n_prims, ctx->n_nodes, ctx->n_mats, 0u /* (was: nodes left unquantised by round 4's 64-byte node experiment) */};
    memcpy(out, v, sizeof v);
    return MPT_OK;
}

extern "C" int mpt_build_bvh(mpt_ctx* ctx, const float* prims, uint64_t n_prims, float* bvh_out, uint64_t bvh_capacity_nodes,
                             uint64_t* n_nodes_out, int32_t* prim_idx_out, double* device_ms_out) {
    return guarded(ctx, [&]() -> int {
        if (!ctx || !prims || !bvh_out || !n_nodes_out || !prim_idx_out || n_prims == 0 || n_prims >= (1ull << 27))
            return fail(ctx, MPT_ERR_INVALID_ARG, "bad argument");
        if (bvh_capacity_nodes < 2 * n_prims - 1) return fail(ctx, MPT_ERR_INVALID_ARG, "bvh_out must hold 2 * n_prims - 1 nodes");
        HIPCHK(hipSetDevice(ctx->device));
        float ms = 0.0f;
        const int leaf_max = gpu_leaf_max(n_prims);
        hipError_t e = mpt_lbvh::build(ctx->stream, prims, (uint32_t)n_prims, leaf_max, gpu_builder(), bvh_out, n_nodes_out, prim_idx_out, &ms, &ctx->build_pool);
        if (e != hipSuccess) return fail(ctx, MPT_ERR_HIP, std::string("GPU BVH build: ") + hipGetErrorString(e));
        if (device_ms_out) *device_ms_out = ms;
        return MPT_OK;
    });
}

extern "C" int mpt_build_and_upload(mpt_ctx* ctx, const float* prims, const float* mats, uint64_t n_prims, double* device_ms_out) {
    return guarded(ctx, [&] { return build_and_upload_impl(ctx, prims, mats, n_prims, device_ms_out); });
}

extern "C" int mpt_download_bvh(mpt_ctx* ctx, float* bvh_out, uint64_t bvh_capacity_nodes, uint64_t* n_nodes_out, int32_t* prim_idx_out) {
    return guarded(ctx, [&] { return download_bvh_impl(ctx, bvh_out, bvh_capacity_nodes, n_nodes_out, prim_idx_out); });
}

extern "C" int mpt_kat_pcg(mpt_ctx* ctx, const uint32_t* seeds, uint64_t n, uint32_t* h, float* f) {
    return guarded(ctx, [&] { return kat_pcg_impl(ctx, seeds, n, h, f); });
}

extern "C" int mpt_kat_philox(mpt_ctx* ctx, const uint32_t* c, const uint32_t* k, uint64_t n, uint32_t* o) {
    return guarded(ctx, [&] { return kat_philox_impl(ctx, c, k, n, o); });
}

extern "C" int mpt_kat_sincos(mpt_ctx* ctx, const float* u, uint64_t n, float* s, float* c) {
    return guarded(ctx, [&] { return kat_sincos_impl(ctx, u, n, s, c); });
}

extern "C" int mpt_kat_rcp(mpt_ctx* ctx, uint64_t* out4) {
    return guarded(ctx, [&] { return kat_rcp_impl(ctx, out4); });
}

// ---- multi-GPU: RCCL reduce of the HDR sum (include/mpt.h) ----------------------------------------------------------
#include <dlfcn.h>
#include <rccl/rccl.h>
namespace {
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;
static const char* rccl_load() {  // nullptr = ok, else what failed
    if (g_rccl.lib) return nullptr;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return "librccl.so not found";
    RcclApi a;
    a.lib = h;
#define MPT_SYM(field, name)                                 \
    *(void**)(&a.field) = dlsym(h, name);                    \
    if (!a.field) {                                          \
        dlclose(h);                                          \
        return "librccl.so lacks " name;                     \
    }
    MPT_SYM(GetUniqueId, "ncclGetUniqueId")
    MPT_SYM(CommInitAll, "ncclCommInitAll")
    MPT_SYM(CommInitRank, "ncclCommInitRank")
    MPT_SYM(CommDestroy, "ncclCommDestroy")
    MPT_SYM(CommAbort, "ncclCommAbort")
    MPT_SYM(Reduce, "ncclReduce")
    MPT_SYM(GroupStart, "ncclGroupStart")
    MPT_SYM(GroupEnd, "ncclGroupEnd")
    MPT_SYM(GetErrorString, "ncclGetErrorString")
#undef MPT_SYM
    g_rccl = a;
    return nullptr;
}
}  // namespace
static_assert(sizeof(ncclUniqueId) == MPT_COMM_ID_BYTES, "MPT_COMM_ID_BYTES must match ncclUniqueId");

struct mpt_comm {
    std::vector<mpt_ctx*> ctxs;      // local contexts (all N in one process, or this rank's one)
    std::vector<ncclComm_t> comms;   // one per local context; empty when nranks == 1
    int nranks = 1, first_rank = 0;  // global size; global rank of ctxs[0]
    bool aborted = false;            // a local failure aborted the communicators: nothing more can be reduced
    std::string err;
};

extern "C" const char* mpt_comm_last_error(const mpt_comm* c) { return c ? c->err.c_str() : "null communicator"; }

extern "C" int mpt_comm_unique_id(void* id_out) {
    if (!id_out) return MPT_ERR_INVALID_ARG;
    if (rccl_load()) return MPT_ERR_HIP;
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) return MPT_ERR_HIP;
    memcpy(id_out, &id, sizeof id);
    return MPT_OK;
}

extern "C" int mpt_comm_create_all(mpt_ctx* const* ctxs, int n, mpt_comm** out) {
    if (!out) return MPT_ERR_INVALID_ARG;
    *out = nullptr;
    if (!ctxs || n < 1) return MPT_ERR_INVALID_ARG;
    for (int i = 0; i < n; ++i)
        if (!ctxs[i]) return MPT_ERR_INVALID_ARG;
    try {
        std::unique_ptr<mpt_comm> c(new mpt_comm());
        c->ctxs.assign(ctxs, ctxs + n);
        c->nranks = n;
        if (n > 1) {
            if (const char* why = rccl_load()) return fail(ctxs[0], MPT_ERR_HIP, why);
            std::vector<int> devs(n);
            for (int i = 0; i < n; ++i) devs[i] = ctxs[i]->device;
            c->comms.resize(n);
            ncclResult_t r = g_rccl.CommInitAll(c->comms.data(), n, devs.data());
            if (r != ncclSuccess) return fail(ctxs[0], MPT_ERR_HIP, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r));
        }
        *out = c.release();
        return MPT_OK;
    } catch (...) {
        return MPT_ERR_HIP;
    }
}

extern "C" int mpt_comm_create_rank(mpt_ctx* ctx, int rank, int nranks, const void* id, mpt_comm** out) {
    if (!out) return MPT_ERR_INVALID_ARG;
    *out = nullptr;
    if (!ctx || nranks < 1 || rank < 0 || rank >= nranks || (nranks > 1 && !id)) return MPT_ERR_INVALID_ARG;
    try {
        std::unique_ptr<mpt_comm> c(new mpt_comm());
        c->ctxs.push_back(ctx);
        c->nranks = nranks;
        c->first_rank = rank;
        if (nranks > 1) {
            if (const char* why = rccl_load()) return fail(ctx, MPT_ERR_HIP, why);
            if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, MPT_ERR_HIP, "hipSetDevice failed");
            ncclUniqueId uid;
            memcpy(&uid, id, sizeof uid);
            c->comms.resize(1);
            ncclResult_t r = g_rccl.CommInitRank(&c->comms[0], nranks, uid, rank);
            if (r != ncclSuccess) return fail(ctx, MPT_ERR_HIP, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r));
        }
        *out = c.release();
        return MPT_OK;
    } catch (...) {
        return MPT_ERR_HIP;
    }
}

// A rank that cannot enter the collective must not leave its peers waiting in it: the local communicators are aborted
// (ncclCommAbort), which makes the peers' ncclReduce fail instead of hanging on their GPUs.  The job is over then — the
// communicator cannot be used again (mpt_reduce_sum returns MPT_ERR_NOT_READY on it), the caller tears down and restarts.
static void comm_abort(mpt_comm* c) {
    for (ncclComm_t& k : c->comms) {
        if (k && g_rccl.CommAbort) g_rccl.CommAbort(k);
        k = nullptr;
    }
    c->aborted = true;
}

extern "C" int mpt_reduce_sum(mpt_comm* c, int root) {
    if (!c || root < 0 || root >= c->nranks) return MPT_ERR_INVALID_ARG;
    if (c->aborted) {
        c->err = "communicator was aborted by an earlier failure";
        return MPT_ERR_NOT_READY;
    }
    // every local context: collect the renders in flight (their resolves have then updated the HDR sum)
    int local_rc = MPT_OK;
    for (mpt_ctx* ctx : c->ctxs) {
        int rc = drain_submit(ctx);
        if (!rc) rc = wait_impl(ctx);
        if (rc) {
            c->err = ctx->err;
            local_rc = rc;
            break;
        }
        if (!ctx->d_sum || ctx->W != c->ctxs[0]->W || ctx->H != c->ctxs[0]->H) {
            c->err = "contexts of a communicator must be sized alike (mpt_resize)";
            local_rc = MPT_ERR_NOT_READY;
            break;
        }
    }
    if (local_rc) {
        if (c->nranks > 1) comm_abort(c);
        return local_rc;
    }
    if (c->nranks == 1) return MPT_OK;  // one GPU holds the whole image already
    const size_t count = (size_t)c->ctxs[0]->W * c->ctxs[0]->H * 4;
    ncclResult_t r = g_rccl.GroupStart();
    if (r == ncclSuccess) {   // (GroupEnd always follows a successful GroupStart)
        for (size_t i = 0; i < c->ctxs.size() && r == ncclSuccess; ++i) {
            mpt_ctx* ctx = c->ctxs[i];
            if (hipSetDevice(ctx->device) != hipSuccess) r = ncclUnhandledCudaError;
            else r = g_rccl.Reduce(ctx->d_sum, ctx->d_sum, count, ncclFloat32, ncclSum, root, c->comms[i], ctx->stream);
        }
        ncclResult_t e = g_rccl.GroupEnd();
        if (r == ncclSuccess) r = e;
    }
    if (r != ncclSuccess) {
        c->err = std::string("ncclReduce: ") + g_rccl.GetErrorString(r);
        comm_abort(c);
        return MPT_ERR_HIP;
    }
    for (mpt_ctx* ctx : c->ctxs) {
        if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
            c->err = "stream synchronisation after ncclReduce failed";
            comm_abort(c);
            return MPT_ERR_HIP;
        }
    }
    return MPT_OK;
}

extern "C" int mpt_comm_destroy(mpt_comm* c) {
    if (!c) return MPT_ERR_INVALID_ARG;
    for (ncclComm_t k : c->comms)
        if (k && g_rccl.CommDestroy) g_rccl.CommDestroy(k);
    delete c;
    return MPT_OK;
}

#ifdef MPT_OT_TIMES
extern "C" int mpt_debug_ot_times(unsigned long long* out40, int reset) {   // [0,24) g_ot_times, [24,40) g_ot_walk
    hipError_t e = hipMemcpyFromSymbol(out40, HIP_SYMBOL(g_ot_times), 24 * 8);
    if (e == hipSuccess) e = hipMemcpyFromSymbol(out40 + 24, HIP_SYMBOL(g_ot_walk), 16 * 8);
    if (e == hipSuccess && reset) {
        unsigned long long z[24] = {};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_ot_times), z, sizeof z);
        if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_ot_walk), z, 16 * 8);
    }
    return (int)e;
}
#endif

#ifdef MPT_CLOCK_STAMP
extern "C" int mpt_debug_clock(unsigned long long* out2, int reset) {   // sums over waves: shader cycles, 100 MHz ticks
    hipError_t e = hipMemcpyFromSymbol(out2, HIP_SYMBOL(g_clock), 16);
    if (e == hipSuccess && reset) {
        unsigned long long z[2] = {};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_clock), z, 16);
    }
    return (int)e;
}
#endif

#ifdef MPT_DEBUG_WAVE_TIMES
static mpt_ctx* g_dbg_ctx = nullptr;
extern "C" void mpt_debug_bind(mpt_ctx* ctx) { g_dbg_ctx = ctx; }
extern "C" int mpt_debug_reset() {   // zero the diagnostics block behind lane 0's rings
    mpt_ctx* ctx = g_dbg_ctx;
    const Lane& L = ctx->lane[0];
    if (!L.ring.base) return -1;
    const size_t off = L.ring_waves * MPT_WL_LEVELS * MPT_WL_RING;
    return (int)hipMemset((char*)L.ring.tv() + off * 16, 0, L.ring_waves * 128 + 1024);
}
extern "C" int mpt_debug_wave_times(unsigned long long* out, int n) {
    mpt_ctx* ctx = g_dbg_ctx;
    const Lane& L = ctx->lane[0];
    const size_t off = L.ring_waves * MPT_WL_LEVELS * MPT_WL_RING;
    return (int)hipMemcpy(out, (const char*)L.ring.tv() + off * 16, (size_t)n * 128 + 1024, hipMemcpyDeviceToHost);
}
#endif
