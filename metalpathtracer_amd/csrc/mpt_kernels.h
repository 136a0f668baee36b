// mpt_kernels.h — the HIP kernels of the hot path (gfx950 only) and the device-side structures they share with the
// host code in mpt_hip.hip.  Building blocks (RNG, closest hit, shading) are in mpt_device.h.
//
//   k_wavelocal      default pipeline: persistent waves, wave-private work-sorted ray rings (MPT_PIPE_WAVELOCAL)
//   k_step/k_advance global wavefront: device-wide SoA queues, one launch per bounce generation (MPT_PIPE_WAVEFRONT)
//   k_megakernel     one lane per path (MPT_PIPE_MEGAKERNEL)
//   k_resolve_*      per-path result slots -> HDR sum / the reference's running-mean frame
//   k_trace_rays, k_kat_*  unit-test hooks
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mpt_device.h"

// Minimum waves per SIMD the two secondary trace kernels (k_step, k_megakernel) are compiled for (2nd __launch_bounds__
// argument): 8 waves/SIMD (<= 64 VGPRs, 2 x 1024-thread workgroups per CU next to 2 x 57 KB of LDS).  k_wavelocal has
// its own operating point (MPT_WL_THREADS / MPT_WL_WAVES below: 6 waves/SIMD, 80 VGPRs), k_ordered another (mpt_ordered.h).
#ifndef MPT_MIN_WAVES
#define MPT_MIN_WAVES 8
#endif

// =====================================================================================================
// device-side structures
// =====================================================================================================
struct QueueDev {  // struct-of-arrays ray queue, 16-byte records -> dwordx4 per lane, 1 KiB per wave access
    float4* od;    // (o.x, o.y, o.z, d.x)
    float4* dt;    // (d.y, d.z, thr.r, thr.g)
    float4* tl;    // (thr.b, L.r, L.g, L.b)
    uint2* ia;     // (path | bounce << 27, bits(L.a))
};

struct PassDesc {
    // written by k_advance, read by k_step
    uint32_t in_count[MPT_NSHARD];
    uint32_t item_prefix[MPT_NSHARD + 1];  // queue items of shards [0,s) ; [NSHARD] = all queue items
    uint32_t n_items;
    uint32_t regen_base;
    uint32_t range_end[MPT_NGROUP];
    // pass state
    uint32_t next_path, total_paths;
    uint32_t slots_items;  // wavefront width in 64-slot items
    uint32_t done, overflow, iterations;
    unsigned long long paths, rays, node_visits, aabb_hits, prim_tests, node_iters, prim_iters, leaf_phases;
    unsigned long long flagged, parked;  // closest-first pipeline: rays handed to the reference-order walk / parked for the tree
    unsigned long long trace_ticks;      // persistent trace kernels: first workgroup's start to last wave's end, in 100 MHz ticks, summed over the launches (note_wave_exit)
    unsigned long long t_first;          // ... the start of the running launch (announce_resident)
};
#define MPT_DESC_COUNTERS 11   // paths .. trace_ticks: cleared behind every statistics copy

// Atomic counters live on lines of their own (MPT_CTR_STRIDE words apart): device-scope atomics are
// executed at the memory side, and counters that share a line serialise there.
#define MPT_CTR_STRIDE 1024u  // in uint32 words = 4 KiB
#define MPT_CTR_CURSOR(g) ((g) * MPT_CTR_STRIDE)
#define MPT_CTR_OUT(s) ((MPT_NGROUP + (s)) * MPT_CTR_STRIDE)
#define MPT_CTR_STARTED ((MPT_NGROUP + MPT_NSHARD) * MPT_CTR_STRIDE)   // workgroups of the trace kernel that have started (announce_resident)
#define MPT_CTR_EXITED ((MPT_NGROUP + MPT_NSHARD + 1u) * MPT_CTR_STRIDE)   // waves of the trace kernel that have finished (note_wave_exit)
#define MPT_CTR_WORDS ((MPT_NGROUP + MPT_NSHARD + 2u) * MPT_CTR_STRIDE)
#define MPT_HOST_RESIDENT 8u   // word of the lane's pinned `done` block that receives the launch id once every workgroup of the launch is resident

struct PassParams {
    SceneDev scene;
    uint32_t* ctr;               // work cursors [NGROUP] and output counters [NSHARD], padded
    QueueDev q[2];
    uint32_t shard_cap;
    PassDesc* desc;
    float4* slots;               // per-path final radiance (clamped), index = path
    const uint32_t* pixel_seed;  // literal RNG: per-pixel u32 seed (host sin-hash, Fragment.metal:29)
    // camera (mpt_uniforms)
    F3 cam, first, vu, vv;
    float W, H;
    uint32_t width, height, tiles_x;
    uint32_t S, sample_begin;    // samples per pixel in this pass, first sample index
    uint32_t s_shift;            // log2(S) when S is a power of two, else 0xFF
    const uint32_t* tile_xy;     // this rank's tiles in processing order: x | y << 16
    uint32_t rank, nranks;
    uint32_t launch_id;          // != 0: the last workgroup to start writes it to host_done[MPT_HOST_RESIDENT] (announce_resident)
    uint32_t tile_magic;         // != 0: the rank's k-th tile is tile T = k * nranks + rank in row-major order (tile order 0) and
                                 // T / tiles_x = umulhi(T, tile_magic) exactly (the host checks the range): path -> pixel needs no table load
    ShadeParams sp;
    volatile uint32_t* host_done;
};

// Claim ranges of the single-launch pipelines: range g owns the rank's tiles k with k % MPT_NGROUP == g (every 8th
// tile along the rows, so all ranges cover the image uniformly, progress at the same rate and run dry together).
// A range is addressed by a virtual index v in [0, range_paths(g)): v -> tile k = (v / (S*64)) * 8 + g, sample
// (v / 64) % S, lane v % 64.
__device__ __forceinline__ uint32_t range_paths(uint32_t n_tiles, uint32_t S, uint32_t g) {
    const uint32_t n_g = n_tiles > g ? (n_tiles - g + MPT_NGROUP - 1u) / MPT_NGROUP : 0u;
    return n_g * S * 64u;
}
__device__ __forceinline__ uint32_t range_chunk_to_path_chunk(const PassParams& pp, uint32_t vchunk, uint32_t g) {
    uint32_t tv, s;
    if (pp.s_shift != 0xFFu) {
        tv = vchunk >> pp.s_shift;
        s = vchunk & (pp.S - 1u);
    } else {
        tv = vchunk / pp.S;
        s = vchunk - tv * pp.S;
    }
    return (tv * MPT_NGROUP + g) * pp.S + s;  // chunk index in the pass's path-id space: tile * S + sample
}

// per-wave statistics: reduce over the 64 lanes, one atomic per counter per wave
template <bool COUNT>
__device__ __forceinline__ void flush_stats(PassDesc* desc, uint32_t n_rays, uint32_t n_paths, const WorkCount& wc) {
    unsigned long long v[8] = {n_rays, n_paths, wc.node_visits, wc.aabb_hits, wc.prim_tests,
                               wc.node_iters, wc.prim_iters, wc.outer_iters};
    const int n = COUNT ? 8 : 2;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (k >= n) break;
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off);
    }
    if ((threadIdx.x & 63u) == 0) {
        unsigned long long* dst[8] = {&desc->rays, &desc->paths, &desc->node_visits, &desc->aabb_hits,
                                      &desc->prim_tests, &desc->node_iters, &desc->prim_iters, &desc->leaf_phases};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k >= n) break;
            if (v[k]) atomicAdd(dst[k], v[k]);
        }
    }
}

// path index -> pixel.  path = ((tile_local * S) + s) * 64 + lane; a wave = one 8x8 pixel tile.
// tile_xy[tile_local] = tile x | tile y << 16 of this rank's tile_local-th tile: the host lays the rank's tiles out
// in a strided (low-discrepancy) order so that any window of consecutive path ids mixes cheap (sky) and expensive
// (mesh) tiles — with row-major order the expensive tiles cluster and the end of a pass is all slow work.
__device__ __forceinline__ bool path_to_pixel(const PassParams& pp, uint32_t path, uint32_t& px, uint32_t& py,
                                              uint32_t& s) {
    const uint32_t lane = path & 63u, chunk = path >> 6;
    uint32_t tl;
    if (pp.s_shift != 0xFFu) {  // samples per pass is a power of two (the usual case): no division
        tl = chunk >> pp.s_shift;
        s = chunk & (pp.S - 1u);
    } else {
        tl = chunk / pp.S;
        s = chunk - tl * pp.S;
    }
    if (pp.tile_magic != 0u) {   // (wave-uniform) arithmetic instead of a dependent table load: the hit rings recompute the pixel of every popped record
        const uint32_t T = tl * pp.nranks + pp.rank, ty = __umulhi(T, pp.tile_magic), tx = T - ty * pp.tiles_x;
        px = tx * 8u + (lane & 7u);
        py = ty * 8u + (lane >> 3);
    } else {
        const uint32_t xy = pp.tile_xy[tl];
        px = (xy & 0xFFFFu) * 8u + (lane & 7u);
        py = (xy >> 16) * 8u + (lane >> 3);
    }
    return px < pp.width && py < pp.height;
}

// Fragment.metal:29-42 — seed, sub-pixel jitter, primary ray.
// (uvx, uvy): the pixel centre in [0,1]^2, Vertex.metal:5-17 — they depend on the pixel only, so k_wavelocal keeps them
// across the primary steps of one tile (two IEEE divisions less per step)
// The camera as gen_primary reads it: from the kernel arguments (cam_of) or, in k_wavelocal, from a copy in LDS — 14 scalar
// registers that the persistent step loop does not have to keep alive (or spill) between two primary steps.
struct CamView {
    F3 cam, first, vu, vv;
    float W, H;
};
__device__ __forceinline__ CamView cam_of(const PassParams& pp) { return CamView{pp.cam, pp.first, pp.vu, pp.vv, pp.W, pp.H}; }
__device__ __forceinline__ void pixel_uv(const CamView& cv, uint32_t px, uint32_t py, float& uvx, float& uvy) {
    uvx = ((float)px + 0.5f) / cv.W;
    uvy = ((float)py + 0.5f) / cv.H;
}
__device__ __forceinline__ void pixel_uv(const PassParams& pp, uint32_t px, uint32_t py, float& uvx, float& uvy) { pixel_uv(cam_of(pp), px, py, uvx, uvy); }
__device__ __forceinline__ void gen_primary(const PassParams& pp, const CamView& cv, uint32_t px, uint32_t py, float uvx, float uvy, uint32_t sample,
                                            PathState& ps, PathRngDev& g) {
    float xOff, yOff;
    g.pixel = py * pp.width + px;
    g.sample = sample;
    if (pp.sp.rng_mode == 0) {
        uint32_t seed = pp.pixel_seed[g.pixel];
        xOff = (pcg_float(seed) - 0.5f) / cv.W;
        seed = pcg_hash(seed);
        yOff = (pcg_float(seed) - 0.5f) / cv.H;
        seed = pcg_hash(seed);
        g.lit_seed = seed;
    } else {
        U4 r = philox4x32_10<true>(g.pixel, sample, 0xFFFFFFFFu, 0u, pp.sp.seed_lo, pp.sp.seed_hi);
        xOff = (u01(r.x) - 0.5f) / cv.W;
        yOff = (u01(r.y) - 0.5f) / cv.H;
        g.lit_seed = 0;
    }
    F3 dir = (cv.first + (uvx + xOff) * cv.vu + (uvy + yOff) * cv.vv) - cv.cam;
    ps.o = cv.cam;
    ps.d = normalize3(dir);
    ps.thr = f3(1, 1, 1);
    ps.L = f3(0, 0, 0);
    ps.La = 0.0f;
    ps.bounce = 0;
}

__device__ __forceinline__ void gen_primary(const PassParams& pp, uint32_t px, uint32_t py, float uvx, float uvy, uint32_t sample,
                                            PathState& ps, PathRngDev& g) {
    gen_primary(pp, cam_of(pp), px, py, uvx, uvy, sample, ps, g);
}
__device__ __forceinline__ void gen_primary(const PassParams& pp, uint32_t px, uint32_t py, uint32_t sample,
                                            PathState& ps, PathRngDev& g) {
    float uvx, uvy;
    pixel_uv(pp, px, py, uvx, uvy);
    gen_primary(pp, px, py, uvx, uvy, sample, ps, g);
}

// literal RNG: the seed entering rayColor is a pure function of the pixel; recompute it for bounce rays
__device__ __forceinline__ void rng_for_path(const PassParams& pp, uint32_t path, PathRngDev& g) {
    uint32_t px, py, s;
    path_to_pixel(pp, path, px, py, s);
    g.pixel = py * pp.width + px;
    g.sample = pp.sample_begin + s;
    g.lit_seed = 0;
    if (pp.sp.rng_mode == 0) g.lit_seed = pcg_hash(pcg_hash(pp.pixel_seed[g.pixel]));
}

__device__ __forceinline__ void stage_nodes(const SceneDev& sc, float4* lds) {
    const uint32_t n4 = sc.n_lds_nodes * 2u, p4 = sc.n_lds_prims * 3u;
    for (uint32_t i = threadIdx.x; i < n4; i += blockDim.x) lds[i] = sc.nodes[i];
    for (uint32_t i = threadIdx.x; i < p4; i += blockDim.x) lds[sc.lds_prim_off + i] = sc.prims[i];
    for (uint32_t i = threadIdx.x; i < sc.n_lds_mats * 2u; i += blockDim.x) lds[sc.lds_mat_off + i] = sc.mats[i];
    __syncthreads();
}

// Wave-uniform work fetch of the GLOBAL wavefront (k_step): a wave claims a run of consecutive 64-slot items from its
// home group's cursor (then steals from the other groups).  The run length is guided — remaining / (2 * waves) clamped
// to [1, 16] — so the bulk of an iteration costs few atomics and the tail stays fine-grained.  (k_wavelocal / k_ordered
// claim path ids with remaining / (16 * waves): see the comment at their claim code.)
// (every lane of the wave is active here, so readfirstlane returns lane 0's value as an SGPR)
__device__ __forceinline__ void fetch_items(uint32_t* ctr, const uint32_t* s_range_end, uint32_t home,
                                            uint32_t waves_per_group, uint32_t& first, uint32_t& last) {
    uint32_t f = MPT_NONE, l = 0;
    if ((threadIdx.x & 63u) == 0) {
        for (uint32_t t = 0; t < MPT_NGROUP; ++t) {
            uint32_t g = (home + t) & (MPT_NGROUP - 1);
            uint32_t end = s_range_end[g];
            uint32_t cur = __hip_atomic_load(&ctr[MPT_CTR_CURSOR(g)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur >= end) continue;
            uint32_t run = (end - cur) / (2u * waves_per_group);
            run = run < 1u ? 1u : (run > 16u ? 16u : run);
            uint32_t k = atomicAdd(&ctr[MPT_CTR_CURSOR(g)], run);
            if (k < end) {
                f = k;
                l = (k + run < end) ? k + run : end;
                break;
            }
        }
    }
    first = __builtin_amdgcn_readfirstlane(f);
    last = __builtin_amdgcn_readfirstlane(l);
}

// Global wavefront iteration (MPT_PIPE_WAVEFRONT).  Persistent workgroups; each WAVE pulls 64-slot work items from
// per-XCD cursors.  An item is either 64 consecutive entries of the device-wide SoA input ray queue or 64 new paths
// (primary rays are generated in registers and never touch HBM).  Per item: closest hit against the LDS-staged threaded
// BVH, one bounce of shading, then the surviving rays are compacted with a wave64 ballot + mbcnt prefix and appended to
// the output queue with ONE atomic per wave (16 sharded counters).  k_advance turns the counters into the next
// iteration's descriptor; iterations are enqueued without host round trips and drained ones exit at once.
template <bool COUNT, bool ALL_LDS>
__global__ __launch_bounds__(1024, MPT_MIN_WAVES) void k_step(PassParams pp, uint32_t parity) {
    extern __shared__ float4 lds_nodes[];
    PassDesc* desc = pp.desc;
    if (desc->n_items == 0) return;  // drained: iterations enqueued past the end of the pass cost a launch only
    // iteration descriptor -> LDS, behind the node + primitive image (16-byte aligned)
    uint32_t* s_prefix = (uint32_t*)(lds_nodes + 2 * pp.scene.n_lds_nodes + 3 * pp.scene.n_lds_prims + 2 * pp.scene.n_lds_mats);  // [NSHARD+1]
    uint32_t* s_incount = s_prefix + (MPT_NSHARD + 1);                        // [NSHARD]
    uint32_t* s_range_end = s_incount + MPT_NSHARD;                          // [NGROUP]
    if (threadIdx.x <= MPT_NSHARD) s_prefix[threadIdx.x] = desc->item_prefix[threadIdx.x];
    if (threadIdx.x < MPT_NSHARD) s_incount[threadIdx.x] = desc->in_count[threadIdx.x];
    if (threadIdx.x < MPT_NGROUP) s_range_end[threadIdx.x] = desc->range_end[threadIdx.x];
    stage_nodes(pp.scene, lds_nodes);

    const QueueDev qin = pp.q[parity], qout = pp.q[parity ^ 1u];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t home = blockIdx.x & (MPT_NGROUP - 1);
    const uint32_t nq_items = s_prefix[MPT_NSHARD];
    const uint32_t total_paths = desc->total_paths;
    const uint32_t regen_base = desc->regen_base;
    uint32_t n_rays = 0, n_paths = 0;
    WorkCount wc = {};

    const uint32_t waves_per_group = (gridDim.x * (blockDim.x >> 6) + MPT_NGROUP - 1) / MPT_NGROUP;
    uint32_t item = 0, item_last = 0;
    for (;; ++item) {
        if (item >= item_last) {
            fetch_items(pp.ctr, s_range_end, home, waves_per_group, item, item_last);
            if (item == MPT_NONE) break;
        }
        PathState ps;
        PathRngDev g;
        bool valid;
        if (item < nq_items) {  // 64 entries of the input queue
            uint32_t s = 0;
#pragma unroll
            for (uint32_t k = 1; k < MPT_NSHARD; ++k) s += (item >= s_prefix[k]) ? 1u : 0u;
            const uint32_t idx = (item - s_prefix[s]) * 64u + lane;
            valid = idx < s_incount[s];
            if (valid) {
                const uint32_t at = s * pp.shard_cap + idx;
                const float4 a = qin.od[at], b = qin.dt[at], c = qin.tl[at];
                const uint2 ia = qin.ia[at];
                ps.o = f3(a.x, a.y, a.z);
                ps.d = f3(a.w, b.x, b.y);
                ps.thr = f3(b.z, b.w, c.x);
                ps.L = f3(c.y, c.z, c.w);
                ps.La = __uint_as_float(ia.y);
                ps.path = ia.x & 0x07FFFFFFu;
                ps.bounce = ia.x >> 27;
                rng_for_path(pp, ps.path, g);
            }
        } else {  // 64 new paths: one 8x8 pixel tile at one sample index
            ps.path = regen_base + (item - nq_items) * 64u + lane;
            valid = ps.path < total_paths;
            if (valid) {
                uint32_t px, py, s;
                valid = path_to_pixel(pp, ps.path, px, py, s);
                if (valid) {
                    gen_primary(pp, px, py, pp.sample_begin + s, ps, g);
                    n_paths++;
                }
            }
        }
        bool alive = false;
        if (valid) {
            float t;
            int prim;
            closest_hit<COUNT, ALL_LDS>(pp.scene, (LdsNodes)lds_nodes, ps.o, ps.d, t, prim, wc);
            n_rays++;
            alive = shade_bounce(pp.scene, (LdsNodes)lds_nodes, pp.sp, g, ps, t, prim);
            if (!alive)  // PathTracing.h:258 per-sample clamp
                pp.slots[ps.path] = make_float4(clamp01(ps.L.x), clamp01(ps.L.y), clamp01(ps.L.z), clamp01(ps.La));
        }
        // wave64 stream compaction: ballot + prefix popcount, one atomic per wave
        const unsigned long long mask = __ballot(alive);
        if (mask != 0ull) {
            const uint32_t n = (uint32_t)__popcll(mask);
            const uint32_t shard = item & (MPT_NSHARD - 1);
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&pp.ctr[MPT_CTR_OUT(shard)], n);
            base = __shfl(base, 0);
            if (alive) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                                __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                const uint32_t idx = base + rank;
                if (idx < pp.shard_cap) {
                    const uint32_t at = shard * pp.shard_cap + idx;
                    qout.od[at] = make_float4(ps.o.x, ps.o.y, ps.o.z, ps.d.x);
                    qout.dt[at] = make_float4(ps.d.y, ps.d.z, ps.thr.x, ps.thr.y);
                    qout.tl[at] = make_float4(ps.thr.z, ps.L.x, ps.L.y, ps.L.z);
                    qout.ia[at] = make_uint2(ps.path | (ps.bounce << 27), __float_as_uint(ps.La));
                } else {
                    desc->overflow = 1u;
                }
            }
        }
    }
    flush_stats<COUNT>(desc, n_rays, n_paths, wc);
}

// one wave; lane 0 does the (tiny) serial work
__device__ void advance_desc(PassDesc* d, uint32_t* ctr, volatile uint32_t* host_done) {
    uint32_t items = 0;
    for (uint32_t s = 0; s < MPT_NSHARD; ++s) {
        uint32_t c = ctr[MPT_CTR_OUT(s)];
        d->in_count[s] = c;
        ctr[MPT_CTR_OUT(s)] = 0;
        d->item_prefix[s] = items;
        items += (c + 63u) >> 6;
    }
    d->item_prefix[MPT_NSHARD] = items;
    uint32_t room = d->slots_items > items ? d->slots_items - items : 0u;
    uint32_t left = (d->total_paths - d->next_path) >> 6;
    uint32_t regen = room < left ? room : left;
    d->regen_base = d->next_path;
    d->next_path += regen * 64u;
    uint32_t n = items + regen;
    d->n_items = n;
    for (uint32_t g = 0; g < MPT_NGROUP; ++g) {
        uint32_t b = (uint32_t)(((unsigned long long)n * g) / MPT_NGROUP);
        uint32_t e = (uint32_t)(((unsigned long long)n * (g + 1)) / MPT_NGROUP);
        ctr[MPT_CTR_CURSOR(g)] = b;
        d->range_end[g] = e;
    }
    d->done = (n == 0) ? 1u : 0u;
    if (n != 0) d->iterations++;
    if (host_done) *host_done = d->done | (d->overflow << 1);
}

// A persistent trace kernel tells the host when ALL its workgroups are resident (one thread per workgroup calls this first thing): from
// then on the workgroups of a kernel launched behind it can only take the slots its own workgroups give up as they finish — the
// overlap of the end of one pass with the start of the next that mpt_render_async wants — and never sit side by side with them from
// the start, each kernel on half of the chip for its whole life.
__device__ __forceinline__ void announce_resident(const PassParams& pp) {
    if (pp.launch_id == 0u) return;
    const uint32_t before = atomicAdd(&pp.ctr[MPT_CTR_STARTED], 1u);
    if (before == 0u) __hip_atomic_store(&pp.desc->t_first, (unsigned long long)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (before + 1u == gridDim.x)
        __hip_atomic_store((uint32_t*)pp.host_done + MPT_HOST_RESIDENT, pp.launch_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ... and how long it ran: the last wave to finish adds (now - start of the first workgroup) to the descriptor, on the chip's 100 MHz
// clock.  With renders that overlap, a HIP event in front of a launch fires when the launch is QUEUED behind the running kernel, not
// when it starts: event pairs then read 25-29 ms for a kernel that runs 16.6 (and the kernel trace of rocprofv3 says 16.6).
__device__ __forceinline__ void note_wave_exit(const PassParams& pp) {
    if (pp.launch_id == 0u || (threadIdx.x & 63u) != 0u) return;
    const uint32_t before = atomicAdd(&pp.ctr[MPT_CTR_EXITED], 1u);
    if (before + 1u == gridDim.x * (blockDim.x >> 6)) {
        const unsigned long long t0 = __hip_atomic_load(&pp.desc->t_first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicAdd(&pp.desc->trace_ticks, (unsigned long long)__builtin_amdgcn_s_memrealtime() - t0);
    }
}

__global__ void k_begin_pass(PassDesc* d, uint32_t* ctr, uint32_t total_paths, uint32_t slots_items,
                             volatile uint32_t* host_done, int path_cursors) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    ctr[MPT_CTR_STARTED] = 0u;
    ctr[MPT_CTR_EXITED] = 0u;
    if (path_cursors) {  // single-launch pipelines: the cursors count path ids, one contiguous range per group
        d->total_paths = total_paths;  // overflow stays sticky until the host has read it (enqueue_stats_copy clears it)
        for (uint32_t g = 0; g < MPT_NGROUP; ++g) ctr[MPT_CTR_CURSOR(g)] = 0u;
        return;
    }
    for (uint32_t s = 0; s < MPT_NSHARD; ++s) ctr[MPT_CTR_OUT(s)] = 0;
    d->next_path = 0;
    d->total_paths = total_paths;
    d->slots_items = slots_items;
    d->iterations = 0;  // overflow stays sticky until the host has read it (collect_pass_stats)
    advance_desc(d, ctr, host_done);
}

__global__ void k_advance(PassDesc* d, uint32_t* ctr, volatile uint32_t* host_done) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    advance_desc(d, ctr, host_done);
}

// Megakernel variant: one thread per path, whole bounce loop in registers (A/B baseline).
template <bool COUNT, bool ALL_LDS>
__global__ __launch_bounds__(1024, MPT_MIN_WAVES) void k_megakernel(PassParams pp) {
    extern __shared__ float4 lds_nodes[];
    stage_nodes(pp.scene, lds_nodes);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t total_paths = pp.desc->total_paths;
    const uint32_t waves_per_block = blockDim.x >> 6;
    uint32_t n_rays = 0, n_paths = 0;
    WorkCount wc = {};
    for (uint32_t chunk = blockIdx.x * waves_per_block + (threadIdx.x >> 6); chunk * 64u < total_paths;
         chunk += gridDim.x * waves_per_block) {
        PathState ps;
        PathRngDev g;
        ps.path = chunk * 64u + lane;
        uint32_t px, py, s;
        if (!path_to_pixel(pp, ps.path, px, py, s)) continue;
        gen_primary(pp, px, py, pp.sample_begin + s, ps, g);
        n_paths++;
        bool alive = true;
        while (alive) {
            float t;
            int prim;
            closest_hit<COUNT, ALL_LDS>(pp.scene, (LdsNodes)lds_nodes, ps.o, ps.d, t, prim, wc);
            n_rays++;
            alive = shade_bounce(pp.scene, (LdsNodes)lds_nodes, pp.sp, g, ps, t, prim);
        }
        pp.slots[ps.path] = make_float4(clamp01(ps.L.x), clamp01(ps.L.y), clamp01(ps.L.z), clamp01(ps.La));
    }
    flush_stats<COUNT>(pp.desc, n_rays, n_paths, wc);
}

// Result slots are written once and read once, much later, by the resolve kernel: streaming (non-temporal) accesses
// keep them from evicting the ring records the trace kernel is about to pop.
typedef float v4f_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_slot(float4* slots, uint32_t path, float r, float g, float b, float a) {
#ifdef MPT_SLOTS_TEMPORAL
    slots[path] = make_float4(r, g, b, a);
#else
    v4f_nt v = {r, g, b, a};
    __builtin_nontemporal_store(v, (v4f_nt*)(slots + path));
#endif
}
__device__ __forceinline__ float4 load_slot(const float4* slots, uint32_t idx) {
#ifdef MPT_SLOTS_TEMPORAL
    return slots[idx];
#else
    v4f_nt v = __builtin_nontemporal_load((const v4f_nt*)(slots + idx));
    return make_float4(v.x, v.y, v.z, v.w);
#endif
}

__device__ __forceinline__ uint32_t wave_rank(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Wave-local wavefront (MPT_PIPE_WAVELOCAL): the wavefront idea at wave scope, with rays sorted by remaining work.
// Every persistent wave owns MPT_WL_LEVELS private rings of ray records (SoA, 16-byte fields, in global memory but
// touched by this wave only) and runs full-width steps:
//   ring k holds >= 64 rays -> pop 64, continue their closest-hit queries for at most budget[k] box-test loop trips,
//                              shade the rays that finished (survivors are fresh rays -> ring 0), and PARK the rays
//                              that are still traversing in ring k+1 with their exact traversal state (next node,
//                              best t, best primitive).  The last ring has no budget.  Deepest ready ring first.
//   no ring is ready        -> take the next 8x8-pixel tile sample (64 new paths), generate the primary rays in
//                              registers, closest hit, one bounce of shading; survivors -> ring 0
// Compaction into the rings is a wave64 ballot + mbcnt prefix; ring counts are wave-uniform registers: no kernel
// boundary, no shared counter, no atomic per step.  Records are popped newest first (the "rings" are stacks unless
// MPT_WL_FIFO is defined): what was pushed a few steps ago is still in L2 / Infinity Cache and the live part of a ring
// stays below ~128 records — 6 % faster than first-in-first-out.  The order in which rays are traced does not affect
// any result: every ray writes only its own path's slot.
// Why budgets: a wave runs as long as its slowest lane, and bounce rays are heavy-tailed on this kind of scene — 88 %
// need <= 8 box tests, 10 % need 30-160 (they cross the mesh): a full wave of bounce rays used only 14 % of its
// box-test lane slots.  Budgets sort rays by remaining work, so every step runs rays of similar length (rounds 1-3: a ladder
// 8/20/50/125/inf over five rings, box-test lane slots per ray 37.8 -> 17.8, VALU instructions per ray 40 -> 24).  A parked ray
// resumes with exactly the state it stopped with and sees the same sequence of tests: results are bit-identical.
// Round 4, TWO rings: ring 0 (hits; their bounce rays get 8 trips) and ONE ring of unfinished walks that runs without a budget
// but regroups — a step ends once fewer than 24 of its lanes still traverse and the stragglers go back into the same ring.  Since
// ring 0 holds hits that are shaded at full width, the deeper ladder only cost scalar registers and leftovers at the end of a
// pass: 5 -> 2 rings is 17.69 -> 16.88 ms on scene.xml (3 rings 17.4-17.6, 4 rings 17.5-17.7, ONE ring = no budget at all 26.1);
// a 1/8 shard's step 2.46 -> 2.36 ms (a wave drains 2 x ~32 leftover rays, not 5 x ~32).
// Capacity: steps on rings never increase the total number of queued rays (64 out, <= 64 in) and a primary step
// (+<= 64) runs only when every ring holds < 64, so the total never exceeds 64 * LEVELS + 64 < MPT_WL_RING.
#ifndef MPT_WL_LEVELS
#define MPT_WL_LEVELS 2u
#endif
#define MPT_WL_RING 512u       // records per ring
#define MPT_LDS_MATS_N 32u     // materials staged in LDS (= MPT_LDS_MATS of mpt_hip.hip): the configuration block of k_wavelocal / k_ordered starts behind them
#define MPT_LDS_CFG_F4 16u     // ... and is this many float4 long (256 bytes): camera in float4 0..3, then 32 words (budgets 0..15, claim parameters 16..19).
                               // The host reserves it (MPT_LDS_EXTRA, ordered_views): k_ordered's stacks start right behind it.
// Where the configuration block starts in a workgroup's LDS (float4 index), for the kernels AND for the host code that lays the
// image out and sizes the launch (mpt_hip.hip: MPT_LDS_EXTRA, ordered_views) — one definition.  Round 4's development build of
// k_ordered had two: the kernel wrote the block at lds_mat_off + 2 * MPT_LDS_MATS_N while ordered_views still put the per-lane stacks at
// lds_mat_off + 2 * n_lds_mats, i.e. INSIDE the block's 256 bytes for any scene with fewer than 32 materials: stack pushes overwrote
// the claim parameters and budgets, the block's words were popped as child references, and the walk fetched nodes far outside the
// tree — the GPU memory fault behind the four aborted renders and the aborted test of gpurun_out/r04/s11_*.log (docs/HISTORY.md).
__host__ __device__ __forceinline__ uint32_t mpt_lds_cfg_off_f4(uint32_t lds_mat_off) { return lds_mat_off + 2u * MPT_LDS_MATS_N; }
__host__ __device__ __forceinline__ uint32_t mpt_lds_image_end_f4(uint32_t lds_mat_off) { return mpt_lds_cfg_off_f4(lds_mat_off) + MPT_LDS_CFG_F4; }
#define MPT_WL_BLOCK 1024u     // upper bound of a path-id claim
#define MPT_WL_NO_BUDGET 0x7FFFFFFFu  // budgets at or above this mean "run to completion"
// Ring record (round 4, "the traffic diet"): 48 bytes that every ray needs — od, dt, ia — and 16 more (tl: the light gathered so
// far) only for the rays that HAVE gathered light: L and alpha are all-zero bits until a path meets an emitter, which most bounce
// rays of most scenes never have (a flag in ia says whether tl was written; the pop restores exact zeros otherwise).  The sample
// index is recomputed from the path id.  Measured before building it (round 4's traffic what-if builds, in-kernel clock stamped: 16
// bytes MORE per push cost 3.0 % (19.23 -> 19.81 ms, clock 2.243 -> 2.203 GHz), dropping the result slots altogether — the bound of
// any slot diet — gains 3.2 %; tests/experiments/rejected_r04_flags.h).  (Round 3's 64-byte record is gone from the source.)
#define MPT_RING_HAS_LIGHT 0x100u
struct WaveRings {             // [n_waves][MPT_WL_LEVELS][MPT_WL_RING] records in five arrays of 16-byte fields, ONE allocation: the kernel
    float4* base;              // keeps one base pointer and the array length in scalar registers instead of five pointers
    uint32_t n;                // records per array = n_waves * MPT_WL_LEVELS * MPT_WL_RING
    __host__ __device__ float4* od() const { return base; }                        // (o.xyz, d.x)
    __host__ __device__ float4* dt() const { return base + n; }                    // (d.y, d.z, thr.r, thr.g)
    __host__ __device__ float4* tl() const { return base + 2u * (size_t)n; }       // (L.rgb, L.a), written only with MPT_RING_HAS_LIGHT
    __host__ __device__ uint4* ia() const { return (uint4*)(base + 3u * (size_t)n); }  // (thr.b bits, path, pixel, bounce | MPT_RING_HAS_LIGHT); ring 0 (hits): see ring_push_hit
    __host__ __device__ uint4* tv() const { return (uint4*)(base + 4u * (size_t)n); }  // rings >= 1: (next node, best t bits, best primitive, 0)
};
// what a push writes / a pop reads (both pipelines' rings use the same record)
__device__ __forceinline__ bool ring_has_light(const PathState& ps) {
    return (__float_as_uint(ps.L.x) | __float_as_uint(ps.L.y) | __float_as_uint(ps.L.z) | __float_as_uint(ps.La)) != 0u;
}
__device__ __forceinline__ uint32_t sample_of_path(const PassParams& pp, uint32_t path) {
    const uint32_t chunk = path >> 6;
    return pp.sample_begin + (pp.s_shift != 0xFFu ? chunk & (pp.S - 1u) : chunk % pp.S);
}
__device__ __forceinline__ void ring_push(const WaveRings& ring, uint32_t at, const PathState& ps, const PathRngDev& g) {
    ring.od()[at] = make_float4(ps.o.x, ps.o.y, ps.o.z, ps.d.x);
    ring.dt()[at] = make_float4(ps.d.y, ps.d.z, ps.thr.x, ps.thr.y);
    const bool lit = ring_has_light(ps);
    ring.ia()[at] = make_uint4(__float_as_uint(ps.thr.z), ps.path, g.pixel, ps.bounce | (lit ? MPT_RING_HAS_LIGHT : 0u));
    if (lit) ring.tl()[at] = make_float4(ps.L.x, ps.L.y, ps.L.z, ps.La);
}
__device__ __forceinline__ void ring_pop(const PassParams& pp, const WaveRings& ring, uint32_t at, PathState& ps, PathRngDev& g) {
    const float4 a = ring.od()[at], b = ring.dt()[at];
    const uint4 ia = ring.ia()[at];
    ps.o = f3(a.x, a.y, a.z);
    ps.d = f3(a.w, b.x, b.y);
    ps.thr = f3(b.z, b.w, __uint_as_float(ia.x));
    ps.path = ia.y;
    g.pixel = ia.z;
    ps.bounce = ia.w & 0xFFu;
    g.sample = sample_of_path(pp, ps.path);
    ps.L = f3(0.0f, 0.0f, 0.0f);
    ps.La = 0.0f;
    if ((ia.w & MPT_RING_HAS_LIGHT) != 0u) {
        const float4 cc = ring.tl()[at];
        ps.L = f3(cc.x, cc.y, cc.z);
        ps.La = cc.w;
    }
    g.lit_seed = 0;
    if (pp.sp.rng_mode == 0) g.lit_seed = pcg_hash(pcg_hash(pp.pixel_seed[g.pixel]));
}
// Ring 0 as a ring of HITS (round 4).  Round 3 shaded a ray where its closest hit
// was found: in a step of 64 rays about half hit a surface and half the sky, so the 250-instruction hit branch ran at half
// width — 8 wave-instructions per hit.  Now the step that finds a hit pushes the HIT (the ray, its t and primitive) to ring 0,
// and the step that pops 64 hits shades all of them first, at full width, and traces the 64 bounce rays right away: the same
// one ring hop per bounce as before, the same 48 (+ 16) bytes (t takes the place of the pixel index, which is recomputed from
// the path id, and the primitive shares a word with the bounce count), and every ray sees the same arithmetic in the same order.
// record: od, dt as above; ia = (thr.b bits, path, t bits | light flag in the sign bit (t > 0), primitive | bounce << 27); tl as above
__device__ __forceinline__ void ring_push_hit(const WaveRings& ring, uint32_t at, const PathState& ps, float t, uint32_t prim) {
    ring.od()[at] = make_float4(ps.o.x, ps.o.y, ps.o.z, ps.d.x);
    ring.dt()[at] = make_float4(ps.d.y, ps.d.z, ps.thr.x, ps.thr.y);
    const bool lit = ring_has_light(ps);
    ring.ia()[at] = make_uint4(__float_as_uint(ps.thr.z), ps.path, (__float_as_uint(t) & 0x7FFFFFFFu) | (lit ? 0x80000000u : 0u),
                               (prim & 0x07FFFFFFu) | (ps.bounce << 27));
    if (lit) ring.tl()[at] = make_float4(ps.L.x, ps.L.y, ps.L.z, ps.La);
}
__device__ __forceinline__ void ring_pop_hit(const WaveRings& ring, uint32_t at, PathState& ps, float& t, int& prim) {
    const float4 a = ring.od()[at], b = ring.dt()[at];
    const uint4 ia = ring.ia()[at];
    ps.o = f3(a.x, a.y, a.z);
    ps.d = f3(a.w, b.x, b.y);
    ps.thr = f3(b.z, b.w, __uint_as_float(ia.x));
    ps.path = ia.y;
    t = __uint_as_float(ia.z & 0x7FFFFFFFu);
    prim = (int)(ia.w & 0x07FFFFFFu);
    ps.bounce = ia.w >> 27;
    ps.L = f3(0.0f, 0.0f, 0.0f);
    ps.La = 0.0f;
    if ((ia.z & 0x80000000u) != 0u) {
        const float4 cc = ring.tl()[at];
        ps.L = f3(cc.x, cc.y, cc.z);
        ps.La = cc.w;
    }
}

struct WaveBudgets {
    uint32_t b[MPT_WL_LEVELS]; // box-test loop trips granted per step of ring k (last entry unused: unlimited)
    uint32_t min_active[MPT_WL_LEVELS];  // a step of ring k ends early once fewer lanes than this are still traversing
};

// Operating point (measured, 1080p x 256 spp): the kernel is VALU-issue-bound and the 64-VGPR cap of 8 waves/SIMD costs
// 92-150 B/lane of scratch spills in the hot loops — 6 waves/SIMD (80 VGPRs, no scratch with the BVH in LDS, workgroups
// of 768) is 19 % faster on scene.xml (36.8 -> 30.9 ms).  Scenes whose nodes come from L2 preferred 8 waves/SIMD while
// the kernel was still latency-sensitive (FIFO rings, cached result slots); with the current kernel 6 waves/SIMD wins
// there too (bunny x20 6.15 -> 6.82 Grays/s, 1 M triangles 4.40 -> 5.25).  7 waves/SIMD (896 threads, 72 VGPRs, waves
// unevenly spread over the SIMDs) and 5 are worse than both.
#ifndef MPT_WL_THREADS_N
#define MPT_WL_THREADS_N 768
#endif
#ifndef MPT_WL_WAVES_N
#define MPT_WL_WAVES_N 6
#endif
#define MPT_WL_THREADS(ALL_LDS) MPT_WL_THREADS_N
#define MPT_WL_WAVES(ALL_LDS) MPT_WL_WAVES_N
template <bool COUNT, bool ALL_LDS>
__device__ __forceinline__ void wavelocal_body(const PassParams& pp, const WaveRings& ring, const WaveBudgets& budgets, uint32_t wl_block, uint32_t wl_min,
                                               uint32_t wl_div);
template <bool COUNT, bool ALL_LDS>
__global__ __launch_bounds__(MPT_WL_THREADS(ALL_LDS), MPT_WL_WAVES(ALL_LDS)) void k_wavelocal(PassParams pp, WaveRings ring, WaveBudgets budgets,
                                                                                             uint32_t wl_block, uint32_t wl_min, uint32_t wl_div) {
    wavelocal_body<COUNT, ALL_LDS>(pp, ring, budgets, wl_block, wl_min, wl_div);
}
// The same kernel held to 96 scalar registers, for renders that overlap (mpt_render_async).  A wave's scalar registers are allocated
// in blocks of 16 plus 16: with the 102 the compiler takes when it is free to, six waves per SIMD hold 6 x 128 = 768 of the SIMD's
// 800 and NOTHING else can start on the CU — not the resolve of the render before, not one wave of anybody's kernel (measured: a
// one-element torch kernel launched beside the resident trace kernel waits 13-15 ms for it to end; tools/gpu_corun.py).  At <= 96 it
// is 6 x 112 = 672 and 128 are left: the resolve's two waves per SIMD run beside it.  13 more scalar values live in VGPR lanes for
// that: a render on its own is 1.7 % slower with this variant (17.5 against 17.2 ms), which is why mpt_render keeps the other one.
#ifndef MPT_WL_CORUN_SGPRS
#define MPT_WL_CORUN_SGPRS 96
#endif
template <bool ALL_LDS>
__global__ __launch_bounds__(MPT_WL_THREADS(ALL_LDS), MPT_WL_WAVES(ALL_LDS)) __attribute__((amdgpu_num_sgpr(MPT_WL_CORUN_SGPRS)))
void k_wavelocal_corun(PassParams pp, WaveRings ring, WaveBudgets budgets, uint32_t wl_block, uint32_t wl_min, uint32_t wl_div) {
    wavelocal_body<false, ALL_LDS>(pp, ring, budgets, wl_block, wl_min, wl_div);
}
template <bool COUNT, bool ALL_LDS>
__device__ __forceinline__ void wavelocal_body(const PassParams& pp, const WaveRings& ring, const WaveBudgets& budgets, uint32_t wl_block, uint32_t wl_min,
                                               uint32_t wl_div) {
    extern __shared__ float4 lds_nodes_raw[];
    // what only some steps need goes to LDS, behind the scene image (the 256-byte descriptor area of MPT_LDS_EXTRA), instead of
    // living in scalar registers across the whole step loop: the camera (14 words, primary steps) and the ring budgets (10 words,
    // ring steps).  The loop needs more scalar registers than the 102 a wave has; every value kept out of it is one spill less.
    const uint32_t cfg_off = mpt_lds_cfg_off_f4(pp.scene.lds_mat_off);   // in float4 units
    if (threadIdx.x == 0) {
        announce_resident(pp);
        lds_nodes_raw[cfg_off + 0] = make_float4(pp.cam.x, pp.cam.y, pp.cam.z, pp.W);
        lds_nodes_raw[cfg_off + 1] = make_float4(pp.first.x, pp.first.y, pp.first.z, pp.H);
        lds_nodes_raw[cfg_off + 2] = make_float4(pp.vu.x, pp.vu.y, pp.vu.z, 0.0f);
        lds_nodes_raw[cfg_off + 3] = make_float4(pp.vv.x, pp.vv.y, pp.vv.z, 0.0f);
        uint32_t* w = (uint32_t*)(lds_nodes_raw + cfg_off + 4);
        for (uint32_t k = 0; k < MPT_WL_LEVELS; ++k) {
            w[k] = budgets.b[k];
            w[8u + k] = budgets.min_active[k];
        }
        // ... and what only a CLAIM of path ids needs (words 16..19 of the block's 32: static_assert below)
        static_assert(4u * (MPT_LDS_CFG_F4 - 4u) >= 20u, "configuration block too small");
        w[16] = wl_block;
        w[17] = wl_min;
        w[18] = wl_div * (((uint32_t)(gridDim.x * (blockDim.x >> 6)) + MPT_NGROUP - 1u) / MPT_NGROUP);   // wl_div * waves per claim range
        w[19] = pp.desc->total_paths / (pp.S * 64u);                                                   // this rank's tiles
    }
    stage_nodes(pp.scene, lds_nodes_raw);
    MPT_CLOCK_BEGIN();
    const LdsNodes lds_nodes = (LdsNodes)lds_nodes_raw;
    const __attribute__((address_space(3))) uint32_t* lds_cfg_u32 = (const __attribute__((address_space(3))) uint32_t*)(lds_nodes + cfg_off + 4u);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave_id = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
#ifdef MPT_DEBUG_WAVE_TIMES
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
#endif
    const uint32_t wbase = wave_id * (MPT_WL_LEVELS * MPT_WL_RING);
    const uint32_t M = MPT_WL_RING - 1u;
    uint32_t head[MPT_WL_LEVELS], cnt[MPT_WL_LEVELS];  // wave-uniform ring state (fully unrolled accesses)
#pragma unroll
    for (uint32_t k = 0; k < MPT_WL_LEVELS; ++k) head[k] = cnt[k] = 0;
    uint32_t cur = 0, end = 0;     // wave-uniform private range of path ids (multiples of 64)
    uint32_t grp = blockIdx.x & (MPT_NGROUP - 1u);     // range this wave currently claims from
    uint32_t seen = 0;                                 // cursor value (virtual index) at this wave's previous claim
    bool exhausted = false;
    uint32_t tile_cached = 0xFFFFFFFFu, tile_xy_cached = 0u;
    float uvx_cached = 0.0f, uvy_cached = 0.0f;   // pixel_uv of this lane's pixel in the cached tile
    uint32_t n_rays = 0, n_paths = 0;
    WorkCount wc = {};
#ifdef MPT_DEBUG_WAVE_TIMES
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
    unsigned long long t_exh = 0ull, t_claim = 0ull, dbg_steps = 0ull, dbg_left = 0ull, dbg_blk = 0ull;
    unsigned long long reg_select = 0, reg_fetch = 0, reg_trace = 0, reg_shade = 0, reg_push = 0;  // cycles per region
#endif
    for (;;) {
        MPT_TIC(tic_);
        // ---- step choice: deepest ring with a full wave of rays; else new paths; else drain ----------------------
        int level = -1;  // -1 = primary step
#pragma unroll
        for (int k = (int)MPT_WL_LEVELS - 1; k >= 0; --k)
            if (level < 0 && cnt[k] >= 64u) level = k;
        if (level < 0) {
            if (!exhausted && cur == end) {
                // Guided self-scheduling of path ids.  The pass's tiles are dealt to MPT_NGROUP interleaved ranges,
                // each with its own cursor on its own line (a single cursor saturates at ~88 claims/us: in the sky
                // part of the image a step takes ~1 us and 8192 waves would queue on it).  A wave claims from its
                // home range (blockIdx % 8: the blocks of one XCD under round-robin placement), then steals from the
                // next ranges.  Claim size = remaining_in_range / (wl_div * waves_per_range) rounded to whole 64-path
                // tile samples and clamped to [wl_min, wl_block]: few atomics while there is plenty of work, fine grain
                // at the end (tile samples differ ~5x in cost between sky and geometry; with remaining/(2*waves) the
                // last wave finished 23 ms after the first, with /16 within ~1 ms).  `seen` is the cursor value of
                // this wave's previous claim: an extra load of the hot cursor line before the atomic made the kernel
                // 4x slower (loads of a line under atomic fire serialise at the memory side).
                uint32_t k = 0, blk = 0, rend = 0;
                bool got = false;
                if (lane == 0) {
                    const uint32_t c_block = lds_cfg_u32[16], c_min = lds_cfg_u32[17], c_div = lds_cfg_u32[18], c_tiles = lds_cfg_u32[19];   // (claim parameters: from LDS)
                    for (uint32_t t = 0; t < MPT_NGROUP && !got; ++t) {
                        const uint32_t re = range_paths(c_tiles, pp.S, grp);
                        const uint32_t left = seen < re ? re - seen : 0u;
                        blk = (left / c_div) & ~63u;
                        blk = blk < c_min ? c_min : (blk > c_block ? c_block : blk);
                        // a range entered by stealing is near its end but `seen` knows nothing about it yet: the first
                        // claim there is the minimum (it returns the cursor, which sizes the following ones)
                        if (t > 0u) blk = c_min;
                        k = atomicAdd(&pp.ctr[MPT_CTR_CURSOR(grp)], blk);
                        if (k < re) {
                            got = true;
                            rend = re;
                        } else {
                            grp = (grp + 1u) & (MPT_NGROUP - 1u);
                            seen = 0;
                        }
                    }
                }
                got = __builtin_amdgcn_readfirstlane((int)got) != 0;
                k = __builtin_amdgcn_readfirstlane(k);
                blk = __builtin_amdgcn_readfirstlane(blk);
                rend = __builtin_amdgcn_readfirstlane(rend);
                grp = __builtin_amdgcn_readfirstlane(grp);
                seen = k;
#ifdef MPT_DEBUG_WAVE_TIMES
                if (got) {
                    t_claim = __builtin_amdgcn_s_memrealtime();
                    dbg_steps = 0ull;
                    dbg_blk = blk;
                } else {
                    t_exh = __builtin_amdgcn_s_memrealtime();
                    for (uint32_t q = 0; q < MPT_WL_LEVELS; ++q) dbg_left |= (unsigned long long)cnt[q] << (12u * q);
                }
#endif
                if (!got) {
                    exhausted = true;
                } else {
                    cur = k;
                    end = (k + blk < rend) ? k + blk : rend;
                }
            }
            if (exhausted) {
                // Drain: no new paths and no ring holds a full wave.  What is left in ALL rings is merged into
                // full-width steps without a budget (deepest ring first).  Measured alternatives: one partial step per
                // ring and bounce generation (1.4-2.2 ms per wave instead of 0.5-0.9), merged steps that keep the
                // largest finite budget and park heavy rays in the unbudgeted ring (3 % slower at 32 spp, equal at
                // 256), staggered retirement of the waves of a workgroup (no gain).  The drain is bound by the
                // latency of the heaviest rays (~1000 box tests, ~100 us each even on an idle SIMD) times the bounce
                // chain, not by the number of rays left.
                uint32_t total = 0;
#pragma unroll
                for (int k = 0; k < (int)MPT_WL_LEVELS; ++k) total += cnt[k];
                if (total == 0u) break;
                level = (int)MPT_WL_LEVELS;  // merged drain step
            }
        }
#ifdef MPT_DEBUG_WAVE_TIMES
        if (!exhausted) dbg_steps += 1ull << (12u * (uint32_t)(level + 1));  // field 0 = primary steps
#endif
        MPT_TOC(reg_select, tic_);
        PathState ps;
        PathRngDev g;
        bool valid = false;
        uint32_t node = 0;
        float best_t = INFINITY;
        int best_prim = -1;
        uint32_t budget = 0xFFFFFFFFu, min_active = 0u;
        bool fresh = false;  // this lane starts a new closest-hit query (primary ray or ring-0 record)
        bool need_shade = false;   // this lane popped a hit from ring 0
        float hit_t = 0.0f;
        int hit_prim = -1;
        if (level < 0) {
            fresh = true;
            // 64 new paths = one 8x8 pixel tile at one sample index: everything about the tile is wave-uniform, and its
            // table entry is kept from the previous primary step (the steps of a claim walk the samples of one tile) —
            // a dependent vector load from L2 at the head of every primary step otherwise
            const uint32_t pchunk = range_chunk_to_path_chunk(pp, cur >> 6, grp);  // cur is a virtual index of range grp
            cur += 64u;
            uint32_t tl, sidx;
            if (pp.s_shift != 0xFFu) {
                tl = pchunk >> pp.s_shift;
                sidx = pchunk & (pp.S - 1u);
            } else {
                tl = pchunk / pp.S;
                sidx = pchunk - tl * pp.S;
            }
            const bool new_tile = tl != tile_cached;
            if (new_tile) {
                tile_cached = tl;
                tile_xy_cached = (uint32_t)__builtin_amdgcn_readfirstlane((int)pp.tile_xy[tl]);
            }
            ps.path = pchunk * 64u + lane;
            const uint32_t px = (tile_xy_cached & 0xFFFFu) * 8u + (lane & 7u), py = (tile_xy_cached >> 16) * 8u + (lane >> 3);
            CamView cv;
            {
                const v4f c0 = lds_nodes[cfg_off], c1 = lds_nodes[cfg_off + 1u], c2 = lds_nodes[cfg_off + 2u], c3 = lds_nodes[cfg_off + 3u];
                cv.cam = f3(c0.x, c0.y, c0.z);
                cv.first = f3(c1.x, c1.y, c1.z);
                cv.vu = f3(c2.x, c2.y, c2.z);
                cv.vv = f3(c3.x, c3.y, c3.z);
                cv.W = c0.w;
                cv.H = c1.w;
            }
            if (new_tile) pixel_uv(cv, px, py, uvx_cached, uvy_cached);
            if (px < pp.width && py < pp.height) {
                gen_primary(pp, cv, px, py, uvx_cached, uvy_cached, pp.sample_begin + sidx, ps, g);
                valid = true;
                n_paths++;
            }
        } else {
            // lanes -> ring records.  Normal step: 64 records of ring `level`.  Merged drain step: up to 64 records
            // taken from all rings, deepest first (lane ranges [lo_k, lo_k + take_k) per ring).
            uint32_t my_ring = 0, my_off = 0;
            bool take = false;
            uint32_t assigned = 0;
#pragma unroll
            for (int k = (int)MPT_WL_LEVELS - 1; k >= 0; --k) {
                const bool use = (level == (int)MPT_WL_LEVELS) || (level == k);
                uint32_t tk = use ? cnt[k] : 0u;
                tk = tk < 64u - assigned ? tk : 64u - assigned;
                if (lane >= assigned && lane < assigned + tk) {
                    take = true;
                    my_ring = (uint32_t)k;
#ifdef MPT_WL_FIFO
                    my_off = (head[k] + (lane - assigned)) & M;
#else
                    my_off = cnt[k] - tk + (lane - assigned);  // newest records first: they are still in L2
#endif
                }
                (void)k;
#ifdef MPT_WL_FIFO
                head[k] = (head[k] + tk) & M;
#endif
                cnt[k] -= tk;
                assigned += tk;
            }
            if (level < (int)MPT_WL_LEVELS) {   // (a ring step: its budget and its minimum of working lanes, from the LDS copy)
                budget = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_cfg_u32[level]);
                min_active = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_cfg_u32[8 + level]);
            }
            // merged drain step: no trip budget, but the stragglers of a step (fewer than a third of the lanes it
            // started with) go back to the last ring and are traced together once the rest is gone (4.76 -> 4.54 ms
            // for a 32-spp pass)
            if (level == (int)MPT_WL_LEVELS && assigned >= 12u) min_active = assigned / 3u;
            if (take) {
                const uint32_t at = wbase + my_ring * MPT_WL_RING + my_off;
                uint4 tv = make_uint4(0u, 0u, 0u, 0u);
                if (my_ring > 0u) tv = ring.tv()[at];   // (issued with the record's other loads, ahead of ring_pop's conditional one)
                ring_pop_hit(ring, at, ps, hit_t, hit_prim);
                need_shade = my_ring == 0u;
                if (my_ring > 0u) {
                    node = tv.x;
                    best_t = __uint_as_float(tv.y);
                    best_prim = (int)tv.z;
                } else {
                    fresh = true;
                }
                valid = true;
            }
        }
        // ---- the hits popped from ring 0 are shaded first — all 64 lanes of a ring-0 step — and leave their bounce rays in `ps` ------
        if (need_shade) {
            uint32_t px, py, sidx;
            path_to_pixel(pp, ps.path, px, py, sidx);
            g.pixel = py * pp.width + px;
            g.sample = pp.sample_begin + sidx;
            g.lit_seed = 0;
            if (pp.sp.rng_mode == 0) g.lit_seed = pcg_hash(pcg_hash(pp.pixel_seed[g.pixel]));
            if (!shade_bounce(pp.scene, lds_nodes, pp.sp, g, ps, hit_t, hit_prim)) {   // the path ends here (depth limit, material guard)
                store_slot(pp.slots, ps.path, clamp01(ps.L.x), clamp01(ps.L.y), clamp01(ps.L.z), clamp01(ps.La));
                valid = false;
            }
        }
        // the last ring never has a budget; a ring whose budget is "none" runs the plain (unsynchronised) loop
        // (the last ring has no trip budget; with a min_active rule its stragglers go back on top of the same ring)
        const bool budgeted = level >= 0 && (level < (int)MPT_WL_LEVELS || min_active != 0u) &&
                              ((level < (int)MPT_WL_LEVELS - 1 && budget < MPT_WL_NO_BUDGET) || min_active != 0u);
        const int park_ring = level + 1 < (int)MPT_WL_LEVELS ? level + 1 : (int)MPT_WL_LEVELS - 1;  // where unfinished queries go
        bool alive = false, parked = false;
        MPT_TOC(reg_fetch, tic_);
#ifdef MPT_DEBUG_WAVE_TIMES
        const WorkCount wc0 = wc;
#endif
        if (valid) {
            bool done;
            if (budgeted)
                done = closest_hit_resume<COUNT, ALL_LDS, true>(pp.scene, lds_nodes, ps.o, ps.d, node, best_t, best_prim,
                                                               budget, wc, min_active);
            else
                done = closest_hit_resume<COUNT, ALL_LDS, false>(pp.scene, lds_nodes, ps.o, ps.d, node, best_t, best_prim,
                                                                0xFFFFFFFFu, wc);
            if (fresh) n_rays++;  // a resumed query was counted when it started
            MPT_TOC(reg_trace, tic_);
#ifdef MPT_DEBUG_WAVE_TIMES
            if (COUNT) {  // per-level divergence diagnostics: [level + 1][steps, box trips, box lane work, prim trips, prim lane work]
                unsigned long long* lv = (unsigned long long*)(ring.tv() + (size_t)n_waves * MPT_WL_LEVELS * MPT_WL_RING) +
                                         16ull * n_waves + 8ull * (unsigned)(level + 1);
                if (first_active_lane()) atomicAdd(lv + 0, 1ull);
                // the per-wave trip counters live in whichever lane was first active at the time: sum over lanes
                if (wc.node_iters != wc0.node_iters) atomicAdd(lv + 1, (unsigned long long)(wc.node_iters - wc0.node_iters));
                if (wc.prim_iters != wc0.prim_iters) atomicAdd(lv + 3, (unsigned long long)(wc.prim_iters - wc0.prim_iters));
                atomicAdd(lv + 2, (unsigned long long)(wc.node_visits - wc0.node_visits));
                atomicAdd(lv + 4, (unsigned long long)(wc.prim_tests - wc0.prim_tests));
                atomicAdd(lv + 5, 1ull);  // rays in the step
                if (wc.wait_leaf != wc0.wait_leaf) atomicAdd(lv + 7, (unsigned long long)(wc.wait_leaf - wc0.wait_leaf));
                if (done) atomicAdd(lv + 6, 1ull);
            }
#endif
            if (done) {
                alive = best_prim >= 0;   // a hit: to ring 0 as it is, shaded by the step that pops it
                if (!alive) {             // the sky ends the path (PathTracing.h:225-232)
                    shade_bounce(pp.scene, lds_nodes, pp.sp, g, ps, best_t, -1);
                    store_slot(pp.slots, ps.path, clamp01(ps.L.x), clamp01(ps.L.y), clamp01(ps.L.z), clamp01(ps.La));
                }
            } else {
                parked = true;
            }
        }
        MPT_TOC(reg_shade, tic_);
        const unsigned long long am = __ballot(alive), pm = __ballot(parked);
        if (am != 0ull) {  // survivors are fresh rays -> ring 0
            if (alive) {
                const uint32_t at = wbase + ((head[0] + cnt[0] + wave_rank(am)) & M);
                ring_push_hit(ring, at, ps, best_t, (uint32_t)best_prim);
            }
            cnt[0] += (uint32_t)__popcll(am);
        }
        if (pm != 0ull) {  // unfinished queries -> next ring, with their traversal state (only from budgeted steps)
            uint32_t h = 0, c = 0;
#pragma unroll
            for (int k = 1; k < (int)MPT_WL_LEVELS; ++k)
                if (k == park_ring) {
                    h = head[k];
                    c = cnt[k];
                }
            if (parked) {
                const uint32_t at = wbase + (uint32_t)park_ring * MPT_WL_RING + ((h + c + wave_rank(pm)) & M);
                ring_push_hit(ring, at, ps, 0.0f, 0u);   // (a parked ray: its traversal state is in tv)
                ring.tv()[at] = make_uint4(node, __float_as_uint(best_t), (uint32_t)best_prim, 0u);
            }
#pragma unroll
            for (int k = 1; k < (int)MPT_WL_LEVELS; ++k)
                if (k == park_ring) cnt[k] = c + (uint32_t)__popcll(pm);
        }
        uint32_t worst = 0;
#pragma unroll
        for (uint32_t k = 0; k < MPT_WL_LEVELS; ++k) worst = cnt[k] > worst ? cnt[k] : worst;
        if (worst > MPT_WL_RING) pp.desc->overflow = 1u;  // cannot happen (see capacity note above)
        MPT_TOC(reg_push, tic_);
    }
#ifdef MPT_DEBUG_WAVE_TIMES  // diagnostics build: per-wave (start, cursor exhausted, end) timestamps, 100 MHz ticks
    if (lane == 0) {
        unsigned long long* dbg = (unsigned long long*)(ring.tv() + (size_t)n_waves * MPT_WL_LEVELS * MPT_WL_RING);
        dbg[8 * wave_id] = t_start;
        dbg[8 * wave_id + 1] = t_exh;
        dbg[8 * wave_id + 2] = __builtin_amdgcn_s_memrealtime();
        dbg[8 * wave_id + 3] = t_claim;
        dbg[8 * wave_id + 4] = dbg_steps;
        dbg[8 * wave_id + 5] = dbg_left;
        dbg[8 * wave_id + 6] = dbg_blk;
        unsigned long long* reg = dbg + 8ull * n_waves + 8ull * wave_id;
        reg[0] = reg_select;
        reg[1] = reg_fetch;
        reg[2] = reg_trace;
        reg[3] = reg_shade;
        reg[4] = reg_push;
        reg[5] = wc.t_box;
        reg[6] = wc.t_leaf;
    }
#endif
    MPT_CLOCK_END();
    flush_stats<COUNT>(pp.desc, n_rays, n_paths, wc);
    note_wave_exit(pp);
}

// sum[pixel] += sum over the pass's samples (in sample order) of the clamped per-sample colour.
// Grid-stride: the host chooses the footprint.  A resolve that runs BESIDE the next render's trace kernel gets two workgroups of 256 per
// CU — 2 waves per SIMD next to the trace kernel's 6 and inside the VGPRs it leaves (launch bound 256 x 8: <= 64), so that both fit on every
// CU whichever is dispatched first; a resolve with the chip to itself gets a workgroup per 256 pixels.  Four loads in flight per thread;
// the additions stay in sample order.
__global__ __launch_bounds__(256, 8) void k_resolve_sum(PassParams pp, float4* sum, uint32_t n_local_tiles) {
    const uint32_t n = n_local_tiles * 64u;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t tl = i >> 6, lane = i & 63u;
        uint32_t px, py, s0;
        if (!path_to_pixel(pp, (tl * pp.S) * 64u + lane, px, py, s0)) continue;
        float4 acc = sum[py * pp.width + px];
        const uint32_t base = tl * pp.S;
        uint32_t s = 0;
        for (; s + 4u <= pp.S; s += 4u) {
            const float4 v0 = load_slot(pp.slots, (base + s) * 64u + lane), v1 = load_slot(pp.slots, (base + s + 1u) * 64u + lane),
                         v2 = load_slot(pp.slots, (base + s + 2u) * 64u + lane), v3 = load_slot(pp.slots, (base + s + 3u) * 64u + lane);
            acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
            acc.x += v1.x; acc.y += v1.y; acc.z += v1.z; acc.w += v1.w;
            acc.x += v2.x; acc.y += v2.y; acc.z += v2.z; acc.w += v2.w;
            acc.x += v3.x; acc.y += v3.y; acc.z += v3.z; acc.w += v3.w;
        }
        for (; s < pp.S; ++s) {
            const float4 v = load_slot(pp.slots, (base + s) * 64u + lane);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        sum[py * pp.width + px] = acc;
    }
}

// Fragment.metal:23-27,62-69 — running mean with the frameCount+1 weight and the clamp.
__global__ void k_resolve_frame(PassParams pp, const float4* last, float4* cur, uint32_t n_local_tiles,
                                unsigned long long frameCount) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_local_tiles * 64u) return;
    uint32_t tl = i >> 6, lane = i & 63u;
    uint32_t px, py, s0;
    if (!path_to_pixel(pp, tl * 64u + lane, px, py, s0)) return;
    float4 c = pp.slots[tl * 64u + lane];
    float4 l = make_float4(0, 0, 0, 0);
    if (frameCount != 0) l = last[py * pp.width + px];
    unsigned long long fc = frameCount + 1ull;
    float w = (float)(fc - 1ull), fcf = (float)fc;
    float4 o;
    o.x = clamp01((c.x + l.x * w) / fcf);
    o.y = clamp01((c.y + l.y * w) / fcf);
    o.z = clamp01((c.z + l.z * w) / fcf);
    o.w = clamp01((c.w + l.w * w) / fcf);
    cur[py * pp.width + px] = o;
}

// unit-test kernels ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_trace_rays(SceneDev sc, const float* o, const float* d, uint32_t n,
                                                     float* t_out, int* prim_out, float* n_out, int* front_out) {
    extern __shared__ float4 lds_nodes[];
    stage_nodes(sc, lds_nodes);
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    F3 ro = f3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), rd = f3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    float t;
    int prim;
    WorkCount wc = {};
    closest_hit<false, false>(sc, (LdsNodes)lds_nodes, ro, rd, t, prim, wc);
    t_out[i] = t;
    if (prim >= 0) {
        HitInfo h = finish_hit(sc, (LdsNodes)lds_nodes, ro, rd, t, prim);
        prim_out[i] = h.orig_id;
        n_out[3 * i] = h.normal.x;
        n_out[3 * i + 1] = h.normal.y;
        n_out[3 * i + 2] = h.normal.z;
        front_out[i] = h.front ? 1 : 0;
    } else {
        prim_out[i] = -1;
        n_out[3 * i] = n_out[3 * i + 1] = n_out[3 * i + 2] = 0.0f;
        front_out[i] = 0;
    }
}
__global__ void k_kat_pcg(const uint32_t* s, uint32_t n, uint32_t* h, float* f) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        h[i] = pcg_hash(s[i]);
        f[i] = pcg_float(s[i]);
    }
}
__global__ void k_kat_philox(const uint32_t* c, const uint32_t* k, uint32_t n, uint32_t* o) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        U4 r = philox4x32_10(c[4 * i], c[4 * i + 1], c[4 * i + 2], c[4 * i + 3], k[2 * i], k[2 * i + 1]);
        o[4 * i] = r.x;
        o[4 * i + 1] = r.y;
        o[4 * i + 2] = r.z;
        o[4 * i + 3] = r.w;
    }
}
// All 2^32 operands: rcp_chain(x) against the compiler's correctly rounded 1.0f / x, bit for bit (mpt_device.h, mpt_rcp).
// out = {mismatches with |x| in [MPT_RCP_LO, MPT_RCP_HI], operands in that range, mismatches outside it, operands outside it}
__global__ void k_kat_rcp(unsigned long long* out) {
    unsigned long long bad_in = 0, n_in = 0, bad_out = 0, n_out = 0;
    const uint32_t stride = gridDim.x * blockDim.x;   // a power of two that divides 2^32 (the host launches 65536 x 256)
    uint32_t bits = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t it = 0; it < (uint32_t)(0x100000000ull / stride); ++it, bits += stride) {
        const float x = __uint_as_float(bits);
        const uint32_t a = __float_as_uint(rcp_chain(x)), b = __float_as_uint(1.0f / x);
        const bool same = a == b || ((a & 0x7FFFFFFFu) > 0x7F800000u && (b & 0x7FFFFFFFu) > 0x7F800000u);   // (two NaNs are the same answer)
        if (rcp_chain_exact(x)) {
            n_in++;
            bad_in += same ? 0u : 1u;
        } else {
            n_out++;
            bad_out += same ? 0u : 1u;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        bad_in += __shfl_down(bad_in, off);
        n_in += __shfl_down(n_in, off);
        bad_out += __shfl_down(bad_out, off);
        n_out += __shfl_down(n_out, off);
    }
    if ((threadIdx.x & 63u) == 0) {
        if (bad_in) atomicAdd(out + 0, bad_in);
        atomicAdd(out + 1, n_in);
        if (bad_out) atomicAdd(out + 2, bad_out);
        atomicAdd(out + 3, n_out);
    }
}
__global__ void k_kat_sincos(const float* u, uint32_t n, float* s, float* c) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) sincos_2pi(u[i], s[i], c[i]);
}

