// mpt_ordered.h — the default pipeline (MPT_PIPE_ORDERED): closest-first walk of the product's own 4-wide BVH
// (mpt_accel.h) inside the wave-local wavefront, with the answers of the reference's fixed-order walk.  gfx950 only.
//
// What the reference computes (R/Renderer/Shaders/PathTracing.h:75-204): because every child box lies inside its
// parent's (checked at upload), its stack walk equals a scan over the LEAVES in visit order — enter a leaf when its
// slab test passes with the best t so far, test its primitives in index order, keep strictly closer hits.  So it
// returns the primitive with the smallest t (first visited on ties) whenever that primitive's t is not below its own
// leaf's slab entry ("consistent": then no smaller best-t can have locked the leaf).  closest_hit_ordered() finds the
// smallest-t primitive with a closest-first walk and then runs the reference's exact slab test on the winner's
// REFERENCE leaf box.  A ray is handed to the reference-order walk (ring E, closest_hit_resume) when
//   * a direction component is (nearly) zero or not a number (slab arithmetic of the two walks differs there),
//   * two different primitives tie for the best t (the reference keeps the first it visits),
//   * the winner fails that final check (6.5e-4 of the rays on scene.xml: hits on the r = 10^4 ground sphere whose t
//     is wrong by up to 1e-3 — catastrophic cancellation — and lands in front of its own leaf box).
// Sub-trees are culled only when they start beyond best t * (1 + 2^-10) + eps_abs; own leaf boxes contain the
// reference leaf boxes (padded), so every primitive the reference could accept in front of the winner is seen here —
// EXCEPT a triangle whose computed t is an artefact: in front of the triangle's own box by more than that margin, which
// needs a ray within ~1e-5 / |e1 x e2| of the triangle's plane (|det| just above the reference's 1e-5 threshold, t = f *
// dot(e2, q) without significant digits).  Its box is culled by distance here; the reference computes the bogus t if it
// visits that leaf before the true hit.  The margin is EMPIRICAL, not a bound: a rule that covers those rays needs boxes
// inflated by ~0.06 |e1||e2| |o - v0| (a leaf size on bunny x20).  tests/experiments/ordered_proto.cpp replays
// the rule on the CPU against the oracle: 0 differences in 890 M rays of the headline render (9 of the flagged rays would
// have differed); the GPU tests compare whole renders bit for bit (0 differences in 1.3e11 rays of rendering) and aim
// 1e8 rays along the planes of slivers, needles and huge triangles (tests/test_gpu_adversarial.py: 6e-6 of THOSE differ,
// every one verified in exact arithmetic to be such an artefact of the reference).  include/mpt.h states this at the ABI.
//
// Pipeline shape (per persistent wave; rings as in k_wavelocal, but sorted by KIND of work instead of by trip budget, so
// that every expensive piece of code runs at full width):
//   primary step   64 new paths: primary rays in registers, TOP TEST = the always-list spheres + the root node's four
//                  boxes.  Misses and hits on those spheres are shaded at once; rays that touch a root child go to ring M0
//                  with their (best t, primitive) so far
//   ring R step    64 fresh bounce rays: the same
//   ring M_k step  64 rays that all have to walk the tree: closest-first walk (LDS stack) for the ring's budget of node-loop
//                  trips; unfinished walks are parked in ring M_k+1 with their state (next node, stack, best t /
//                  primitive): rays sorted by the walk they have behind them meet rays of similar length (lane
//                  utilisation of the walk 25 % -> 50 %).  Finished rays: final check, one bounce of shading, survivors -> R
//   ring E step    reference-order walk + shading for the flagged rays (6.5e-4 of the rays)
#pragma once
#include "mpt_device.h"
#include "mpt_kernels.h"

#define MPT_OT_LEAF 0x80000000u
#define MPT_OT_DONE 0xFFFFFFFFu
#define MPT_OT_KEY_MISS 0x7F800000u       // +inf: sorts behind every hit
// Operating point: five workgroups of 256 threads per CU = 5 waves/SIMD at 96 VGPRs (no scratch).  A workgroup's waves
// must spread evenly over the four SIMDs — 4 waves = one per SIMD — or a second workgroup does not fit beside the first:
// workgroups of 640 (10 waves: 3+3+2+2) "for 5 waves/SIMD" ran 40 % SLOWER than one workgroup of 1024, which is what
// earlier occupancy experiments had measured without knowing why.  bunny x20 256 spp: 1024 x 1 (4 waves/SIMD) 72.5 ms,
// 256 x 4 73.5, 256 x 5 65.5, 256 x 6 (80 VGPRs, 80 B of scratch) 65.3; the small workgroups stage less of the tree in
// LDS (30 KB each instead of 160), which costs nothing measurable on the big scenes this pipeline is for.
#ifndef MPT_OT_THREADS
#define MPT_OT_THREADS 256
#endif
#ifndef MPT_OT_WAVES
#define MPT_OT_WAVES 5                    // per SIMD
#endif
#define MPT_OT_WGS_PER_CU ((MPT_OT_WAVES * 256) / MPT_OT_THREADS)   // workgroups that share a CU's 160 KiB of LDS
#ifndef MPT_OT_MLEVELS
#define MPT_OT_MLEVELS 2u                 // tree-walk rings: rays sorted by the walk they have already done (budgets).
#endif                                    // bunny x20, 256 spp: 2 rings 76.8 ms, 3 rings 77.9, 4 rings 82.4
// Ring R holds HITS (round 4; as ring 0 of k_wavelocal, mpt_kernels.h): a step that has a ray's final closest hit pushes the hit (ray,
// t, primitive) to ring R instead of shading it at whatever width the step happens to have — the always-list sphere hits of a top
// test, the finished walks of a ring-M step, the re-traced rays of ring E — and the step that pops 64 hits shades them all at full
// width, then runs the top test on the 64 bounce rays.  Same number of ring hops, same record size.  (Rejected variants of the rings
// and of the node fetch — eight rings keyed by the direction octant, 64-byte quantised nodes, non-temporal pops and pushes, touch and
// prefetch loads — are recorded with their code and numbers in tests/experiments/rejected_r04_flags.h and rejected_r05.h.)
#define MPT_OT_NR 1u                      // fresh-ray rings
#define MPT_OT_RINGS (MPT_OT_NR + 1u + MPT_OT_MLEVELS) // R (x NR) fresh rays, E reference-order walk, M0.. rays walking the tree
#define MPT_OT_PARK 8u                    // stack entries a parked ray takes along (>= the LDS stack depth)
#ifndef MPT_OT_EARLY
#define MPT_OT_EARLY 2u                   // the node loop pauses when fewer than 1/EARLY of the lanes that entered it still search
#endif                                    // (bunny x20 at 4 waves/SIMD: 1/2 79.3 ms, 1/3 77.9, 1/4 77.9, 1/8 79.5, 1/16 83.8;
                                          //  at 5 waves/SIMD: 1/2 64.7, 2/3 66.3, 3/4 68.0, 1/4 65.7, 1/8 69.7)
#define MPT_OT_EARLY_NUM 1u

// Diagnostics build (-DMPT_OT_TIMES): shader-clock cycles per region of k_ordered, summed over all waves, plus step and
// lane counts per step kind (tools/gpu_ot_times.py).  Not compiled into the product library.
#ifdef MPT_OT_TIMES
#define OT_NREG 8   // select, fetch, top test, walk, final check, exact walk, shade, push
__device__ unsigned long long g_ot_walk[16];    // closest-first walk: node-loop cycles, leaf-loop cycles, node trips, leaf trips, rounds, node lane-trips, leaf lane-trips,
                                                // [7..10] node visits served from LDS / from global memory, primitive records loaded from LDS / from global memory (per lane),
                                                // [11..14] stack pops: calls (lanes), entries examined (lanes), calls (waves), loop trips (waves)
__device__ unsigned long long g_ot_times[OT_NREG + 16];   // + steps[RINGS + 1] at 8, lanes[RINGS + 1] at 16
#define OT_TIC() unsigned long long ot_t_ = __builtin_amdgcn_s_memtime()
#define OT_TOC(r)                                                   \
    do {                                                            \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        ot_acc[r] += now_ - ot_t_;                                  \
        ot_t_ = now_;                                               \
    } while (0)
__device__ __forceinline__ void ot_flush_walk_times(const WorkCount& wc, uint32_t lane) {
    unsigned long long v[15] = {wc.ot_node_cycles, wc.ot_leaf_cycles, wc.ot_node_trips, wc.ot_leaf_trips, wc.ot_rounds, wc.ot_node_lanes, wc.ot_leaf_lanes,
                                wc.ot_lds_nodes, wc.ot_glb_nodes, wc.ot_lds_prims, wc.ot_glb_prims, wc.ot_pops, wc.ot_pop_iters, wc.ot_pop_calls, wc.ot_pop_wave_iters};
    for (int k = 0; k < 15; ++k) {
        if (k < 2 || k == 4) v[k] = __shfl(v[k], 0);   // cycles / rounds are per wave: lane 0's copy
        else for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off);
        if (lane == 0) atomicAdd(&g_ot_walk[k], v[k]);
    }
}
#else
#define OT_TIC() do { } while (0)
#define OT_TOC(r) do { } while (0)
#endif

struct OtRings {              // [n_waves][MPT_OT_RINGS][MPT_WL_RING] records, struct of arrays of 16-byte fields; ONE allocation (one base
    float4* base;             // pointer + the array length in scalar registers instead of nine pointers)
    uint32_t n;               // records per array
    __host__ __device__ float4* od() const { return base; }                               // (o.xyz, d.x)
    __host__ __device__ float4* dt() const { return base + n; }                           // (d.y, d.z, thr.r, thr.g)
    __host__ __device__ float4* tl() const { return base + 2u * (size_t)n; }              // as WaveRings (mpt_kernels.h): (L.rgb, L.a), only for rays that have gathered light
    __host__ __device__ uint4* ia() const { return (uint4*)(base + 3u * (size_t)n); }     // ... and (thr.b bits, path, pixel, bounce | MPT_RING_HAS_LIGHT)
    __host__ __device__ uint4* tv() const { return (uint4*)(base + 4u * (size_t)n); }     // rings M: (best t bits, best primitive, next node / leaf of the walk, stack entries)
    __host__ __device__ uint4* sk(uint32_t k) const { return (uint4*)(base + (5u + k) * (size_t)n); }  // rings M: the walk's stack, two (key, ref) entries per field
};
#define MPT_OT_RING_ARRAYS (5u + MPT_OT_PARK / 2u)
struct OtBudgets {
    uint32_t trips[MPT_OT_MLEVELS];       // node-loop trips a step of ring M_k may make (the last level: unlimited)
    uint32_t min_active[MPT_OT_MLEVELS];  // ... and it ends once fewer lanes than this are still walking
    uint32_t inplace_min;                 // a primary / ring-R step whose top test sends at least this many lanes into the tree
};                                        // walks it at once (as a ring-M0 step would) instead of parking them; 65 = never

struct AccelDev {
    const float4* nodes;    // MPT_OT_NODE_STRIDE float4 per node (7 used), breadth-first (mpt_accel.h)
    const float4* refleaf;  // 2 float4 per reference leaf: (bmin, 0) (bmax, 0)
    const float4* refbox;   // the same boxes per PRIMITIVE (2 float4 each; k_prim_refbox, mpt_devbuild.h): what the final check reads
    const float4* always;   // 5 float4 per sphere of the always list: (c, leaf<<1) (r, bits(index), bits(k), mat) (0,0,0, orig id)
                            // + the box of its reference leaf (bmin, 0) (bmax, 0)
    uint32_t n_nodes, n_lds_nodes, n_always;
    uint32_t lds_always_off;  // float4 index of the always list in LDS
    uint32_t lds_stack_off;   // byte offset of the stacks in LDS
    uint32_t stack_depth;     // LDS stack entries per lane
    float eps_abs, cull_rel;  // sub-trees are culled beyond best t * (1 + cull_rel) + eps_abs (cull_rel = 2^-10 unless MPT_OT_CULL_REL says otherwise)
    float o_limit;            // ray origins farther out than this (64 x the triangle extent) exceed what the box padding covers
};

typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) v2u* LdsStack;

// Per-lane stack in LDS: entry e of a lane at lds[e * 64] (entry-major: the 64 lanes of a wave hit 64 different bank
// pairs whatever their depths).  entry = (entry-distance key, child reference).  When a node's children do not all fit,
// the FARTHEST of them are the ones dropped (ot_push_sorted) and the walk is marked incomplete: the caller walks the tree
// once more from the root with the best t found so far — nearly everything is culled then, so that second pass is
// short (1.5e-5 of the rays on scene.xml, 0.7 % on bunny x20 with 8 entries).
struct OtStack {
    LdsStack lds;
    uint32_t depth;
};
// k1 <= k2 <= k3: the keys of a node's children behind the nearest one (KEY_MISS = not hit; hits come first).  Pushed
// farthest first, so that the nearest is popped first; with `room` free entries only the nearest `room` of them go in.
__device__ __forceinline__ void ot_push_sorted(const OtStack& st, uint32_t& sp, const uint4& ref, uint32_t k1, uint32_t k2,
                                               uint32_t k3, bool& lost);
// Near / far planes by ADDRESS (round 5, MPT_OT_SIGNSEL): a node stores lo.x[4] lo.y[4] lo.z[4] hi.x[4] hi.y[4] hi.z[4]; which of a slab's
// two planes the ray meets first is the sign of 1/d — per ray, not per node.  Instead of computing both distances and taking
// min / max (3 + 3 instructions per child box), a lane whose direction is negative on an axis swaps the two LOAD ADDRESSES of that axis
// (a byte offset of 0 or 48 added to the node's address: 6 additions per node), and the distances come out as near / far:
// t_near = max3, t_far = min3.  fma(b, 1/d, -o/d) is monotone in b and lo <= hi, so near = min(t_lo, t_hi) and far = max(...) EXACTLY:
// the keys, and with them the walk, are bit for bit what they were (24 vector instructions less per node visit).
#ifndef MPT_OT_SIGNSEL
#define MPT_OT_SIGNSEL 1
#endif
struct OtRay {
    float idx, idy, idz, ox, oy, oz;  // approximate 1/d and o/d: own boxes are padded for it (mpt_hip.hip)
#if MPT_OT_SIGNSEL
    uint32_t sx, sy, sz;              // 48 where the direction is negative (the hi plane is the near one), else 0: byte offsets into a node
#endif
};
__device__ __forceinline__ OtRay ot_ray(F3 o, F3 d) {
    OtRay r;
    r.idx = __builtin_amdgcn_rcpf(d.x);
    r.idy = __builtin_amdgcn_rcpf(d.y);
    r.idz = __builtin_amdgcn_rcpf(d.z);
    r.ox = o.x * r.idx;
    r.oy = o.y * r.idy;
    r.oz = o.z * r.idz;
#if MPT_OT_SIGNSEL
    r.sx = r.idx < 0.0f ? 48u : 0u;
    r.sy = r.idy < 0.0f ? 48u : 0u;
    r.sz = r.idz < 0.0f ? 48u : 0u;
#endif
    return r;
}
// rays the closest-first walk does not take: a direction component that is (nearly) zero, or anything not finite
__device__ __forceinline__ bool ot_degenerate(F3 o, F3 d, float o_limit) {
    const float tiny = 9.5367431640625e-07f;  // 2^-20
    return !(fabsf(d.x) >= tiny && fabsf(d.y) >= tiny && fabsf(d.z) >= tiny && fabsf(d.x) <= 2.0f && fabsf(d.y) <= 2.0f &&
             fabsf(d.z) <= 2.0f && fabsf(o.x) <= o_limit && fabsf(o.y) <= o_limit && fabsf(o.z) <= o_limit);
}

struct OtNode {
    float4 lx, ly, lz, hx, hy, hz;   // MPT_OT_SIGNSEL: l* = the planes the ray meets first, h* = the ones it leaves through
    uint4 ref;
};
#if MPT_OT_SIGNSEL
typedef const __attribute__((address_space(3))) char* LdsBytes;
template <bool ALL_LDS>
__device__ __forceinline__ OtNode ot_load_node(const AccelDev& ac, LdsNodes lds, uint32_t n, const OtRay& r) {
    static_assert(MPT_OT_NODE_STRIDE == 7u, "MPT_OT_SIGNSEL addresses the 112-byte float node");
    OtNode nd;
    const uint32_t at = __umul24(n, 112u);   // (n < 2^24: checked where the launch is sized, ordered_layout_ok)
    // near planes at `at + s`, far planes at `(at + 48) - s`, the axis in the instruction's immediate offset (0 / 16 / 32): three
    // registers per ray (sx, sy, sz), seven additions per node.  MPT_OT_SIGNSEL=2: six per-ray offsets, six additions (more registers).
    v4f a, b, c, d, e, f, g;
#if MPT_OT_SIGNSEL == 2
    const uint32_t fx = 48u - r.sx, fy = 64u - r.sy, fz = 80u - r.sz, ny = 16u + r.sy, nz = 32u + r.sz;
    if (ALL_LDS || n < ac.n_lds_nodes) {
        const LdsBytes q = (LdsBytes)lds + at;
        a = *(LdsNodes)(q + r.sx);
        b = *(LdsNodes)(q + ny);
        c = *(LdsNodes)(q + nz);
        d = *(LdsNodes)(q + fx);
        e = *(LdsNodes)(q + fy);
        f = *(LdsNodes)(q + fz);
        g = *(LdsNodes)(q + 96u);
    } else {
        const char* q = (const char*)ac.nodes;   // (32-bit offsets from the uniform base: global_load ... v_off, s[base:base+1])
        a = *(const v4f*)(q + (at + r.sx));
        b = *(const v4f*)(q + (at + ny));
        c = *(const v4f*)(q + (at + nz));
        d = *(const v4f*)(q + (at + fx));
        e = *(const v4f*)(q + (at + fy));
        f = *(const v4f*)(q + (at + fz));
        g = *(const v4f*)(q + (at + 96u));
    }
#else
    uint32_t at48 = at + 48u;
    asm volatile("" : "+v"(at48));   // (kept as one value: the compiler would otherwise fold the 48 into six per-ray constants and hold six registers)
    if (ALL_LDS || n < ac.n_lds_nodes) {
        const LdsBytes qn = (LdsBytes)lds + at, qf = (LdsBytes)lds + at48;
        a = *(LdsNodes)(qn + r.sx);
        b = *(LdsNodes)(qn + r.sy + 16u);
        c = *(LdsNodes)(qn + r.sz + 32u);
        d = *(LdsNodes)(qf - r.sx);
        e = *(LdsNodes)(qf - r.sy + 16u);
        f = *(LdsNodes)(qf - r.sz + 32u);
        g = *(LdsNodes)(qn + 96u);
    } else {
        const char* q = (const char*)ac.nodes;   // (32-bit offsets from the uniform base: global_load ... v_off, s[base:base+1] offset:imm)
        a = *(const v4f*)(q + (at + r.sx));
        b = *(const v4f*)(q + (at + r.sy) + 16);
        c = *(const v4f*)(q + (at + r.sz) + 32);
        d = *(const v4f*)(q + (at48 - r.sx));
        e = *(const v4f*)(q + (at48 - r.sy) + 16);
        f = *(const v4f*)(q + (at48 - r.sz) + 32);
        g = *(const v4f*)(q + at + 96);
    }
#endif
    nd.lx = make_float4(a.x, a.y, a.z, a.w);
    nd.ly = make_float4(b.x, b.y, b.z, b.w);
    nd.lz = make_float4(c.x, c.y, c.z, c.w);
    nd.hx = make_float4(d.x, d.y, d.z, d.w);
    nd.hy = make_float4(e.x, e.y, e.z, e.w);
    nd.hz = make_float4(f.x, f.y, f.z, f.w);
    nd.ref = make_uint4(__float_as_uint(g.x), __float_as_uint(g.y), __float_as_uint(g.z), __float_as_uint(g.w));
    return nd;
}
#else
template <bool ALL_LDS>
__device__ __forceinline__ OtNode ot_load_node(const AccelDev& ac, LdsNodes lds, uint32_t n, const OtRay&) {
    OtNode nd;
    if (ALL_LDS || n < ac.n_lds_nodes) {
        const LdsNodes q = lds + 7u * n;
        const v4f a = q[0], b = q[1], c = q[2], d = q[3], e = q[4], f = q[5], g = q[6];
        nd.lx = make_float4(a.x, a.y, a.z, a.w);
        nd.ly = make_float4(b.x, b.y, b.z, b.w);
        nd.lz = make_float4(c.x, c.y, c.z, c.w);
        nd.hx = make_float4(d.x, d.y, d.z, d.w);
        nd.hy = make_float4(e.x, e.y, e.z, e.w);
        nd.hz = make_float4(f.x, f.y, f.z, f.w);
        nd.ref = make_uint4(__float_as_uint(g.x), __float_as_uint(g.y), __float_as_uint(g.z), __float_as_uint(g.w));
    } else {
        const float4* q = ac.nodes + MPT_OT_NODE_STRIDE * (size_t)n;
        nd.lx = q[0];
        nd.ly = q[1];
        nd.lz = q[2];
        nd.hx = q[3];
        nd.hy = q[4];
        nd.hz = q[5];
        const float4 g = q[6];
        nd.ref = make_uint4(__float_as_uint(g.x), __float_as_uint(g.y), __float_as_uint(g.z), __float_as_uint(g.w));
    }
    return nd;
}
#endif
// entry distance of the ray into one child box as a sort key: float bits with the child slot in the two low bits
// (t >= 0, so unsigned order = float order; the key rounds the distance DOWN by at most 3 ulp), or KEY_MISS
__device__ __forceinline__ uint32_t ot_box_key(const OtRay& r, float lx, float ly, float lz, float hx, float hy, float hz,
                                               uint32_t ref, float lim, uint32_t slot) {
#if MPT_OT_SIGNSEL   // (lx .. lz are the near planes, hx .. hz the far ones: ot_load_node)
    const float tn = fmaxf(fmaxf(fmaf(lx, r.idx, -r.ox), fmaf(ly, r.idy, -r.oy)), fmaxf(fmaf(lz, r.idz, -r.oz), 0.0f));
    const float tf = fminf(fminf(fmaf(hx, r.idx, -r.ox), fmaf(hy, r.idy, -r.oy)), fmaf(hz, r.idz, -r.oz));
#else
    float t0 = fmaf(lx, r.idx, -r.ox), t1 = fmaf(hx, r.idx, -r.ox);
    float tn = fminf(t0, t1), tf = fmaxf(t0, t1);
    t0 = fmaf(ly, r.idy, -r.oy);
    t1 = fmaf(hy, r.idy, -r.oy);
    tn = fmaxf(tn, fminf(t0, t1));
    tf = fminf(tf, fmaxf(t0, t1));
    t0 = fmaf(lz, r.idz, -r.oz);
    t1 = fmaf(hz, r.idz, -r.oz);
    tn = fmaxf(fmaxf(tn, fminf(t0, t1)), 0.0f);
    tf = fminf(tf, fmaxf(t0, t1));
#endif
    // one comparison: tn <= tf * (1 + 2^-21) and tn <= lim (no NaN here: planes and 1/d are finite for the rays the walk
    // takes, or +inf for the x planes of an empty slot, which no ray enters — mpt_accel.h emit)
    (void)ref;
    const bool hit = tn <= fminf(tf * 1.00000048f, lim);
    return hit ? ((__float_as_uint(tn) & ~3u) | slot) : (MPT_OT_KEY_MISS | slot);
}
__device__ __forceinline__ void ot_sort2(uint32_t& a, uint32_t& b) {
    const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
    a = lo;
    b = hi;
}
__device__ __forceinline__ float ot_cull_limit(float T, const AccelDev& ac) { return T + (T * ac.cull_rel + ac.eps_abs); }

// One primitive against the ray — the reference's tests, PathTracing.h:120-176, bit for bit (as leaf_test in
// mpt_device.h), plus the tie flag.  `index` = position in the device primitive array.
__device__ __forceinline__ void ot_test_prim(const Prim3& pr, uint32_t index, F3 o, F3 d, float& T, int& W, bool& tie) {
    const float4 p0 = pr.p0, p1 = pr.p1, p2 = pr.p2;
    float tt = 0.0f;
    bool hit = false;
    if (prim_type(p0) == 1) {
        // Triangles are tested in the leaf loop, 64 different ones per wave: some lane nearly always needs every stage, so
        // the reference's nested ifs would only cost exec-mask bookkeeping (a dozen scalar instructions per test) and
        // serialise the division behind the first comparison.  Written without early-outs, the same operations produce the
        // same values; where the reference leaves early the rest is computed from garbage (possibly NaN / Inf) and
        // discarded by `hit` (bunny x20: 112.0 -> 109.9 ms).
        const F3 v0 = f3(p0.x, p0.y, p0.z), e1 = f3(p1.x, p1.y, p1.z), e2 = f3(p2.x, p2.y, p2.z);
        const F3 h = cross3(d, e2);
        const float a = dot3(e1, h);
        const float f = mpt_rcp(a);
        const F3 s = o - v0;
        const float u = f * dot3(s, h);
        const F3 q = cross3(s, e1);
        const float v = f * dot3(d, q);
        tt = f * dot3(e2, q);
        hit = fabsf(a) > 1e-5f && u >= 0.0f && u <= 1.0f && v >= 0.0f && u + v <= 1.0f && tt > 0.0001f;
    } else {
        // Spheres are tested by the top test, the same sphere in all 64 lanes: whole waves miss it and skip the square
        // root and the division (scene.xml: 27.9 -> 26.8 ms against the version without early-outs)
        const F3 c = f3(p0.x, p0.y, p0.z);
        const float radius = p1.x;
        const F3 oc = o - c;
        const float a = dot3(d, d);
        const float b = dot3(oc, d);
        // b >= 0: the centre lies behind the ray.  Then -b <= 0 and sqrt >= 0, so the root (-b - sqrt(disc)) / a is <= 0
        // and fails "> 0.0001" whatever disc is — signs are exact in floating point, so skipping the square root and
        // the division here changes nothing (a NaN b compares false and takes the full path).  Bounce rays leaving the
        // ground and every ray looking away from a sphere take this exit, usually as a whole wave.
        if (!(b >= 0.0f)) {
            const float cc = dot3(oc, oc) - radius * radius;
            const float disc = b * b - a * cc;
            if (disc > 0.0f) {
                const float sq = sqrtf(disc);
                tt = (-b - sq) / a;
                hit = tt > 0.0001f;
            }
        }
    }
    if (hit) {
        if (tt < T) {
            T = tt;
            W = (int)index;
        } else if (tt == T && (int)index != W) {
            tie = true;
        }
    }
}

// TOP TEST: the always-list spheres, then "does the ray touch any child of the root?" (need).
template <bool COUNT>
__device__ __forceinline__ void ot_top_test(const AccelDev& ac, LdsNodes lds, F3 o, F3 d, const OtRay& r, float& T, int& W,
                                            bool& tie, bool& need, WorkCount& wc) {
    for (uint32_t k = 0; k < ac.n_always; ++k) {
        const LdsNodes q = lds + ac.lds_always_off + 5u * k;
        const v4f a = q[0], b = q[1], c = q[2];
        Prim3 pr;
        pr.p0 = make_float4(a.x, a.y, a.z, a.w);
        pr.p1 = make_float4(b.x, b.y, b.z, b.w);
        pr.p2 = make_float4(c.x, c.y, c.z, c.w);
        if (COUNT) wc.prim_tests++;
        ot_test_prim(pr, __float_as_uint(b.y), o, d, T, W, tie);
    }
    const OtNode nd = ot_load_node<true>(ac, lds, 0u, r);   // the root is always staged
    const float lim = ot_cull_limit(T, ac);
    const uint32_t k0 = ot_box_key(r, nd.lx.x, nd.ly.x, nd.lz.x, nd.hx.x, nd.hy.x, nd.hz.x, nd.ref.x, lim, 0u);
    const uint32_t k1 = ot_box_key(r, nd.lx.y, nd.ly.y, nd.lz.y, nd.hx.y, nd.hy.y, nd.hz.y, nd.ref.y, lim, 1u);
    const uint32_t k2 = ot_box_key(r, nd.lx.z, nd.ly.z, nd.lz.z, nd.hx.z, nd.hy.z, nd.hz.z, nd.ref.z, lim, 2u);
    const uint32_t k3 = ot_box_key(r, nd.lx.w, nd.ly.w, nd.lz.w, nd.hx.w, nd.hy.w, nd.hz.w, nd.ref.w, lim, 3u);
    uint32_t m = k0 < k1 ? k0 : k1;
    m = m < k2 ? m : k2;
    m = m < k3 ? m : k3;
    need = m < MPT_OT_KEY_MISS;
    if (COUNT) wc.node_visits++;
}

// Closest-first walk ("while-while": every lane walks nodes until it holds a leaf, then the wave tests leaves together).
// in/out T, W (best t / primitive so far, e.g. from the top test); tie / overflow are only ever set.
__device__ __forceinline__ uint32_t ot_pick(const uint4& ref, uint32_t key) {  // two levels of v_cndmask, no branches
    const bool odd = (key & 1u) != 0u, high = (key & 2u) != 0u;
    const uint32_t a = odd ? ref.y : ref.x, b = odd ? ref.w : ref.z;
    return high ? b : a;
}
// Round 5, two changes that take dependent LDS round trips and exec-mask bookkeeping out of a walk trip (MPT_OT_STACK2, measured on
// bunny x20 with the diagnostics build: a wave-level pop made 1.95 trips of its loop — one LDS round trip each — and 78 % of all
// node / leaf trips end in one):
//  * the push is branch-free when three entries fit: the keys are sorted and hits come first, so the three children are written
//    farthest first at sp, sp + [k3 hit], sp + [k3 hit] + [k2 hit] — an entry that is not a hit is overwritten by the next write
//    or stays above the new top, where nothing reads it (three unconditional ds_write_b64 instead of three predicated regions);
//  * the pop reads the TWO topmost entries with one instruction (entries of a lane are 512 bytes apart: ds_read2st64_b64).
#ifndef MPT_OT_STACK2
#define MPT_OT_STACK2 1
#endif
__device__ __forceinline__ void ot_push_sorted(const OtStack& st, uint32_t& sp, const uint4& ref, uint32_t k1, uint32_t k2,
                                               uint32_t k3, bool& lost) {
    const uint32_t room = st.depth - sp;
    const bool h1 = k1 < MPT_OT_KEY_MISS, h2 = k2 < MPT_OT_KEY_MISS, h3 = k3 < MPT_OT_KEY_MISS;
#if MPT_OT_STACK2
    if (room >= 3u) {   // (nearly always: the stack is 8 deep)
        const uint32_t p2 = sp + (h3 ? 1u : 0u), p1 = p2 + (h2 ? 1u : 0u);
        st.lds[sp * 64u] = v2u{k3, ot_pick(ref, k3)};
        st.lds[p2 * 64u] = v2u{k2, ot_pick(ref, k2)};
        st.lds[p1 * 64u] = v2u{k1, ot_pick(ref, k1)};
        sp = p1 + (h1 ? 1u : 0u);
        return;
    }
#endif
    if (h3 && room >= 3u) st.lds[sp++ * 64u] = v2u{k3, ot_pick(ref, k3)};
    if (h2 && room >= 2u) st.lds[sp++ * 64u] = v2u{k2, ot_pick(ref, k2)};
    if (h1 && room >= 1u) st.lds[sp++ * 64u] = v2u{k1, ot_pick(ref, k1)};
    lost = lost || (uint32_t)h1 + (uint32_t)h2 + (uint32_t)h3 > room;
}
__device__ __forceinline__ uint32_t ot_pop_next(const OtStack& st, uint32_t& sp, float lim, WorkCount& wc) {
#ifdef MPT_OT_TIMES
    wc.ot_pops++;
    if (first_active_lane()) wc.ot_pop_calls++;
#endif
#if MPT_OT_STACK2
    while (sp > 0u) {
        // entries sp - 2 and sp - 1 (with one entry left: 0 and 1, of which 1 is stale and not looked at)
        const uint32_t b = (sp > 2u ? sp : 2u) - 2u;
        const v2u lo = st.lds[b * 64u], hi = st.lds[(b + 1u) * 64u];
#ifdef MPT_OT_TIMES
        wc.ot_pop_iters++;
        if (first_active_lane()) wc.ot_pop_wave_iters++;
#endif
        const bool two = sp >= 2u;
        const v2u top = two ? hi : lo, below = lo;
        if (__uint_as_float(top.x & ~3u) <= lim) {
            sp -= 1u;
            return top.y;
        }
        if (two && __uint_as_float(below.x & ~3u) <= lim) {
            sp -= 2u;
            return below.y;
        }
        sp = two ? sp - 2u : 0u;
    }
#else
    while (sp > 0u) {
        --sp;
        const v2u e = st.lds[sp * 64u];
#ifdef MPT_OT_TIMES
        wc.ot_pop_iters++;
        if (first_active_lane()) wc.ot_pop_wave_iters++;
#endif
        if (__uint_as_float(e.x & ~3u) <= lim) return e.y;
    }
#endif
    (void)wc;
    return MPT_OT_DONE;
}
// Resumable: (cur, sp) and the lane's stack are the walk's state.  BUDGETED: the step ends when the wave has made
// `budget` trips of the node loop, or fewer than `min_active` lanes still walk; lanes that are not finished then keep
// their state (the caller parks them in the next ring, where they meet rays with as long a walk behind them).
// `overflow` is set when a stack entry had to be dropped: the caller repeats the walk from the root with the T found.
// Returns true when this lane's walk is complete.
template <bool COUNT, bool BUDGETED, bool ALL_LDS>
__device__ __forceinline__ bool ot_walk(const AccelDev& ac, const SceneDev& sc, LdsNodes lds, const OtStack& st, F3 o, F3 d,
                                        const OtRay& r, uint32_t& cur, uint32_t& sp, float& T, int& W, bool& tie,
                                        bool& overflow, uint32_t budget, uint32_t min_active, WorkCount& wc) {
    uint32_t trips = 0;
#ifdef MPT_OT_TIMES
    unsigned long long wt_ = __builtin_amdgcn_s_memtime();
#define OT_WTOC(field)                                                \
    do {                                                              \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        wc.field += now_ - wt_;                                       \
        wt_ = now_;                                                   \
    } while (0)
#else
#define OT_WTOC(field) do { } while (0)
#endif
    for (;;) {
#ifdef MPT_OT_TIMES
        wc.ot_rounds++;
#endif
        // The node loop is wave-uniform in its CONTROL (round 4): the wave goes round while enough of its lanes search, a lane that
        // has found its leaf sits the trip out under the exec mask.  So `trips` — the wave's count, which a lane that left early has
        // to adopt — is one scalar register; with per-lane loop exits (round 3) it had to be maximised over the lanes after every
        // round: six dependent ds_bpermute + waits, ~600 cycles, three node trips apart.
        const uint32_t n_entered = (uint32_t)__popcll(__ballot(cur < MPT_OT_LEAF)) * MPT_OT_EARLY_NUM;
        for (;;) {
            if (BUDGETED && trips >= budget) break;
            // few lanes still searching, the others hold a leaf: test the leaves now, the search resumes afterwards
            const uint32_t n_search = (uint32_t)__popcll(__ballot(cur < MPT_OT_LEAF));
            if (n_search == 0u || n_search * MPT_OT_EARLY < n_entered) break;
            if (BUDGETED) ++trips;
            if (!(cur < MPT_OT_LEAF)) continue;
            // (taking ONE source per trip for the whole wave — LDS only when every searching lane is at a staged node —
            // instead of a per-lane choice was measured on bunny x20: no difference, 20.7 ms either way)
            const OtNode nd = ot_load_node<ALL_LDS>(ac, lds, cur, r);
            const float lim = ot_cull_limit(T, ac);
            uint32_t k0 = ot_box_key(r, nd.lx.x, nd.ly.x, nd.lz.x, nd.hx.x, nd.hy.x, nd.hz.x, nd.ref.x, lim, 0u);
            uint32_t k1 = ot_box_key(r, nd.lx.y, nd.ly.y, nd.lz.y, nd.hx.y, nd.hy.y, nd.hz.y, nd.ref.y, lim, 1u);
            uint32_t k2 = ot_box_key(r, nd.lx.z, nd.ly.z, nd.lz.z, nd.hx.z, nd.hy.z, nd.hz.z, nd.ref.z, lim, 2u);
            uint32_t k3 = ot_box_key(r, nd.lx.w, nd.ly.w, nd.lz.w, nd.hx.w, nd.hy.w, nd.hz.w, nd.ref.w, lim, 3u);
            if (COUNT) {
                wc.node_visits++;
                wc.aabb_hits += (k0 < MPT_OT_KEY_MISS) + (k1 < MPT_OT_KEY_MISS) + (k2 < MPT_OT_KEY_MISS) + (k3 < MPT_OT_KEY_MISS);
                if (first_active_lane()) wc.node_iters++;
            }
#ifdef MPT_OT_TIMES
            if (first_active_lane()) wc.ot_node_trips++;
            wc.ot_node_lanes++;
            if (ALL_LDS || cur < ac.n_lds_nodes) wc.ot_lds_nodes++;
            else wc.ot_glb_nodes++;
#endif
            ot_sort2(k0, k1);
            ot_sort2(k2, k3);
            ot_sort2(k0, k2);
            ot_sort2(k1, k3);
            ot_sort2(k1, k2);
            if (k0 < MPT_OT_KEY_MISS) {
                cur = ot_pick(nd.ref, k0);
                if (k1 < MPT_OT_KEY_MISS) ot_push_sorted(st, sp, nd.ref, k1, k2, k3, overflow);
            } else {
                cur = ot_pop_next(st, sp, lim, wc);
            }
        }
        OT_WTOC(ot_node_cycles);
        if (cur != MPT_OT_DONE && cur >= MPT_OT_LEAF) {  // a leaf: primitives [first, first + count) in index order
            const uint32_t first = cur & 0x07FFFFFFu, count = ((cur >> 27) & 15u) + 1u;
            if (COUNT && first_active_lane()) wc.outer_iters++;
            // The first two primitives of the leaf are loaded TOGETHER (one memory round trip, not two: the walk is a chain of
            // latencies, and the big scenes' leaves hold at most two), then tested in index order as before.
            uint32_t k_from = 0u;
            {
                const bool two = count > 1u;
                const Prim3 pa = load_prim(sc, lds, first);
                Prim3 pb = pa;
                if (two) pb = load_prim(sc, lds, first + 1u);
#ifdef MPT_OT_TIMES
                if (first_active_lane()) wc.ot_leaf_trips++;
                wc.ot_leaf_lanes++;
                if (first < sc.n_lds_prims) wc.ot_lds_prims++;
                else wc.ot_glb_prims++;
                if (two) {
                    if (first + 1u < sc.n_lds_prims) wc.ot_lds_prims++;
                    else wc.ot_glb_prims++;
                }
#endif
                if (COUNT && first_active_lane()) wc.prim_iters++;
                if (!(ac.n_always != 0u && prim_type(pa.p0) == 0)) {
                    if (COUNT) wc.prim_tests++;
                    ot_test_prim(pa, first, o, d, T, W, tie);
                }
                if (two) {
                    if (COUNT && first_active_lane()) wc.prim_iters++;
                    if (!(ac.n_always != 0u && prim_type(pb.p0) == 0)) {
                        if (COUNT) wc.prim_tests++;
                        ot_test_prim(pb, first + 1u, o, d, T, W, tie);
                    }
                }
                k_from = 2u;
            }
            for (uint32_t k = k_from; k < count; ++k) {
                const Prim3 pr = load_prim(sc, lds, first + k);
                if (COUNT) {
                    if (first_active_lane()) wc.prim_iters++;
                }
#ifdef MPT_OT_TIMES
                if (first_active_lane()) wc.ot_leaf_trips++;
                wc.ot_leaf_lanes++;
                if (first + k < sc.n_lds_prims) wc.ot_lds_prims++;
                else wc.ot_glb_prims++;
#endif
                if (ac.n_always != 0u && prim_type(pr.p0) == 0) continue;  // spheres are on the always list
                if (COUNT) wc.prim_tests++;
                ot_test_prim(pr, first + k, o, d, T, W, tie);
            }
            cur = ot_pop_next(st, sp, ot_cull_limit(T, ac), wc);
        }
        OT_WTOC(ot_leaf_cycles);
        const unsigned long long going = __ballot(cur != MPT_OT_DONE && (!BUDGETED || trips < budget));
        if (going == 0ull) break;
        if (BUDGETED && (uint32_t)__popcll(going) < min_active) break;
    }
    return cur == MPT_OT_DONE;
}

// Final check: the reference's slab test (PathTracing.h:52-72, exact arithmetic, best t = +inf) on the winner's
// REFERENCE leaf box must pass, and the winner's t must not lie in front of that box.  The box is first tested with the
// walk's reciprocal arithmetic and an error bound ((|t| + |o/d|) * 2^-20 covers v_rcp_f32's 1 ulp, the rounding of o/d
// and of the fma, with a factor 2 to spare): when the answer is certain, the three IEEE divisions of the exact test are
// skipped — they are needed only for winners within that bound of their box (hits on the ground sphere next to the origin).
__device__ __forceinline__ bool ot_final_check(const AccelDev& ac, const SceneDev& sc, LdsNodes lds, F3 o, F3 d, const OtRay& r,
                                               float T, int W) {
    // the box of the winner's reference leaf, by primitive: two loads, one round trip (sc, lds: not needed)
    const float4 n0 = ac.refbox[2u * (size_t)(uint32_t)W], n1 = ac.refbox[2u * (size_t)(uint32_t)W + 1u];
    (void)sc;
    (void)lds;
    {
        float t0 = fmaf(n0.x, r.idx, -r.ox), t1 = fmaf(n1.x, r.idx, -r.ox);
        float lo = fminf(t0, t1), hi = fmaxf(t0, t1), m = fmaxf(fabsf(t0), fabsf(t1));
        t0 = fmaf(n0.y, r.idy, -r.oy);
        t1 = fmaf(n1.y, r.idy, -r.oy);
        lo = fmaxf(lo, fminf(t0, t1));
        hi = fminf(hi, fmaxf(t0, t1));
        m = fmaxf(m, fmaxf(fabsf(t0), fabsf(t1)));
        t0 = fmaf(n0.z, r.idz, -r.oz);
        t1 = fmaf(n1.z, r.idz, -r.oz);
        lo = fmaxf(fmaxf(lo, fminf(t0, t1)), 0.0001f);
        hi = fminf(hi, fmaxf(t0, t1));
        m = fmaxf(m, fmaxf(fabsf(t0), fabsf(t1)));
        const float err = (m + fmaxf(fabsf(r.ox), fmaxf(fabsf(r.oy), fabsf(r.oz)))) * 9.5367431640625e-07f;
        if (hi - err > lo + err && T >= lo + err) return true;  // (a NaN anywhere fails this and takes the exact test)
    }
    const float idx = mpt_rcp(d.x), idy = mpt_rcp(d.y), idz = mpt_rcp(d.z);
    float t0 = (n0.x - o.x) * idx, t1 = (n1.x - o.x) * idx;
    float lo = fmaxf(0.0001f, idx < 0.0f ? t1 : t0);
    float hi = idx < 0.0f ? t0 : t1;
    t0 = (n0.y - o.y) * idy;
    t1 = (n1.y - o.y) * idy;
    lo = fmaxf(lo, idy < 0.0f ? t1 : t0);
    hi = fminf(hi, idy < 0.0f ? t0 : t1);
    t0 = (n0.z - o.z) * idz;
    t1 = (n1.z - o.z) * idz;
    lo = fmaxf(lo, idz < 0.0f ? t1 : t0);
    hi = fminf(hi, idz < 0.0f ? t0 : t1);
    return hi > lo && T >= lo;
}

// Closest hit with the reference's answer, for one wave of rays outside the pipeline (unit-test kernel): walk, check,
// and the reference-order walk for the flagged lanes.  `sc` = reference-order scene (threaded nodes in global memory).
template <bool COUNT>
__device__ __forceinline__ void closest_hit_ordered(const AccelDev& ac, const SceneDev& sc, LdsNodes lds, const OtStack& st,
                                                    F3 o, F3 d, bool valid, float& T, int& W, uint32_t& flags, WorkCount& wc) {
    T = INFINITY;
    W = -1;
    flags = 0;
    bool tie = false, overflow = false, need = false;
    const bool degenerate = ot_degenerate(o, d, ac.o_limit);
    if (degenerate) flags |= 1u;
    const OtRay r = ot_ray(o, d);
    if (valid && !degenerate) ot_top_test<COUNT>(ac, lds, o, d, r, T, W, tie, need, wc);
    uint32_t cur = valid && need && !degenerate ? 0u : MPT_OT_DONE, sp = 0;
    ot_walk<COUNT, false, false>(ac, sc, lds, st, o, d, r, cur, sp, T, W, tie, overflow, 0xFFFFFFFFu, 0u, wc);
    for (int pass = 0; pass < 4 && __ballot(overflow) != 0ull; ++pass) {  // entries were dropped: again, with the T found
        cur = overflow ? 0u : MPT_OT_DONE;
        sp = 0;
        bool again = false;
        ot_walk<COUNT, false, false>(ac, sc, lds, st, o, d, r, cur, sp, T, W, tie, again, 0xFFFFFFFFu, 0u, wc);
        overflow = again;
    }
    if (tie) flags |= 2u;
    if (overflow) flags |= 8u;
    if (valid && flags == 0u && W >= 0 && !ot_final_check(ac, sc, lds, o, d, r, T, W)) flags |= 4u;
    if (valid && flags != 0u) {
        uint32_t node = 0;
        T = INFINITY;
        W = -1;
        closest_hit_resume<COUNT, false, false>(sc, lds, o, d, node, T, W, 0xFFFFFFFFu, wc);
    }
}

__device__ __forceinline__ void ot_stage(const SceneDev& sc, const AccelDev& ac, float4* lds) {
    const uint32_t n4 = ac.n_lds_nodes * 7u, p4 = sc.n_lds_prims * 3u;
    for (uint32_t i = threadIdx.x; i < n4; i += blockDim.x) {  // 7 of a node's MPT_OT_NODE_STRIDE float4 (the rest is padding)
        const uint32_t node = i / 7u;
        lds[i] = ac.nodes[node * MPT_OT_NODE_STRIDE + (i - node * 7u)];
    }
    for (uint32_t i = threadIdx.x; i < ac.n_always * 5u; i += blockDim.x) lds[ac.lds_always_off + i] = ac.always[i];
    for (uint32_t i = threadIdx.x; i < p4; i += blockDim.x) lds[sc.lds_prim_off + i] = sc.prims[i];
    for (uint32_t i = threadIdx.x; i < sc.n_lds_mats * 2u; i += blockDim.x) lds[sc.lds_mat_off + i] = sc.mats[i];
    __syncthreads();
}
__device__ __forceinline__ OtStack ot_stack(const AccelDev& ac, float4* lds_raw, uint32_t wave_global) {
    OtStack st;
    const uint32_t lane = threadIdx.x & 63u, wave_local = threadIdx.x >> 6;
    st.lds = (LdsStack)(lds_raw + (ac.lds_stack_off >> 4)) + wave_local * ac.stack_depth * 64u + lane;
    st.depth = ac.stack_depth;
    (void)wave_global;
    return st;
}

// ring records: plain loads and stores (non-temporal pops lost 5 % in round 4, non-temporal pushes 11 % in round 5: a record is
// popped a step or two after it was pushed and must still be in the L2 — tests/experiments/rejected_r05.h)
__device__ __forceinline__ float4 ot_pop4(const float4* p) { return *p; }
__device__ __forceinline__ uint4 ot_pop4u(const uint4* p) { return *p; }
__device__ __forceinline__ void ot_put4(float4* p, const float4& v) { *p = v; }
__device__ __forceinline__ void ot_put4u(uint4* p, const uint4& v) { *p = v; }

// ---- the pipeline kernel ------------------------------------------------------------------------------------------
#define MPT_OT_RING_R 0u                  // R0 ... R(NR-1)
#define MPT_OT_RING_E MPT_OT_NR
#define MPT_OT_RING_M (MPT_OT_NR + 1u)    // M0 ... M(MLEVELS-1)
// A primary step adds up to 64 rays and runs only while no ring holds a full wave; with NR + 1 + MLEVELS rings of < 64 rays each
// that alone no longer bounds the rays a wave has in flight by a ring's 512 records, so a primary step also needs the total to
// leave room (the fullest fresh-ray ring runs a partial step otherwise).
#define MPT_OT_MAX_INFLIGHT (MPT_WL_RING - 64u)
#define MPT_OT_NONE 0xFFu

// Two operating points of ONE body (round 5):
//   k_ordered<.., 5>   five workgroups of 256 threads per CU = 5 waves/SIMD at 96 VGPRs, no scratch, ~100 nodes of the tree in each workgroup's LDS
//   k_ordered<.., 6>   two workgroups of 768 per CU = 6 waves/SIMD at 80 VGPRs — the walk loops fit, 22-24 values of a step's bookkeeping
//                are spilled around them (a few dwords of scratch per STEP, none per trip) — and ~270 nodes in LDS
// Same-box A/B (gpurun_out/r05/s5-s8): the sixth wave wins on scenes whose tree a compute die's L2 can mostly hold and loses on
// bigger ones, where one more wave per SIMD is one more stream of misses — 8 / 20 / 80 bunnies (40 k / 99 k / 397 k primitives):
// -2.5 % / -1.0..-1.5 % / -0.6 %, with each claim range rendering a vertical stripe of the image (MPT_TILE_ORDER=3, mpt_hip.hip) -3.6 % /
// -2.6 % / -2.7 %; the 1 M-triangle height fields of configs[4]: +0.5..+7 % SLOWER, with or without the stripes.  The host picks by
// the scene's size (mpt_hip.hip: ordered_point; MPT_OT_OCC=5|6 forces one).
#define MPT_OT6_THREADS 768
#define MPT_OT6_WAVES 6
// (one kernel template, the operating point in its launch bounds: a shared __forceinline__ body behind two __global__ wrappers cost the
//  five-wave instantiation 4-18 spilled VGPRs — the by-reference / by-value copies of the argument structs changed the allocation)
template <bool COUNT, bool ALL_LDS, int OCC = MPT_OT_WAVES>
__global__ __launch_bounds__(OCC == MPT_OT6_WAVES ? MPT_OT6_THREADS : MPT_OT_THREADS, OCC) void k_ordered(PassParams pp, AccelDev ac, OtRings ring, OtBudgets budgets,
                                                                                                         uint32_t wl_block, uint32_t wl_min, uint32_t wl_div) {
    extern __shared__ float4 lds_raw[];
    // camera and budgets live in LDS (the MPT_LDS_CFG_F4 block behind the material table), not in scalar registers across the step loop
    // (as in k_wavelocal, mpt_kernels.h)
    const uint32_t cfg_off = mpt_lds_cfg_off_f4(pp.scene.lds_mat_off);
    if (threadIdx.x == 0) {
        announce_resident(pp);
        lds_raw[cfg_off + 0] = make_float4(pp.cam.x, pp.cam.y, pp.cam.z, pp.W);
        lds_raw[cfg_off + 1] = make_float4(pp.first.x, pp.first.y, pp.first.z, pp.H);
        lds_raw[cfg_off + 2] = make_float4(pp.vu.x, pp.vu.y, pp.vu.z, 0.0f);
        lds_raw[cfg_off + 3] = make_float4(pp.vv.x, pp.vv.y, pp.vv.z, 0.0f);
        uint32_t* w = (uint32_t*)(lds_raw + cfg_off + 4);
        for (uint32_t k = 0; k < MPT_OT_MLEVELS; ++k) {
            w[k] = budgets.trips[k];
            w[8u + k] = budgets.min_active[k];
        }
        w[15] = budgets.inplace_min;
        // ... and what only a CLAIM of path ids needs (words 16..19 of the block's 32; the stacks start behind the block: ordered_views)
        static_assert(4u * (MPT_LDS_CFG_F4 - 4u) >= 20u, "configuration block too small");
        w[16] = wl_block;
        w[17] = wl_min;
        w[18] = wl_div * (((uint32_t)(gridDim.x * (blockDim.x >> 6)) + MPT_NGROUP - 1u) / MPT_NGROUP);   // wl_div * waves per claim range
        w[19] = pp.desc->total_paths / (pp.S * 64u);                                                   // this rank's tiles
    }
    ot_stage(pp.scene, ac, lds_raw);
    MPT_CLOCK_BEGIN();
    const LdsNodes lds = (LdsNodes)lds_raw;
    const __attribute__((address_space(3))) uint32_t* lds_cfg_u32 = (const __attribute__((address_space(3))) uint32_t*)(lds + cfg_off + 4u);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave_id = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const OtStack st = ot_stack(ac, lds_raw, wave_id);
    const uint32_t wbase = wave_id * (MPT_OT_RINGS * MPT_WL_RING);
    uint32_t cnt[MPT_OT_RINGS];   // wave-uniform ring fills (the rings are stacks: newest first)
#pragma unroll
    for (uint32_t k = 0; k < MPT_OT_RINGS; ++k) cnt[k] = 0u;
    uint32_t cur = 0, end = 0;
    uint32_t grp = blockIdx.x & (MPT_NGROUP - 1u);
    uint32_t seen = 0;
    bool exhausted = false;
    uint32_t tile_cached = 0xFFFFFFFFu, tile_xy_cached = 0u;
    uint32_t n_rays = 0, n_paths = 0, n_flagged = 0, n_parked = 0;
    WorkCount wc = {};
#ifdef MPT_OT_TIMES
    unsigned long long ot_acc[OT_NREG] = {}, ot_steps[8] = {}, ot_lanes[8] = {};   // slots: R (all octants), E, M0.., primary last
#endif
    for (;;) {
        OT_TIC();
        // ---- step choice: a full wave of the most advanced kind of work; else new paths; else what is left ----------------
        uint32_t kind = MPT_OT_NONE;  // ring to pop from; NONE = primary step
        if (cnt[MPT_OT_RING_E] >= 64u) kind = MPT_OT_RING_E;
#pragma unroll
        for (int k = (int)MPT_OT_RINGS - 1; k >= (int)MPT_OT_RING_M; --k)  // the longest walks first
            if (kind == MPT_OT_NONE && cnt[k] >= 64u) kind = (uint32_t)k;
        uint32_t r_best = MPT_OT_RING_R, r_fill = 0u, in_flight = 0u;   // the fullest fresh-ray ring
#pragma unroll
        for (uint32_t k = 0; k < MPT_OT_RINGS; ++k) {
            in_flight += cnt[k];
            if (k < MPT_OT_RING_E && cnt[k] > r_fill) {
                r_fill = cnt[k];
                r_best = k;
            }
        }
        if (kind == MPT_OT_NONE && (r_fill >= 64u || (MPT_OT_NR > 1u && in_flight > MPT_OT_MAX_INFLIGHT && r_fill != 0u))) kind = r_best;
        if (kind == MPT_OT_NONE) {
            if (!exhausted && cur == end) {  // guided self-scheduling of path ids, as k_wavelocal (mpt_kernels.h)
                uint32_t k = 0, blk = 0, rend = 0;
                bool got = false;
                if (lane == 0) {
                    const uint32_t c_block = lds_cfg_u32[16], c_min = lds_cfg_u32[17], c_div = lds_cfg_u32[18], c_tiles = lds_cfg_u32[19];   // (claim parameters: from LDS)
                    for (uint32_t t = 0; t < MPT_NGROUP && !got; ++t) {
                        const uint32_t re = range_paths(c_tiles, pp.S, grp);
                        const uint32_t left = seen < re ? re - seen : 0u;
                        blk = (left / c_div) & ~63u;
                        blk = blk < c_min ? c_min : (blk > c_block ? c_block : blk);
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
                if (!got) {
                    exhausted = true;
                } else {
                    cur = k;
                    end = (k + blk < rend) ? k + blk : rend;
                }
            }
            if (exhausted) {  // drain: partial steps, tree walks and hits first (they feed ring R)
#pragma unroll
                for (int k = (int)MPT_OT_RINGS - 1; k >= (int)MPT_OT_RING_M; --k)
                    if (kind == MPT_OT_NONE && cnt[k] != 0u) kind = (uint32_t)k;
                if (kind == MPT_OT_NONE) {
                    if (r_fill != 0u) kind = r_best;
                    else if (cnt[MPT_OT_RING_E] != 0u) kind = MPT_OT_RING_E;
                    else break;
                }
            }
        }
        OT_TOC(0);
        // ---- rays of the step ---------------------------------------------------------------------------------------
        PathState ps;
        PathRngDev g;
        bool valid = false;
        float T = INFINITY;
        int W = -1;
        uint32_t at = 0, walk_cur = 0u, walk_sp = 0u;   // walk state of a ring-M ray (fresh: the root, empty stack)
        bool walk_lost = false, walk_again = false;     // a stack entry was dropped / this is already the second walk
        if (kind == MPT_OT_NONE) {
            // 64 new paths = one 8x8 pixel tile at one sample index.  Everything about the tile is wave-uniform; its table
            // entry is kept from the previous primary step (consecutive steps of a claim walk the samples of one tile)
            const uint32_t pchunk = range_chunk_to_path_chunk(pp, cur >> 6, grp);  // = tile * S + sample
            cur += 64u;
            uint32_t tl, sidx;
            if (pp.s_shift != 0xFFu) {
                tl = pchunk >> pp.s_shift;
                sidx = pchunk & (pp.S - 1u);
            } else {
                tl = pchunk / pp.S;
                sidx = pchunk - tl * pp.S;
            }
            if (tl != tile_cached) {
                tile_cached = tl;
                tile_xy_cached = (uint32_t)__builtin_amdgcn_readfirstlane((int)pp.tile_xy[tl]);
            }
            ps.path = pchunk * 64u + lane;
            const uint32_t px = (tile_xy_cached & 0xFFFFu) * 8u + (lane & 7u), py = (tile_xy_cached >> 16) * 8u + (lane >> 3);
            if (px < pp.width && py < pp.height) {
                CamView cv;
                const v4f c0 = lds[cfg_off], c1 = lds[cfg_off + 1u], c2 = lds[cfg_off + 2u], c3 = lds[cfg_off + 3u];
                cv.cam = f3(c0.x, c0.y, c0.z);
                cv.first = f3(c1.x, c1.y, c1.z);
                cv.vu = f3(c2.x, c2.y, c2.z);
                cv.vv = f3(c3.x, c3.y, c3.z);
                cv.W = c0.w;
                cv.H = c1.w;
                float uvx, uvy;
                pixel_uv(cv, px, py, uvx, uvy);
                gen_primary(pp, cv, px, py, uvx, uvy, pp.sample_begin + sidx, ps, g);
                valid = true;
                n_paths++;
            }
        } else {
            uint32_t c = 0;
#pragma unroll
            for (uint32_t k = 0; k < MPT_OT_RINGS; ++k)
                if (k == kind) c = cnt[k];
            const uint32_t take = c < 64u ? c : 64u;
            valid = lane < take;
            at = wbase + kind * MPT_WL_RING + (c - take + lane);
#pragma unroll
            for (uint32_t k = 0; k < MPT_OT_RINGS; ++k)
                if (k == kind) cnt[k] = c - take;
            if (valid) {
                const float4 a = ot_pop4(ring.od() + at), b = ot_pop4(ring.dt() + at);
                ps.o = f3(a.x, a.y, a.z);
                ps.d = f3(a.w, b.x, b.y);
                ps.thr.x = b.z;
                ps.thr.y = b.w;
                if (kind >= MPT_OT_RING_M) {
                    const uint4 tv = ot_pop4u(ring.tv() + at);
                    T = __uint_as_float(tv.x);
                    W = (int)tv.y;
                    walk_cur = tv.z;
                    walk_sp = tv.w & 0xFFFFu;
                    walk_lost = (tv.w & 0x40000000u) != 0u;
                    walk_again = (tv.w & 0x80000000u) != 0u;
                }
            }
            if (kind >= MPT_OT_RING_M) {  // a parked walk brings its stack along: back into this lane's LDS column
                if (!valid) walk_cur = MPT_OT_DONE;
#pragma unroll
                for (uint32_t k = 0; k < MPT_OT_PARK / 2u; ++k) {
                    if (__ballot(valid && walk_sp > 2u * k) == 0ull) break;   // (no lane's stack is this deep: one ballot, not a maximum over the lanes by six ds_bpermute)
                    if (valid && walk_sp > 2u * k) {
                        const uint4 e = ot_pop4u(ring.sk(k) + at);
                        st.lds[(2u * k) * 64u] = v2u{e.x, e.y};
                        st.lds[(2u * k + 1u) * 64u] = v2u{e.z, e.w};
                    }
                }
            }
        }
        // the rest of a record is not needed by the tree walk: ring M steps load it afterwards
        auto load_rest = [&]() {
            const uint4 ia = ot_pop4u(ring.ia() + at);
            ps.thr.z = __uint_as_float(ia.x);
            ps.path = ia.y;
            ps.bounce = ia.w >> 27;
            if (kind < MPT_OT_RING_E) {
                T = __uint_as_float(ia.z & 0x7FFFFFFFu);
                W = (int)(ia.w & 0x07FFFFFFu);
            }
            ps.L = f3(0.0f, 0.0f, 0.0f);
            ps.La = 0.0f;
            if ((ia.z & 0x80000000u) != 0u) {
                const float4 cc = ot_pop4(ring.tl() + at);
                ps.L = f3(cc.x, cc.y, cc.z);
                ps.La = cc.w;
            }
            return;
            g.lit_seed = 0;
            if (pp.sp.rng_mode == 0) g.lit_seed = pcg_hash(pcg_hash(pp.pixel_seed[g.pixel]));
        };
        if (valid && kind != MPT_OT_NONE && kind < MPT_OT_RING_M) load_rest();
        // ---- the hits popped from ring R are shaded first — all 64 lanes of the step — and leave their bounce rays in `ps` ------------
        if (valid && kind != MPT_OT_NONE && kind < MPT_OT_RING_E) {
            uint32_t px, py, sidx;
            path_to_pixel(pp, ps.path, px, py, sidx);
            g.pixel = py * pp.width + px;
            g.sample = pp.sample_begin + sidx;
            g.lit_seed = 0;
            if (pp.sp.rng_mode == 0) g.lit_seed = pcg_hash(pcg_hash(pp.pixel_seed[g.pixel]));
            if (!shade_bounce(pp.scene, lds, pp.sp, g, ps, T, W)) {   // the path ends here (depth limit, material guard)
                store_slot(pp.slots, ps.path, clamp01(ps.L.x), clamp01(ps.L.y), clamp01(ps.L.z), clamp01(ps.La));
                valid = false;
            }
            T = INFINITY;
            W = -1;
        }
        OT_TOC(1);
#ifdef MPT_OT_TIMES
        const uint32_t ot_slot = kind == MPT_OT_NONE ? 2u + MPT_OT_MLEVELS : kind < MPT_OT_RING_E ? 0u : kind - MPT_OT_RING_E + 1u;
        ot_steps[ot_slot] += 1;
        ot_lanes[ot_slot] += (unsigned long long)__popcll(__ballot(valid));
#endif
        uint32_t dest = MPT_OT_NONE;      // ring this lane's ray goes to next
        bool shade = false;               // ... or its closest hit is final: one bounce of shading now
        uint32_t walk_kind = MPT_OT_NONE; // wave-uniform: the step walks the tree with this ring's budget
        bool walking = false;             // ... and this lane takes part
        const OtRay r = ot_ray(ps.o, ps.d);
        if (kind == MPT_OT_NONE || kind < MPT_OT_RING_E) {
            // ---- TOP TEST: always-list spheres + the root's boxes ------------------------------------------------------
            if (valid) {
                bool tie = false, need = false;
                if (ot_degenerate(ps.o, ps.d, ac.o_limit)) {
                    dest = MPT_OT_RING_E;
                } else {
                    ot_top_test<COUNT>(ac, lds, ps.o, ps.d, r, T, W, tie, need, wc);
                    if (tie) dest = MPT_OT_RING_E;
                    else if (need) dest = MPT_OT_RING_M;
                    // a sphere of the always list and nothing to walk: checked and shaded here ... or nothing hit: the sky.
                    // (A ring of hits, shaded 64 at a time at full width, was tried and measured: the extra ring hop
                    // costs more than the divergence of the hit branch — scene.xml 28.7 vs 28.1 ms, bunny x20 22.0 vs 21.2)
                    else if (W >= 0 && !ot_final_check(ac, pp.scene, lds, ps.o, ps.d, r, T, W)) dest = MPT_OT_RING_E;
                    else shade = true;
                }
            }
            // most of the wave has to walk the tree: do it now, as a step of ring M0 would, and save those rays the trip
            // through the ring (80 bytes written and read back per ray)
            if ((uint32_t)__popcll(__ballot(dest == MPT_OT_RING_M)) >= (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_cfg_u32[15])) {
                walk_kind = MPT_OT_RING_M;
                walking = dest == MPT_OT_RING_M;
                if (walking) dest = MPT_OT_NONE;
                else walk_cur = MPT_OT_DONE;
            }
            OT_TOC(2);
        } else if (kind >= MPT_OT_RING_M) {
            walk_kind = kind;
            walking = valid;
        }
        if (walk_kind != MPT_OT_NONE) {
            // ---- closest-first walk, continued for this ring's budget of node-loop trips ------------------------------
            bool tie = false, done;
            uint32_t budget = 0u, min_active = 0u;
            budget = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_cfg_u32[walk_kind - MPT_OT_RING_M]);
            min_active = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_cfg_u32[8u + walk_kind - MPT_OT_RING_M]);
            // (the drain walks to the end: parking a handful of rays again and again does not pay)
            if (!exhausted && (walk_kind + 1u < MPT_OT_RINGS || min_active != 0u))
                done = ot_walk<COUNT, true, ALL_LDS>(ac, pp.scene, lds, st, ps.o, ps.d, r, walk_cur, walk_sp, T, W, tie, walk_lost,
                                                     walk_kind + 1u < MPT_OT_RINGS ? budget : 0x7FFFFFFFu, min_active, wc);
            else
                done = ot_walk<COUNT, false, ALL_LDS>(ac, pp.scene, lds, st, ps.o, ps.d, r, walk_cur, walk_sp, T, W, tie, walk_lost,
                                                      0xFFFFFFFFu, 0u, wc);
            if (walking) {
                if (kind == walk_kind) load_rest();   // (a step of ring M; an in-place walk has everything in registers)
                if (tie) dest = MPT_OT_RING_E;   // (whatever is left of the walk does not matter then)
                else if (!done) dest = walk_kind + 1u < MPT_OT_RINGS ? walk_kind + 1u : walk_kind;  // parked with its walk state
                else if (walk_lost) {  // the stack dropped entries: once more from the root, now with the T found
                    if (walk_again) dest = MPT_OT_RING_E;
                    else {
                        dest = MPT_OT_RING_M;
                        walk_cur = 0u;
                        walk_sp = 0u;
                        walk_lost = false;
                        walk_again = true;
                    }
                }
                else if (W >= 0 && !ot_final_check(ac, pp.scene, lds, ps.o, ps.d, r, T, W)) dest = MPT_OT_RING_E;
                else shade = true;
            }
            OT_TOC(3);
        } else if (kind == MPT_OT_RING_E) {
            // ---- reference-order walk (PathTracing.h:75-204 as closest_hit_resume restates it) ----------------------
            if (valid) {
                uint32_t node = 0;
                T = INFINITY;
                W = -1;
                closest_hit_resume<COUNT, false, false>(pp.scene, lds, ps.o, ps.d, node, T, W, 0xFFFFFFFFu, wc);
                shade = true;
            }
            OT_TOC(5);
        }
        // ---- one bounce of shading for the rays whose closest hit is final ----------------------------------------------
        if (shade) {
            n_rays++;
            if (W >= 0) {
                dest = MPT_OT_RING_R;    // the hit, as it is: shaded by the step that pops it
            } else {                     // the sky ends the path (PathTracing.h:225-232)
                shade_bounce(pp.scene, lds, pp.sp, g, ps, T, -1);
                store_slot(pp.slots, ps.path, clamp01(ps.L.x), clamp01(ps.L.y), clamp01(ps.L.z), clamp01(ps.La));
            }
        }
        n_flagged += dest == MPT_OT_RING_E ? 1u : 0u;
        n_parked += dest != MPT_OT_NONE && dest >= MPT_OT_RING_M ? 1u : 0u;
        OT_TOC(6);
        // ---- wave64 compaction into the rings: ballot + mbcnt prefix per ring, fills stay wave-uniform ------------------
        if (__ballot(dest != MPT_OT_NONE) != 0ull) {
            uint32_t to = 0;
#pragma unroll
            for (uint32_t k = 0; k < MPT_OT_RINGS; ++k) {
                const unsigned long long m = __ballot(dest == k);
                if (dest == k) to = wbase + k * MPT_WL_RING + cnt[k] + wave_rank(m);
                cnt[k] += (uint32_t)__popcll(m);
            }
            if (dest != MPT_OT_NONE) {
                ot_put4(ring.od() + to, make_float4(ps.o.x, ps.o.y, ps.o.z, ps.d.x));
                ot_put4(ring.dt() + to, make_float4(ps.d.y, ps.d.z, ps.thr.x, ps.thr.y));
                {
                    const bool lit = ring_has_light(ps);
                    ot_put4u(ring.ia() + to, make_uint4(__float_as_uint(ps.thr.z), ps.path, (__float_as_uint(T) & 0x7FFFFFFFu) | (lit ? 0x80000000u : 0u),
                                                      ((uint32_t)W & 0x07FFFFFFu) | (ps.bounce << 27)));
                    if (lit) ot_put4(ring.tl() + to, make_float4(ps.L.x, ps.L.y, ps.L.z, ps.La));
                }
                if (dest >= MPT_OT_RING_M) {
                    // a ray parked by a top test starts its walk at the root (walk_cur = 0, walk_sp = 0 there)
                    ot_put4u(ring.tv() + to, make_uint4(__float_as_uint(T), (uint32_t)W, walk_cur,
                                                      walk_sp | (walk_lost ? 0x40000000u : 0u) | (walk_again ? 0x80000000u : 0u)));
                    if (walk_kind != MPT_OT_NONE) {   // (walk_sp = 0 for a ray that has not started)
#pragma unroll
                        for (uint32_t k = 0; k < MPT_OT_PARK / 2u; ++k) {
                            if (walk_sp > 2u * k) {
                                const v2u e0 = st.lds[(2u * k) * 64u], e1 = st.lds[(2u * k + 1u) * 64u];
                                ot_put4u(ring.sk(k) + to, make_uint4(e0.x, e0.y, e1.x, e1.y));
                            }
                        }
                    }
                }
            }
        }
        uint32_t worst = 0;
#pragma unroll
        for (uint32_t k = 0; k < MPT_OT_RINGS; ++k) worst = cnt[k] > worst ? cnt[k] : worst;
        if (worst > MPT_WL_RING) pp.desc->overflow = 1u;  // cannot happen: only a primary step adds rays (<= 64), and it
                                                          // runs only while every ring holds < 64: at most RINGS * 63 + 64
        OT_TOC(7);
    }
#ifdef MPT_OT_TIMES
    ot_flush_walk_times(wc, lane);
    if (lane == 0) {
        for (int k = 0; k < OT_NREG; ++k) atomicAdd(&g_ot_times[k], ot_acc[k]);
        for (int k = 0; k <= (int)(2u + MPT_OT_MLEVELS); ++k) {
            atomicAdd(&g_ot_times[OT_NREG + k], ot_steps[k]);
            atomicAdd(&g_ot_times[OT_NREG + 8 + k], ot_lanes[k]);
        }
    }
#endif
    MPT_CLOCK_END();
    flush_stats<COUNT>(pp.desc, n_rays, n_paths, wc);
    note_wave_exit(pp);
    {
        unsigned long long a = n_flagged, b = n_parked;
        for (int off = 32; off > 0; off >>= 1) {
            a += __shfl_down(a, off);
            b += __shfl_down(b, off);
        }
        if (lane == 0) {
            if (a) atomicAdd(&pp.desc->flagged, a);
            if (b) atomicAdd(&pp.desc->parked, b);
        }
    }
}

// unit-test kernel: closest hit of arbitrary rays through the ordered walk (flags tell which rule sent a ray to the
// reference-order walk)
__global__ __launch_bounds__(256) void k_trace_rays_ordered(SceneDev sc, AccelDev ac, const float* o, const float* d, uint32_t n,
                                                           float* t_out, int* prim_out, float* n_out, int* front_out,
                                                           uint32_t* flags_out) {
    extern __shared__ float4 lds_raw[];
    ot_stage(sc, ac, lds_raw);
    const LdsNodes lds = (LdsNodes)lds_raw;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const OtStack st = ot_stack(ac, lds_raw, wave);
    const bool valid = i < n;
    F3 ro = f3(1, 1, 1), rd = f3(1, 1, 1);
    if (valid) {
        ro = f3(o[3 * i], o[3 * i + 1], o[3 * i + 2]);
        rd = f3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    }
    float t;
    int prim;
    uint32_t flags;
    WorkCount wc = {};
    closest_hit_ordered<false>(ac, sc, lds, st, ro, rd, valid, t, prim, flags, wc);
    if (!valid) return;
    t_out[i] = t;
    flags_out[i] = flags;
    if (prim >= 0) {
        HitInfo h = finish_hit(sc, lds, ro, rd, t, prim);
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
