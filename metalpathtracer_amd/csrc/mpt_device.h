// mpt_device.h — device-side building blocks of the hot path (gfx950 only).
//
// Restated from the reference's Metal shader (R/ = "MetalCpp Path Tracer/"):
//   RNG            R/Renderer/Shaders/Random.h:6-16
//   unit vector    R/Renderer/Shaders/PathTracing.h:25-31
//   slab test      R/Renderer/Shaders/PathTracing.h:52-72
//   closest hit    R/Renderer/Shaders/PathTracing.h:75-204
//   shading        R/Renderer/Shaders/PathTracing.h:207-259, Scatter.h:10-43
// but laid out for CDNA4: a stackless (threaded) BVH whose hot nodes live in LDS, primitives
// pre-gathered into leaf order (no index indirection), a de-duplicated material table.
//
// FP contract: this translation unit is compiled with -ffp-contract=off; every expression below is
// a sequence of single IEEE-754 binary32 operations in the same order as oracle/mpt_oracle.cpp, so
// the two agree bit for bit wherever libm is not involved (DESIGN.md "Parity").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MPT_WAVE 64
#define MPT_NSHARD 16
#define MPT_NGROUP 8          // work-cursor groups (one per XCD under round-robin placement)
#define MPT_NONE 0xFFFFFFFFu

struct F3 {
    float x, y, z;
};
__device__ __forceinline__ F3 f3(float x, float y, float z) { return F3{x, y, z}; }
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return F3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return F3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ F3 operator*(F3 a, float s) { return F3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ F3 operator*(float s, F3 a) { return F3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ F3 operator-(F3 a) { return F3{-a.x, -a.y, -a.z}; }
__device__ __forceinline__ float dot3(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ F3 cross3(F3 a, F3 b) {
    return F3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// ---- 1.0f / x ---------------------------------------------------------------------------------------
// The compiler expands a correctly rounded FP32 division into v_div_scale x 2, v_rcp_f32, one multiply and five fused
// multiply-adds, v_div_fmas and v_div_fixup: 11 vector instructions.  v_div_scale / v_div_fmas / v_div_fixup only do
// something when an operand or the quotient leaves the normal range (or is 0 / Inf / NaN); everywhere else the result
// is the fma chain alone — rcp_chain() below, 7 instructions, the SAME operations in the same order.  The range in
// which the two agree bit for bit is not argued but MEASURED on the device over all 2^32 operands (k_kat_rcp,
// tests/test_gpu_parity.py::test_reciprocal_chain_equals_ieee_division_on_all_operands): every x with
// MPT_RCP_LO <= |x| <= MPT_RCP_HI.  mpt_rcp() takes the chain when EVERY lane of the wave is inside that range (one ballot, one
// scalar branch) and the full expansion otherwise.  MPT_FAST_RCP: 3 = that wave-uniform guard (the default: scene.xml 16.91 ->
// 16.80 ms, the closest-first kernel unchanged), 0 = always the full expansion (rounds 1-4), 1 = guarded per lane (-0.5 %: the
// exec-mask bookkeeping eats the gain), 2 = unguarded (pricing experiment only, wrong outside the range: -1.7 %, the price of the
// 22 divisions of k_wavelocal).
#ifndef MPT_FAST_RCP
#define MPT_FAST_RCP 3
#endif
#define MPT_RCP_LO 1.1754943508222875e-38f   // 2^-126 (smallest normal)
#define MPT_RCP_HI 8.5070591730234616e+37f   // 2^126
__device__ __forceinline__ float rcp_chain(float x) {
    float r = __builtin_amdgcn_rcpf(x);
    const float e0 = fmaf(-x, r, 1.0f);
    r = fmaf(e0, r, r);
    float q = r;                               // (1.0f * r)
    const float e1 = fmaf(-x, q, 1.0f);
    q = fmaf(e1, r, q);
    const float e2 = fmaf(-x, q, 1.0f);
    return fmaf(e2, r, q);
}
__device__ __forceinline__ bool rcp_chain_exact(float x) { return fabsf(x) >= MPT_RCP_LO && fabsf(x) <= MPT_RCP_HI; }
__device__ __forceinline__ float mpt_rcp(float x) {
#if MPT_FAST_RCP == 2
    return rcp_chain(x);
#elif MPT_FAST_RCP == 3   // wave-uniform guard: a wave whose operands are all inside the range takes the chain, any other the full expansion
    if (__builtin_expect(__ballot(!rcp_chain_exact(x)) == 0ull, 1)) return rcp_chain(x);
    return 1.0f / x;
#elif MPT_FAST_RCP == 1
    float r = rcp_chain(x);
    if (__builtin_expect(!rcp_chain_exact(x), 0)) r = 1.0f / x;
    return r;
#else
    return 1.0f / x;
#endif
}
// 1/x, 1/y, 1/z under ONE guard (closest_hit_resume: PathTracing.h:61, `1.0 / r.direction[i]` for the three axes)
__device__ __forceinline__ void mpt_rcp3(float x, float y, float z, float& ix, float& iy, float& iz) {
#if MPT_FAST_RCP == 3
    if (__builtin_expect(__ballot(!(rcp_chain_exact(x) && rcp_chain_exact(y) && rcp_chain_exact(z))) == 0ull, 1)) {
        ix = rcp_chain(x);
        iy = rcp_chain(y);
        iz = rcp_chain(z);
        return;
    }
    ix = 1.0f / x;
    iy = 1.0f / y;
    iz = 1.0f / z;
#else
    ix = mpt_rcp(x);
    iy = mpt_rcp(y);
    iz = mpt_rcp(z);
#endif
}
// normalize = v * (1 / sqrt(dot(v,v))) — same definition as the oracle (see its comment).
__device__ __forceinline__ F3 normalize3(F3 a) {
    float inv = mpt_rcp(sqrtf(dot3(a, a)));
    return a * inv;
}
__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }

// ---- RNG ------------------------------------------------------------------------------------------
// Random.h:6-11 — PCG-RXS-M-XS without the final multiply.
__device__ __forceinline__ uint32_t pcg_hash(uint32_t s) {
    uint32_t state = s * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state);
    return (word >> 22u) ^ word;
}
// Random.h:13-16 — seed by value; may return exactly 1.0f.
__device__ __forceinline__ float pcg_float(uint32_t s) { return (float)pcg_hash(s) / 4294967296.0f; }

// Philox4x32-10; counter (pixel, sample, bounce, 0), key (seed_lo, seed_hi).
struct U4 {
    uint32_t x, y, z, w;
};
// UNIFORM_KEY: the key is the same in all lanes (the render's seed).  The compiler then works out the ten round keys
// once per kernel and keeps all twenty words in scalar registers — which the persistent kernels do not have: they were
// spilled to VGPR lanes, and every Philox call paid 20 v_readlane + 20 hazard s_nop for them.  An empty asm after each
// bump makes the key opaque, so that it is bumped on the scalar unit each round (2 s_add) and lives in two registers.
template <bool UNIFORM_KEY = false>
__device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                            uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
        if (UNIFORM_KEY) asm volatile("" : "+s"(k0), "+s"(k1));
    }
    return U4{c0, c1, c2, c3};
}
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-08f; }

// sin/cos(2*pi*u), u in [0,1): exact quadrant reduction on u, Taylor polynomials on [-pi/4,pi/4].
__device__ __forceinline__ void sincos_2pi(float u, float& s_out, float& c_out) {
    float x = u * 4.0f;
    int q = (int)(x + 0.5f);
    float r = x - (float)q;
    float th = r * 1.57079637050628662109375f;
    float t2 = th * th;
    float ps = -1.98412701138295233249664306640625e-4f + t2 * 2.755731884462875314056873321533203125e-6f;
    ps = 8.3333337679505348205566406250e-3f + t2 * ps;
    ps = -0.16666667163372039794921875f + t2 * ps;
    float s = th + (th * t2) * ps;
    float pc = -1.38888892251998186111450195312500e-3f + t2 * 2.48015876422869041562080383300781250e-5f;
    pc = 4.1666667908430099487304687500e-2f + t2 * pc;
    pc = -0.5f + t2 * pc;
    float c = 1.0f + t2 * pc;
    int k = q & 3;
    s_out = (k == 0) ? s : (k == 1) ? c : (k == 2) ? -s : -c;
    c_out = (k == 0) ? c : (k == 1) ? -s : (k == 2) ? -c : s;
}

// LDS image: an explicit address-space-3 pointer so that fetches are ds_read_b128, never flat loads
// (a generic pointer selected between LDS and global turns into flat_load: measured 0 LDS reads per ray).
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) v4f* LdsNodes;

// ---- scene on the device ----------------------------------------------------------------------------
// Threaded BVH node, 32 bytes (same size as the reference's, SURVEY App. D buf 0):
//   n0 = (bmin.xyz, A)   n1 = (bmax.xyz, B)
//   internal: box hit -> go to A (first child in the reference's visit order: the RIGHT child,
//             PathTracing.h:191-193), box miss -> go to B (skip the subtree)
//   leaf:     A = MPT_NODE_HOLD | first*16 + (count-1) — "hold this leaf": a lane's node word then names the primitives
//             to test — and B = the node after the leaf (hit or miss)
//   n_nodes <= index < MPT_NODE_HOLD terminates.  Nodes are stored breadth-first so that the first n_lds nodes (top
//   of the tree) can be staged in LDS.
// Device primitive, 48 bytes, stored in leaf order:
//   triangle: (v0.xyz, bits(leaf << 1 | 1)) (e1.xyz, bits(mat)) (e2.xyz, bits(orig id))     e1 = v1-v0, e2 = v2-v0
//   sphere:   (c.xyz,  bits(leaf << 1))     (r, bits(own index),0, bits(mat)) (0,0,0,  bits(orig id))
struct SceneDev {
    const float4* nodes;
    const float4* prims;
    const float4* mats;   // 2 float4 per unique material: (albedo, type) (emission, power)
    uint32_t n_nodes;
    uint32_t n_lds_nodes;  // nodes [0, n_lds_nodes) are also in LDS
    uint32_t n_lds_prims;  // primitives [0, n_lds_prims) are also in LDS, right behind the node image
    uint32_t n_lds_mats;   // materials [0, n_lds_mats) are also in LDS, behind the primitives
    uint32_t n_prims;
    uint32_t n_mats;
    uint32_t lds_prim_off;  // where the primitive / material images start in the workgroup's LDS (in float4 units):
    uint32_t lds_mat_off;   // the reference-order kernels stage 2 float4 per node in front of them, k_ordered 7
};
// Primitive record word p0.w: (reference leaf id << 1) | type (0 sphere, 1 triangle).  The leaf id is what the ordered
// walk's final check needs (mpt_ordered.h); leaves are numbered in the reference's visit order.
__device__ __forceinline__ int prim_type(float4 p0) { return __float_as_int(p0.w) & 1; }
__device__ __forceinline__ uint32_t prim_ref_leaf(float4 p0) { return (uint32_t)__float_as_int(p0.w) >> 1; }

// Diagnostics build (-DMPT_CLOCK_STAMP): the shader clock a kernel actually holds = delta s_memtime / delta s_memrealtime x 100 MHz
// (MI355X_MICROARCH.md, DVFS item 6), summed over all waves of the trace kernels.  Nothing else reads these words.
#ifdef MPT_CLOCK_STAMP
__device__ unsigned long long g_clock[2];
#define MPT_CLOCK_BEGIN() const unsigned long long clk_t0_ = __builtin_amdgcn_s_memtime(), clk_r0_ = __builtin_amdgcn_s_memrealtime()
#define MPT_CLOCK_END()                                                                    \
    do {                                                                                   \
        if ((threadIdx.x & 63u) == 0) {                                                    \
            atomicAdd(&g_clock[0], __builtin_amdgcn_s_memtime() - clk_t0_);                \
            atomicAdd(&g_clock[1], __builtin_amdgcn_s_memrealtime() - clk_r0_);            \
        }                                                                                  \
    } while (0)
#else
#define MPT_CLOCK_BEGIN() do { } while (0)
#define MPT_CLOCK_END() do { } while (0)
#endif

// true in exactly one lane of the currently active lanes (used to count wave-level loop trips)
__device__ __forceinline__ bool first_active_lane() {
    const uint32_t me = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    return me == (uint32_t)__builtin_amdgcn_readfirstlane((int)me);
}

// max of a value over the active lanes of the wave (values here are small trip counts)
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t other = (uint32_t)__shfl_xor((int)v, off);
        v = other > v ? other : v;
    }
    return v;
}

struct WorkCount {
    uint32_t node_visits, aabb_hits, prim_tests;   // per lane (== the oracle's counters when summed)
    uint32_t node_iters, prim_iters, outer_iters;  // per WAVE loop trips (lane 0 only): x64 = issued lane slots
#ifdef MPT_DEBUG_WAVE_TIMES
    unsigned long long t_box, t_leaf;              // diagnostics build: shader-clock cycles spent in the two loops
    uint32_t wait_leaf, wait_done;                 // box-loop trips this lane idled holding a leaf / after finishing
#endif
#ifdef MPT_OT_TIMES
    unsigned long long ot_node_cycles, ot_leaf_cycles, ot_node_trips, ot_leaf_trips, ot_rounds;  // closest-first walk, per wave
    unsigned long long ot_node_lanes, ot_leaf_lanes;  // ... lanes that took part in those trips (per lane)
    unsigned long long ot_lds_nodes, ot_glb_nodes, ot_lds_prims, ot_glb_prims;  // node visits / primitive records by where they were served from (per lane)
    unsigned long long ot_pops, ot_pop_iters, ot_pop_calls, ot_pop_wave_iters;   // stack pops: calls and entries examined per lane; calls and loop trips per wave
#endif
};
#ifdef MPT_DEBUG_WAVE_TIMES
#define MPT_TIC(var) unsigned long long var = __builtin_amdgcn_s_memtime()
#define MPT_TOC(acc, var)                                          \
    do {                                                           \
        unsigned long long now_ = __builtin_amdgcn_s_memtime();    \
        (acc) += now_ - (var);                                     \
        (var) = now_;                                              \
    } while (0)
#else
#define MPT_TIC(var) do { } while (0)
#define MPT_TOC(acc, var) do { } while (0)
#endif

// One primitive record (3 x 16 B).  Primitives are stored by leaf depth (shallow leaves first); the first
// n_lds_prims of them are staged in LDS behind the node image.  On scene.xml 65 % of all primitive tests hit the three
// spheres near the root: from LDS they cost three ds_read_b128 instead of three scattered L2 requests per lane.
struct Prim3 {
    float4 p0, p1, p2;
};
__device__ __forceinline__ Prim3 load_prim(const SceneDev& sc, LdsNodes lds, uint32_t i) {
    Prim3 r;
    if (i < sc.n_lds_prims) {
        const LdsNodes q = lds + sc.lds_prim_off + 3u * i;
        const v4f a = q[0], b = q[1], c = q[2];
        r.p0 = make_float4(a.x, a.y, a.z, a.w);
        r.p1 = make_float4(b.x, b.y, b.z, b.w);
        r.p2 = make_float4(c.x, c.y, c.z, c.w);
    } else {
        r.p0 = sc.prims[3 * i];
        r.p1 = sc.prims[3 * i + 1];
        r.p2 = sc.prims[3 * i + 2];
    }
    return r;
}

// Primitives of one leaf, in index order — PathTracing.h:106-186.
template <bool COUNT>
__device__ __forceinline__ void leaf_test(const SceneDev& sc, LdsNodes lds, uint32_t first, uint32_t count, F3 o, F3 d,
                                          float& best_t, int& best_prim, WorkCount& wc) {
    for (uint32_t k = 0; k < count; ++k) {
        // the three 16-byte loads of a primitive are issued together (the third is used by triangles only, but a
        // load that waits for the type check costs a second L2 round trip per primitive)
        const Prim3 pr = load_prim(sc, lds, first + k);
        const float4 p0 = pr.p0, p1 = pr.p1, p2 = pr.p2;
        if (COUNT) {
            wc.prim_tests++;
            if (first_active_lane()) wc.prim_iters++;
        }
        const int ptype = prim_type(p0);
        if (ptype == 1) {  // PathTracing.h:143-176 Moeller-Trumbore, two-sided
            F3 v0 = f3(p0.x, p0.y, p0.z), e1 = f3(p1.x, p1.y, p1.z), e2 = f3(p2.x, p2.y, p2.z);
            F3 h = cross3(d, e2);
            float a = dot3(e1, h);
            if (fabsf(a) > 1e-5f) {
                float f = mpt_rcp(a);
                F3 s = o - v0;
                float u = f * dot3(s, h);
                if (u >= 0.0f && u <= 1.0f) {
                    F3 q = cross3(s, e1);
                    float v = f * dot3(d, q);
                    if (v >= 0.0f && u + v <= 1.0f) {
                        float tt = f * dot3(e2, q);
                        if (tt > 0.0001f && tt < best_t) {
                            best_t = tt;
                            best_prim = (int)(first + k);
                        }
                    }
                }
            }
        } else if (ptype == 0) {  // PathTracing.h:120-142 sphere, near root only
            F3 c = f3(p0.x, p0.y, p0.z);
            float radius = p1.x;
            F3 oc = o - c;
            float a = dot3(d, d);
            float b = dot3(oc, d);
            // b >= 0 (the centre lies behind the ray): the root (-b - sqrt(disc)) / a is <= 0 whatever disc is and fails
            // "> 0.0001" — signs are exact in floating point, so leaving before the square root and the division changes
            // nothing (a NaN b compares false and takes the full path).
            // one level of branching (the square root and the division are skipped by whole waves), the rest as selects:
            // every nested if costs the scalar unit ~6 instructions of exec-mask bookkeeping
            const float cc = dot3(oc, oc) - radius * radius;
            const float disc = b * b - a * cc;
            if (!(b >= 0.0f) && disc > 0.0f) {
                const float sq = sqrtf(disc);
                const float temp = (-b - sq) / a;
                const bool closer = temp < best_t && temp > 0.0001f;
                best_t = closer ? temp : best_t;
                best_prim = closer ? (int)(first + k) : best_prim;
            }
        }
    }
}

// Closest hit — PathTracing.h:75-204.  Visits nodes and tests primitives in exactly the reference's
// order (right child first, leaf primitives in index order), so ties resolve identically.
// LDS node image: an explicit address-space-3 pointer so that node fetches are ds_read_b128, never flat loads
// (a generic pointer selected between LDS and global turns into flat_load: measured 0 LDS reads per ray).

// Resumable closest hit with a wave-level budget of box-test loop trips.
//   in/out: node (next node to visit, 0 = root), best_t, best_prim  (fresh query: 0, +inf, -1)
//   returns true when the traversal is complete, false when the budget ran out first (node/best_* then hold the
//   exact state to resume from: the ray sees the same sequence of tests either way, so results do not change).
// "while-while" form: every lane first walks box tests until IT has a leaf to test (or is done); only then does the
// wave run the primitive loop, for all lanes with a pending leaf at once.
#ifndef MPT_LEAF_EARLY
#define MPT_LEAF_EARLY 8u
#endif
#define MPT_NODE_HOLD 0x80000000u   // hit link of a leaf record: HOLD | first << 4 | (count - 1)
template <bool COUNT, bool ALL_LDS, bool BUDGETED>
__device__ __forceinline__ bool closest_hit_resume(const SceneDev& sc, LdsNodes lds_nodes, F3 o, F3 d, uint32_t& node,
                                                   float& best_t, int& best_prim, uint32_t budget, WorkCount& wc,
                                                   uint32_t min_active = 0u) {
    float idx, idy, idz;
    mpt_rcp3(d.x, d.y, d.z, idx, idy, idz);   // PathTracing.h:61 (per call there): 1.0 / r.direction[i]
    const uint32_t n_nodes = sc.n_nodes, n_lds = sc.n_lds_nodes;
    // A direction with a NaN component (normalize of a zero vector: e.g. a refraction at the critical angle whose
    // discriminant rounds below zero) hits nothing — every sphere / triangle test mixes all three components and ends
    // in a comparison with NaN, which is false — but its slab tests PASS every box (min / max drop the NaN operand), so
    // the reference walks the whole tree: 0.5 s for one such ray on a 1 M-triangle scene, with every other wave of
    // the launch waiting for it.  The result of that walk is "miss"; return it without walking.  (The work counters
    // then lack this walk; the oracle, which restates the reference, makes it.)
    uint32_t i = (d.x != d.x || d.y != d.y || d.z != d.z) ? n_nodes : node;
    // The box-test loop is written for the SCALAR unit: one per CU, shared by the four SIMDs, and as busy as the vector
    // units in this kernel (rocprofv3: 0.50 SALU per VALU instruction; a loop with per-lane exits costs ~28 scalar
    // instructions of exec-mask bookkeeping per trip).  So the loop is wave-uniform — every lane runs every trip, lanes
    // that are not searching compute on a clamped node and discard — and a lane's state is one integer:
    //   i < n_nodes           searching: the node to test next
    //   i = HOLD | enc        holds a leaf (a leaf record's hit link IS that word); `skip` = where to go after it
    //   n_nodes <= i < HOLD   done
    // per trip: one compare for "searching", one v_cndmask for the link, two to commit — ~6 scalar instructions.
    uint32_t trips = 0;  // trips of the box-test loop this wave has made in this call (wave-uniform)
    MPT_TIC(tic_);
    for (;;) {
        uint32_t skip = 0;
        // (at least 1: "no lane searching" then leaves the loop by the same comparison as the early leave below)
        const uint32_t n_entered0 = (uint32_t)__popcll(__ballot(i < n_nodes)), n_entered = max(n_entered0, 1u);
#ifdef MPT_DEBUG_WAVE_TIMES
        uint32_t my_trips = 0, round_trips = 0;
#endif
        uint32_t n_searching = n_entered0;
        while (n_searching * MPT_LEAF_EARLY >= n_entered && (!BUDGETED || trips < budget)) {
            const bool searching = i < n_nodes;
            // When fewer than 1/8 of the lanes that entered this search are still looking for their next leaf, the
            // others — who hold a leaf — stop waiting: the leaf phase runs now and the searchers resume afterwards
            // from where they are (each lane still sees its own sequence of tests).  In the deep rings most box-loop
            // lane slots were such waits (utilisation 19-42 %); 28.7 -> 27.4 ms, thresholds 1/6 .. 1/32 all help.
            // (this is the loop condition)
#ifdef MPT_DEBUG_WAVE_TIMES
            my_trips += searching ? 1u : 0u;
            round_trips++;
#endif
            const uint32_t j = i < n_nodes - 1u ? i : n_nodes - 1u;
            float4 n0 = make_float4(0, 0, 0, 0), n1 = n0;
            bool box = false;
            // Lanes that are not searching sit the trip out under the exec mask (one s_and_saveexec / s_or per trip): they issue
            // no LDS read and toggle no ALU, while the loop stays wave-uniform in its control flow.  Letting them compute on a
            // clamped node and discard (no branch at all) issues the same instructions but burns power on 16-60 % idle lanes of
            // a kernel whose clock is power-limited: 22.5 -> 21.95 ms per 256-spp render of scene.xml with the mask.
            if (searching) {
                if (ALL_LDS || j < n_lds) {
                    const v4f a = lds_nodes[2 * j], b = lds_nodes[2 * j + 1];
                    n0 = make_float4(a.x, a.y, a.z, a.w);
                    n1 = make_float4(b.x, b.y, b.z, b.w);
                } else {
                    n0 = sc.nodes[2 * j];
                    n1 = sc.nodes[2 * j + 1];
                }
                // PathTracing.h:52-72 slab test with tMin = 1e-4, tMax = best t.  The per-axis early-outs
                // are equivalent to one test after the third axis (tMin only grows, tMax only shrinks).
                float t0 = (n0.x - o.x) * idx, t1 = (n1.x - o.x) * idx;
                float lo = fmaxf(0.0001f, idx < 0.0f ? t1 : t0);
                float hi = fminf(best_t, idx < 0.0f ? t0 : t1);
                t0 = (n0.y - o.y) * idy;
                t1 = (n1.y - o.y) * idy;
                lo = fmaxf(lo, idy < 0.0f ? t1 : t0);
                hi = fminf(hi, idy < 0.0f ? t0 : t1);
                t0 = (n0.z - o.z) * idz;
                t1 = (n1.z - o.z) * idz;
                lo = fmaxf(lo, idz < 0.0f ? t1 : t0);
                hi = fminf(hi, idz < 0.0f ? t0 : t1);
                box = hi > lo;
            }
            const uint32_t A = __float_as_uint(n0.w), B = __float_as_uint(n1.w);  // links: box hit / box missed
            if (COUNT) {
                wc.node_visits += searching ? 1u : 0u;
                wc.aabb_hits += (searching && box) ? 1u : 0u;
                if (first_active_lane()) wc.node_iters++;
            }
            const uint32_t next = box ? A : B;
            i = searching ? next : i;
            skip = searching ? B : skip;
            if (BUDGETED) trips++;
            n_searching = (uint32_t)__popcll(__ballot(i < n_nodes));
        }
#ifdef MPT_DEBUG_WAVE_TIMES
        if (COUNT) {
            if ((i & MPT_NODE_HOLD) != 0u) wc.wait_leaf += round_trips - my_trips;
            else wc.wait_done += round_trips - my_trips;   // finished, out of budget, or cut by the early leave
        }
#endif
        MPT_TOC(wc.t_box, tic_);
        if ((i & MPT_NODE_HOLD) != 0u) {
            const uint32_t enc = i & ~MPT_NODE_HOLD;
            if (COUNT && first_active_lane()) wc.outer_iters++;
            leaf_test<COUNT>(sc, lds_nodes, enc >> 4, (enc & 15u) + 1u, o, d, best_t, best_prim, wc);
            i = skip;
        }
        MPT_TOC(wc.t_leaf, tic_);
        // the wave goes round again only while some lane still has nodes to visit and budget is left ...
        if (BUDGETED && trips >= budget) break;
        const unsigned long long going = __ballot(i < n_nodes);
        if (going == 0ull) break;
        // ... and, in a budgeted step, while enough lanes are still working: the stragglers of a step are parked and
        // meet other stragglers in the next ring instead of holding 64 lanes for their long walks
        if (BUDGETED && (uint32_t)__popcll(going) < min_active) break;
    }
    node = i;
    return i >= n_nodes;
}

template <bool COUNT, bool ALL_LDS>
__device__ __forceinline__ void closest_hit(const SceneDev& sc, LdsNodes lds_nodes, F3 o, F3 d,
                                            float& best_t, int& best_prim, WorkCount& wc) {
    uint32_t node = 0;
    best_t = INFINITY;
    best_prim = -1;
    closest_hit_resume<COUNT, ALL_LDS, false>(sc, lds_nodes, o, d, node, best_t, best_prim, 0xFFFFFFFFu, wc);
}

struct HitInfo {
    F3 point, normal;
    bool front;
    int mat;      // material table index
    int orig_id;  // primitive id in the caller's numbering
};
// Surface point / geometric normal of the winning primitive — the values PathTracing.h:138-139,
// 168-169 compute when the hit is accepted, and the front-face flip of :196-201.
__device__ __forceinline__ HitInfo finish_hit(const SceneDev& sc, LdsNodes lds, F3 o, F3 d, float t, int prim) {
    HitInfo h;
    const Prim3 pr = load_prim(sc, lds, (uint32_t)prim);
    const float4 p0 = pr.p0, p1 = pr.p1, p2 = pr.p2;
    h.point = o + t * d;
    if (prim_type(p0) == 1) {
        h.normal = normalize3(cross3(f3(p1.x, p1.y, p1.z), f3(p2.x, p2.y, p2.z)));
    } else {
        h.normal = normalize3(h.point - f3(p0.x, p0.y, p0.z));
    }
    h.mat = __float_as_int(p1.w);
    h.orig_id = __float_as_int(p2.w);
    h.front = dot3(h.normal, d) < 0.0f;
    if (!h.front) h.normal = -h.normal;
    return h;
}

// ---- per-path state and shading --------------------------------------------------------------------------
struct PathState {
    F3 o, d;
    F3 thr;        // absorption.rgb (PathTracing.h:218)
    F3 L;          // light.rgb      (PathTracing.h:219)
    float La;      // light.a (absorption.a stays 1)
    uint32_t path; // path index inside the pass
    uint32_t bounce;
};

struct ShadeParams {
    int rng_mode, bsdf_mode, max_depth;
    uint32_t seed_lo, seed_hi;
    uint32_t primitive_count;  // uniforms.primitiveCount (material guard, PathTracing.h:234-236)
};

struct PathRngDev {
    uint32_t pixel, sample;  // philox counter words 0,1
    uint32_t lit_seed;       // literal: the stuck seed entering rayColor
};

// PathTracing.h:25-31 (literal: same u for z and phi, seed never advances), in two halves: the random
// numbers of the bounce depend on (pixel, sample, bounce) only, so shade_bounce draws them BEFORE it fetches the hit's
// primitive and material records — the 60-instruction Philox chain then runs under the latency of those loads instead of
// behind it.
struct BounceRandoms {
    float uz, uphi, u_extra;   // z = 2 uz - 1, phi = 2 pi uphi, Fresnel test
};
__device__ __forceinline__ BounceRandoms draw_bounce_randoms(const ShadeParams& sp, const PathRngDev& g, uint32_t bounce) {
    BounceRandoms b;
    if (sp.rng_mode == 0) {
        b.uz = b.uphi = b.u_extra = pcg_float(g.lit_seed);
    } else {
        const U4 r = philox4x32_10<true>(g.pixel, g.sample, bounce, 0u, sp.seed_lo, sp.seed_hi);
        b.uz = u01(r.x);
        b.uphi = u01(r.y);
        b.u_extra = u01(r.z);
    }
    return b;
}
__device__ __forceinline__ F3 random_unit_vector(const ShadeParams& sp, const BounceRandoms& b) {
    float z = 2.0f * b.uz - 1.0f, s, c;
    if (sp.rng_mode == 0) {
        const float t = 2.0f * 3.14159274101257324f * b.uphi;
        s = sinf(t);
        c = cosf(t);
    } else {
        sincos_2pi(b.uphi, s, c);
    }
    const float rr = sqrtf(1.0f - z * z);
    return f3(rr * c, rr * s, z);
}

__device__ __forceinline__ F3 reflect3(F3 i, F3 n) { return i - 2.0f * dot3(n, i) * n; }
__device__ __forceinline__ F3 refract3(F3 i, F3 n, float eta) {
    float dd = dot3(n, i);
    float k = 1.0f - eta * eta * (1.0f - dd * dd);
    if (k < 0.0f) return f3(0, 0, 0);
    return eta * i - (eta * dd + sqrtf(k)) * n;
}
// Scatter.h:10-20
__device__ __forceinline__ bool mirror_angle(float ri, F3 normal, F3 rayDir, float u) {
    float cosT = dot3(-1.0f * rayDir, normal);
    float sinT = sqrtf(1.0f - cosT * cosT);
    float r0 = (1.0f - ri) / (1.0f + ri);
    r0 = r0 * r0;
    float m = 1.0f - cosT;
    float m2 = m * m;
    float refl = r0 + (1.0f - r0) * (m2 * m2 * m);
    return (ri * sinT > 1.0f) || (refl > u);
}

// One iteration of the bounce loop of rayColor (PathTracing.h:221-256) for a ray whose closest hit is
// (t, prim).  Returns true if the path continues (ps holds the next ray), false if it ended (ps.L/La
// hold the final light, to be clamped by the caller: PathTracing.h:258).
__device__ __forceinline__ bool shade_bounce(const SceneDev& sc, LdsNodes lds, const ShadeParams& sp, const PathRngDev& g,
                                             PathState& ps, float t, int prim) {
    if (prim < 0) {  // PathTracing.h:225-232 sky
        F3 ud = normalize3(ps.d);
        float tt = 0.5f * (ud.y + 1.0f);
        F3 sky = f3(1.0f + (0.6f - 1.0f) * tt, 1.0f + (0.7f - 1.0f) * tt, 1.0f + (1.0f - 1.0f) * tt);
        ps.L.x += ps.thr.x * sky.x;
        ps.L.y += ps.thr.y * sky.y;
        ps.L.z += ps.thr.z * sky.z;
        ps.La += 1.0f;
        return false;
    }
    const BounceRandoms rnd = draw_bounce_randoms(sp, g, ps.bounce);
    HitInfo h = finish_hit(sc, lds, ps.o, ps.d, t, prim);
    if ((uint32_t)h.orig_id >= sp.primitive_count) return false;  // PathTracing.h:234-236
    float4 m0, m1;
    if ((uint32_t)h.mat < sc.n_lds_mats) {  // the de-duplicated material table is tiny: served from LDS
        const LdsNodes q = lds + sc.lds_mat_off + 2u * (uint32_t)h.mat;
        const v4f a = q[0], b = q[1];
        m0 = make_float4(a.x, a.y, a.z, a.w);
        m1 = make_float4(b.x, b.y, b.z, b.w);
    } else {
        m0 = sc.mats[2 * h.mat];
        m1 = sc.mats[2 * h.mat + 1];
    }
    const float mtype = m0.w, power = m1.w;
    if (power > 0.0f || mtype == 2.0f) {  // PathTracing.h:245-249
        ps.L.x += ps.thr.x * m1.x * power;
        ps.L.y += ps.thr.y * m1.y * power;
        ps.L.z += ps.thr.z * m1.z * power;
        ps.La += power;
    }
    const float u_extra = rnd.u_extra;
    F3 ruv = random_unit_vector(sp, rnd);
    F3 nd;
    bool through = false;
    if (sp.bsdf_mode == 2 && mtype == 0.0f) {  // Scatter.h:24-27,42 with randomFloat3 of Random.h:18-30 (dead in the reference)
        F3 c;
        if (sp.rng_mode == 0) {
            uint32_t s = g.lit_seed;
            c.x = pcg_float(s) * 2.0f - 1.0f;
            s = pcg_hash(s);
            c.y = pcg_float(s) * 2.0f - 1.0f;
            s = pcg_hash(s);
            c.z = pcg_float(s) * 2.0f - 1.0f;
        } else {  // (word 2 of the block is the Fresnel number of a dielectric bounce: never both in one bounce)
            c = f3(rnd.uz * 2.0f - 1.0f, rnd.uphi * 2.0f - 1.0f, u_extra * 2.0f - 1.0f);
        }
        nd = normalize3(h.normal + normalize3(c));
    } else if (sp.bsdf_mode == 0 || mtype == 0.0f) {
        nd = normalize3(h.normal + ruv);  // PathTracing.h:252-254
    } else if (mtype < 0.0f) {            // Scatter.h:28-31
        nd = normalize3(reflect3(ps.d, h.normal));
    } else {                              // Scatter.h:32-40
        float ri = h.front ? 1.0f / mtype : mtype;
        nd = mirror_angle(ri, h.normal, ps.d, u_extra) ? reflect3(ps.d, h.normal) : refract3(ps.d, h.normal, ri);
        nd = normalize3(nd);
        through = dot3(nd, h.normal) < 0.0f;
    }
    ps.o = through ? h.point - 0.0001f * h.normal : h.point + 0.0001f * h.normal;  // PathTracing.h:253
    ps.d = nd;
    ps.thr.x *= m0.x;  // PathTracing.h:255
    ps.thr.y *= m0.y;
    ps.thr.z *= m0.z;
    ps.bounce++;
    return (int)ps.bounce < sp.max_depth;
}
