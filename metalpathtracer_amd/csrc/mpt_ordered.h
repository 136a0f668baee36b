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
// reference leaf boxes (padded), so every primitive the reference could accept in front of the winner is seen here.
// tests/experiments/ordered_proto.cpp replays this rule on the CPU against the oracle: 0 differences in 890 M rays of
// the headline render (9 of the flagged rays would have differed); the GPU tests compare whole renders bit for bit.
//
// Pipeline shape (per persistent wave, rings as in k_wavelocal but sorted by KIND of work instead of by trip budget):
//   primary step   64 new paths: primary rays in registers, TOP TEST = the always-list spheres + the root node's four
//                  boxes.  94 % of the rays of scene.xml end here (sky, ground, spheres): they are shaded at full
//                  width; rays that touch a root child are parked in ring M with their (best t, best primitive)
//   ring R step    64 fresh bounce rays: the same
//   ring M step    64 rays that all have to walk the tree: closest-first walk (LDS stack), final check, shading
//   ring E step    reference-order walk for the flagged rays (rare)
// A step whose rays mostly need the tree (camera inside the mesh bounds) walks it at once instead of parking.
#pragma once
#include "mpt_device.h"
#include "mpt_kernels.h"

#define MPT_OT_LEAF 0x80000000u
#define MPT_OT_DONE 0xFFFFFFFFu
#define MPT_OT_KEY_MISS 0x7F800000u       // +inf: sorts behind every hit
#ifndef MPT_OT_THREADS
#define MPT_OT_THREADS 1024
#endif
#ifndef MPT_OT_WAVES
#define MPT_OT_WAVES 4                    // per SIMD: one 1024-thread workgroup per CU, up to 128 VGPRs
#endif
#define MPT_OT_SPILL 56u                  // stack entries per lane beyond the LDS part (global memory)
#define MPT_OT_RINGS 3u                   // R (fresh rays), M (rays that must walk the tree), E (reference-order walk)

struct AccelDev {
    const float4* nodes;    // 7 float4 per node, breadth-first (mpt_accel.h)
    const float4* refleaf;  // 2 float4 per reference leaf: (bmin, 0) (bmax, 0)
    const float4* always;   // 3 float4 per sphere of the always list: (c, leaf<<1) (r, bits(index), 0, mat) (0,0,0, orig id)
    uint2* spill;           // [waves][MPT_OT_SPILL][64] stack overflow area
    uint32_t n_nodes, n_lds_nodes, n_always;
    uint32_t lds_always_off;  // float4 index of the always list in LDS
    uint32_t lds_stack_off;   // byte offset of the stacks in LDS
    uint32_t stack_depth;     // LDS stack entries per lane
    float eps_abs;            // sub-trees are culled beyond best t * (1 + 2^-10) + eps_abs
    float o_limit;            // ray origins farther out than this (64 x the triangle extent) exceed what the box padding covers
};

typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) v2u* LdsStack;

struct OtStack {
    LdsStack lds;     // this lane's column: entry e at lds[e * 64]
    uint2* spill;     // this lane's column of the wave's spill block: entry e at spill[e * 64]
    uint32_t depth;
};
__device__ __forceinline__ void ot_push(const OtStack& st, uint32_t& sp, uint32_t key, uint32_t parent, bool& overflow) {
    if (sp < st.depth) st.lds[sp * 64u] = v2u{key, parent};
    else if (sp < st.depth + MPT_OT_SPILL) st.spill[(sp - st.depth) * 64u] = make_uint2(key, parent);
    else {
        overflow = true;  // deeper than any tree this builder makes: the ray goes to the reference-order walk
        return;
    }
    ++sp;
}
__device__ __forceinline__ uint2 ot_pop(const OtStack& st, uint32_t& sp) {
    --sp;
    if (sp < st.depth) {
        const v2u e = st.lds[sp * 64u];
        return make_uint2(e.x, e.y);
    }
    return st.spill[(sp - st.depth) * 64u];
}

struct OtRay {
    float idx, idy, idz, ox, oy, oz;  // approximate 1/d and o/d: own boxes are padded for it (mpt_hip.hip)
};
__device__ __forceinline__ OtRay ot_ray(F3 o, F3 d) {
    OtRay r;
    r.idx = __builtin_amdgcn_rcpf(d.x);
    r.idy = __builtin_amdgcn_rcpf(d.y);
    r.idz = __builtin_amdgcn_rcpf(d.z);
    r.ox = o.x * r.idx;
    r.oy = o.y * r.idy;
    r.oz = o.z * r.idz;
    return r;
}
// rays the closest-first walk does not take: a direction component that is (nearly) zero, or anything not finite
__device__ __forceinline__ bool ot_degenerate(F3 o, F3 d, float o_limit) {
    const float tiny = 9.5367431640625e-07f;  // 2^-20
    return !(fabsf(d.x) >= tiny && fabsf(d.y) >= tiny && fabsf(d.z) >= tiny && fabsf(d.x) <= 2.0f && fabsf(d.y) <= 2.0f &&
             fabsf(d.z) <= 2.0f && fabsf(o.x) <= o_limit && fabsf(o.y) <= o_limit && fabsf(o.z) <= o_limit);
}

struct OtNode {
    float4 lx, ly, lz, hx, hy, hz;
    uint4 ref;
};
__device__ __forceinline__ OtNode ot_load_node(const AccelDev& ac, LdsNodes lds, uint32_t n) {
    OtNode nd;
    if (n < ac.n_lds_nodes) {
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
        const float4* q = ac.nodes + 7u * (size_t)n;
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
__device__ __forceinline__ uint32_t ot_child_ref(const AccelDev& ac, LdsNodes lds, uint32_t n, uint32_t slot) {
    if (n < ac.n_lds_nodes) {
        const __attribute__((address_space(3))) uint32_t* q = (const __attribute__((address_space(3))) uint32_t*)(lds + 7u * n + 6u);
        return q[slot];
    }
    return ((const uint32_t*)(ac.nodes + 7u * (size_t)n + 6u))[slot];
}
// entry distance of the ray into one child box as a sort key: float bits with the child slot in the two low bits
// (t >= 0, so unsigned order = float order; the key rounds the distance DOWN by at most 3 ulp), or KEY_MISS
__device__ __forceinline__ uint32_t ot_box_key(const OtRay& r, float lx, float ly, float lz, float hx, float hy, float hz,
                                               uint32_t ref, float lim, uint32_t slot) {
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
    const bool hit = ref != MPT_OT_DONE && tn <= tf * 1.00000048f && tn <= lim;
    return hit ? ((__float_as_uint(tn) & ~3u) | slot) : (MPT_OT_KEY_MISS | slot);
}
__device__ __forceinline__ void ot_sort2(uint32_t& a, uint32_t& b) {
    const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
    a = lo;
    b = hi;
}
__device__ __forceinline__ float ot_cull_limit(float T, float eps_abs) { return T + (T * 9.765625e-4f + eps_abs); }

// One primitive against the ray — the reference's tests, PathTracing.h:120-176, bit for bit (as leaf_test in
// mpt_device.h), plus the tie flag.  `index` = position in the device primitive array.
__device__ __forceinline__ void ot_test_prim(const Prim3& pr, uint32_t index, F3 o, F3 d, float& T, int& W, bool& tie) {
    const float4 p0 = pr.p0, p1 = pr.p1, p2 = pr.p2;
    float tt = 0.0f;
    bool hit = false;
    if (prim_type(p0) == 1) {
        F3 v0 = f3(p0.x, p0.y, p0.z), e1 = f3(p1.x, p1.y, p1.z), e2 = f3(p2.x, p2.y, p2.z);
        F3 h = cross3(d, e2);
        float a = dot3(e1, h);
        if (fabsf(a) > 1e-5f) {
            float f = 1.0f / a;
            F3 s = o - v0;
            float u = f * dot3(s, h);
            if (u >= 0.0f && u <= 1.0f) {
                F3 q = cross3(s, e1);
                float v = f * dot3(d, q);
                if (v >= 0.0f && u + v <= 1.0f) {
                    tt = f * dot3(e2, q);
                    hit = tt > 0.0001f;
                }
            }
        }
    } else {
        F3 c = f3(p0.x, p0.y, p0.z);
        float radius = p1.x;
        F3 oc = o - c;
        float a = dot3(d, d);
        float b = dot3(oc, d);
        float cc = dot3(oc, oc) - radius * radius;
        float disc = b * b - a * cc;
        if (disc > 0.0f) {
            float sq = sqrtf(disc);
            tt = (-b - sq) / a;
            hit = tt > 0.0001f;
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
        const LdsNodes q = lds + ac.lds_always_off + 3u * k;
        const v4f a = q[0], b = q[1], c = q[2];
        Prim3 pr;
        pr.p0 = make_float4(a.x, a.y, a.z, a.w);
        pr.p1 = make_float4(b.x, b.y, b.z, b.w);
        pr.p2 = make_float4(c.x, c.y, c.z, c.w);
        if (COUNT) wc.prim_tests++;
        ot_test_prim(pr, __float_as_uint(b.y), o, d, T, W, tie);
    }
    const OtNode nd = ot_load_node(ac, lds, 0u);
    const float lim = ot_cull_limit(T, ac.eps_abs);
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
template <bool COUNT>
__device__ __forceinline__ void ot_walk(const AccelDev& ac, const SceneDev& sc, LdsNodes lds, const OtStack& st, F3 o, F3 d,
                                        const OtRay& r, bool active, float& T, int& W, bool& tie, bool& overflow,
                                        WorkCount& wc) {
    uint32_t cur = active ? 0u : MPT_OT_DONE;
    uint32_t sp = 0;
    for (;;) {
        while (cur < MPT_OT_LEAF) {
            const OtNode nd = ot_load_node(ac, lds, cur);
            const float lim = ot_cull_limit(T, ac.eps_abs);
            uint32_t k0 = ot_box_key(r, nd.lx.x, nd.ly.x, nd.lz.x, nd.hx.x, nd.hy.x, nd.hz.x, nd.ref.x, lim, 0u);
            uint32_t k1 = ot_box_key(r, nd.lx.y, nd.ly.y, nd.lz.y, nd.hx.y, nd.hy.y, nd.hz.y, nd.ref.y, lim, 1u);
            uint32_t k2 = ot_box_key(r, nd.lx.z, nd.ly.z, nd.lz.z, nd.hx.z, nd.hy.z, nd.hz.z, nd.ref.z, lim, 2u);
            uint32_t k3 = ot_box_key(r, nd.lx.w, nd.ly.w, nd.lz.w, nd.hx.w, nd.hy.w, nd.hz.w, nd.ref.w, lim, 3u);
            if (COUNT) {
                wc.node_visits++;
                wc.aabb_hits += (k0 < MPT_OT_KEY_MISS) + (k1 < MPT_OT_KEY_MISS) + (k2 < MPT_OT_KEY_MISS) + (k3 < MPT_OT_KEY_MISS);
                if (first_active_lane()) wc.node_iters++;
            }
            ot_sort2(k0, k1);
            ot_sort2(k2, k3);
            ot_sort2(k0, k2);
            ot_sort2(k1, k3);
            ot_sort2(k1, k2);
            const uint32_t parent = cur;
            if (k0 < MPT_OT_KEY_MISS) {
                const uint32_t s = k0 & 3u;
                cur = s == 0u ? nd.ref.x : s == 1u ? nd.ref.y : s == 2u ? nd.ref.z : nd.ref.w;
                if (k3 < MPT_OT_KEY_MISS) ot_push(st, sp, k3, parent, overflow);
                if (k2 < MPT_OT_KEY_MISS) ot_push(st, sp, k2, parent, overflow);
                if (k1 < MPT_OT_KEY_MISS) ot_push(st, sp, k1, parent, overflow);
            } else {
                cur = MPT_OT_DONE;
                while (sp > 0u) {
                    const uint2 e = ot_pop(st, sp);
                    if (__uint_as_float(e.x & ~3u) <= lim) {
                        cur = ot_child_ref(ac, lds, e.y, e.x & 3u);
                        break;
                    }
                }
            }
        }
        if (__ballot(cur != MPT_OT_DONE) == 0ull) break;
        if (cur != MPT_OT_DONE) {  // a leaf: primitives [first, first + count) in index order
            const uint32_t first = cur & 0x07FFFFFFu, count = ((cur >> 27) & 15u) + 1u;
            if (COUNT && first_active_lane()) wc.outer_iters++;
            for (uint32_t k = 0; k < count; ++k) {
                const Prim3 pr = load_prim(sc, lds, first + k);
                if (COUNT) {
                    if (first_active_lane()) wc.prim_iters++;
                }
                if (ac.n_always != 0u && prim_type(pr.p0) == 0) continue;  // spheres are on the always list
                if (COUNT) wc.prim_tests++;
                ot_test_prim(pr, first + k, o, d, T, W, tie);
            }
            const float lim = ot_cull_limit(T, ac.eps_abs);
            cur = MPT_OT_DONE;
            while (sp > 0u) {
                const uint2 e = ot_pop(st, sp);
                if (__uint_as_float(e.x & ~3u) <= lim) {
                    cur = ot_child_ref(ac, lds, e.y, e.x & 3u);
                    break;
                }
            }
        }
    }
}

// Final check: the reference's slab test (PathTracing.h:52-72, exact arithmetic, best t = +inf) on the winner's
// REFERENCE leaf box must pass, and the winner's t must not lie in front of that box.
__device__ __forceinline__ bool ot_final_check(const AccelDev& ac, const SceneDev& sc, LdsNodes lds, F3 o, F3 d, float T, int W) {
    const Prim3 pr = load_prim(sc, lds, (uint32_t)W);
    const uint32_t leaf = prim_ref_leaf(pr.p0);
    const float4 n0 = ac.refleaf[2u * (size_t)leaf], n1 = ac.refleaf[2u * (size_t)leaf + 1u];
    const float idx = 1.0f / d.x, idy = 1.0f / d.y, idz = 1.0f / d.z;
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
    ot_walk<COUNT>(ac, sc, lds, st, o, d, r, valid && need && !degenerate, T, W, tie, overflow, wc);
    if (tie) flags |= 2u;
    if (overflow) flags |= 8u;
    if (valid && flags == 0u && W >= 0 && !ot_final_check(ac, sc, lds, o, d, T, W)) flags |= 4u;
    if (valid && flags != 0u) {
        uint32_t node = 0;
        T = INFINITY;
        W = -1;
        closest_hit_resume<COUNT, false, false>(sc, lds, o, d, node, T, W, 0xFFFFFFFFu, wc);
    }
}

__device__ __forceinline__ void ot_stage(const SceneDev& sc, const AccelDev& ac, float4* lds) {
    const uint32_t n4 = ac.n_lds_nodes * 7u, p4 = sc.n_lds_prims * 3u;
    for (uint32_t i = threadIdx.x; i < n4; i += blockDim.x) lds[i] = ac.nodes[i];
    for (uint32_t i = threadIdx.x; i < ac.n_always * 3u; i += blockDim.x) lds[ac.lds_always_off + i] = ac.always[i];
    for (uint32_t i = threadIdx.x; i < p4; i += blockDim.x) lds[sc.lds_prim_off + i] = sc.prims[i];
    for (uint32_t i = threadIdx.x; i < sc.n_lds_mats * 2u; i += blockDim.x) lds[sc.lds_mat_off + i] = sc.mats[i];
    __syncthreads();
}
__device__ __forceinline__ OtStack ot_stack(const AccelDev& ac, float4* lds_raw, uint32_t wave_global) {
    OtStack st;
    const uint32_t lane = threadIdx.x & 63u, wave_local = threadIdx.x >> 6;
    st.lds = (LdsStack)(lds_raw + (ac.lds_stack_off >> 4)) + wave_local * ac.stack_depth * 64u + lane;
    st.spill = ac.spill + (size_t)wave_global * MPT_OT_SPILL * 64u + lane;
    st.depth = ac.stack_depth;
    return st;
}

// ---- the pipeline kernel ------------------------------------------------------------------------------------------
template <bool COUNT>
__global__ __launch_bounds__(MPT_OT_THREADS, MPT_OT_WAVES) void k_ordered(PassParams pp, AccelDev ac, WaveRings ring,
                                                                          uint32_t wl_block, uint32_t wl_min, uint32_t wl_div,
                                                                          uint32_t walk_now_min) {
    extern __shared__ float4 lds_raw[];
    ot_stage(pp.scene, ac, lds_raw);
    const LdsNodes lds = (LdsNodes)lds_raw;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t total_paths = pp.desc->total_paths;
    const uint32_t wave_id = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    const OtStack st = ot_stack(ac, lds_raw, wave_id);
    const uint32_t wbase = wave_id * (MPT_OT_RINGS * MPT_WL_RING);
    uint32_t cntR = 0, cntM = 0, cntE = 0;           // wave-uniform ring fills (the rings are stacks: newest first)
    uint32_t cur = 0, end = 0;
    const uint32_t n_tiles = total_paths / (pp.S * 64u);
    const uint32_t waves_per_group = (n_waves + MPT_NGROUP - 1u) / MPT_NGROUP;
    uint32_t grp = blockIdx.x & (MPT_NGROUP - 1u);
    uint32_t seen = 0;
    bool exhausted = false;
    uint32_t n_rays = 0, n_paths = 0, n_flagged = 0, n_parked = 0;
    WorkCount wc = {};
    for (;;) {
        // ---- step choice ------------------------------------------------------------------------------------------
        uint32_t takeR = 0, takeM = 0, takeE = 0;
        bool primary = false, walk_all = false;
        if (cntE >= 64u) takeE = 64u;
        else if (cntM >= 64u) takeM = 64u;
        else if (cntR >= 64u) takeR = 64u;
        else {
            if (!exhausted && cur == end) {  // guided self-scheduling of path ids, as k_wavelocal (mpt_kernels.h)
                uint32_t k = 0, blk = 0, rend = 0;
                bool got = false;
                if (lane == 0) {
                    for (uint32_t t = 0; t < MPT_NGROUP && !got; ++t) {
                        const uint32_t re = range_paths(n_tiles, pp.S, grp);
                        const uint32_t left = seen < re ? re - seen : 0u;
                        blk = (left / (wl_div * waves_per_group)) & ~63u;
                        blk = blk < wl_min ? wl_min : (blk > wl_block ? wl_block : blk);
                        if (t > 0u) blk = wl_min;
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
            if (!exhausted) {
                primary = true;
            } else if (cntM + cntR != 0u) {  // drain: tree rays first, topped up with fresh rays; everything walks now
                takeM = cntM;
                takeR = cntR < 64u - takeM ? cntR : 64u - takeM;
                walk_all = true;
            } else if (cntE != 0u) {
                takeE = cntE;
            } else {
                break;
            }
        }
        // ---- rays of the step ---------------------------------------------------------------------------------------
        PathState ps;
        PathRngDev g;
        bool valid = false, from_m = false;
        float T = INFINITY;
        int W = -1;
        if (primary) {
            ps.path = range_chunk_to_path_chunk(pp, cur >> 6, grp) * 64u + lane;
            cur += 64u;
            uint32_t px, py, sidx;
            if (path_to_pixel(pp, ps.path, px, py, sidx)) {
                gen_primary(pp, px, py, pp.sample_begin + sidx, ps, g);
                valid = true;
                n_paths++;
            }
        } else {
            uint32_t at = 0;
            if (lane < takeM) {
                at = wbase + MPT_WL_RING + (cntM - takeM + lane);
                from_m = true;
                valid = true;
            } else if (lane < takeM + takeR) {
                at = wbase + (cntR - takeR + (lane - takeM));
                valid = true;
            } else if (lane < takeE) {
                at = wbase + 2u * MPT_WL_RING + (cntE - takeE + lane);
                valid = true;
            }
            cntM -= takeM;
            cntR -= takeR;
            cntE -= takeE;
            if (valid) {
                const float4 a = ring.od[at], b = ring.dt[at], cc = ring.tl[at];
                const uint4 ia = ring.ia[at];
                ps.o = f3(a.x, a.y, a.z);
                ps.d = f3(a.w, b.x, b.y);
                ps.thr = f3(b.z, b.w, cc.x);
                ps.L = f3(cc.y, cc.z, cc.w);
                ps.La = __uint_as_float(ia.y);
                ps.path = ia.x;
                ps.bounce = ia.w >> 27;
                g.pixel = ia.z;
                g.sample = ia.w & 0x07FFFFFFu;
                g.lit_seed = 0;
                if (pp.sp.rng_mode == 0) g.lit_seed = pcg_hash(pcg_hash(pp.pixel_seed[g.pixel]));
                if (from_m) {
                    const uint4 tv = ring.tv[at];
                    T = __uint_as_float(tv.x);
                    W = (int)tv.y;
                }
            }
        }
        bool flagged = false, parked = false, finished = false;
        if (takeE != 0u) {
            // ---- reference-order walk (PathTracing.h:75-204 as closest_hit_resume restates it) ----------------------
            if (valid) {
                uint32_t node = 0;
                T = INFINITY;
                W = -1;
                closest_hit_resume<COUNT, false, false>(pp.scene, lds, ps.o, ps.d, node, T, W, 0xFFFFFFFFu, wc);
                finished = true;
            }
        } else {
            bool tie = false, overflow = false, need = from_m;
            const OtRay r = ot_ray(ps.o, ps.d);
            if (valid && !from_m) {
                if (ot_degenerate(ps.o, ps.d, ac.o_limit)) flagged = true;
                else ot_top_test<COUNT>(ac, lds, ps.o, ps.d, r, T, W, tie, need, wc);
            }
            const bool wants = valid && need && !flagged;
            const bool walk_now = walk_all || takeM != 0u || (uint32_t)__popcll(__ballot(wants)) >= walk_now_min;
            if (walk_now) {
                ot_walk<COUNT>(ac, pp.scene, lds, st, ps.o, ps.d, r, wants, T, W, tie, overflow, wc);
            } else {
                parked = wants;
            }
            if (valid && !parked && !flagged) {
                flagged = tie || overflow;
                if (!flagged && W >= 0) flagged = !ot_final_check(ac, pp.scene, lds, ps.o, ps.d, T, W);
                finished = !flagged;
            }
        }
        // ---- shading of the rays whose closest hit is known ---------------------------------------------------------
        bool alive = false;
        if (finished) {
            n_rays++;
            alive = shade_bounce(pp.scene, lds, pp.sp, g, ps, T, W);
            if (!alive) store_slot(pp.slots, ps.path, clamp01(ps.L.x), clamp01(ps.L.y), clamp01(ps.L.z), clamp01(ps.La));
        }
        n_flagged += flagged ? 1u : 0u;
        n_parked += parked ? 1u : 0u;
        // ---- wave64 compaction into the rings: ballot + mbcnt prefix, ring fills stay wave-uniform -------------------
        const unsigned long long am = __ballot(alive), pm = __ballot(parked), fm = __ballot(flagged);
        if ((am | pm | fm) != 0ull) {
            uint32_t at = 0;
            if (alive) at = wbase + cntR + wave_rank(am);
            if (parked) at = wbase + MPT_WL_RING + cntM + wave_rank(pm);
            if (flagged) at = wbase + 2u * MPT_WL_RING + cntE + wave_rank(fm);
            if (alive || parked || flagged) {
                ring.od[at] = make_float4(ps.o.x, ps.o.y, ps.o.z, ps.d.x);
                ring.dt[at] = make_float4(ps.d.y, ps.d.z, ps.thr.x, ps.thr.y);
                ring.tl[at] = make_float4(ps.thr.z, ps.L.x, ps.L.y, ps.L.z);
                ring.ia[at] = make_uint4(ps.path, __float_as_uint(ps.La), g.pixel, g.sample | (ps.bounce << 27));
                if (parked) ring.tv[at] = make_uint4(__float_as_uint(T), (uint32_t)W, 0u, 0u);
            }
            cntR += (uint32_t)__popcll(am);
            cntM += (uint32_t)__popcll(pm);
            cntE += (uint32_t)__popcll(fm);
        }
        const uint32_t worst = cntR > cntM ? (cntR > cntE ? cntR : cntE) : (cntM > cntE ? cntM : cntE);
        if (worst > MPT_WL_RING) pp.desc->overflow = 1u;  // cannot happen: a step never adds more rays than it took + 64
    }
    flush_stats<COUNT>(pp.desc, n_rays, n_paths, wc);
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
