// mpt_lbvh.h — BVH construction ON THE GPU: a binary tree over the primitives, by one of three builders —
//   sah   (default) top-down binned SAH, level by level (mpt_sah.h): the host binned builder's tree, 1 M primitives in 4 ms
//   ploc  63-bit Morton codes of the primitive centroids, radix sort, then rounds of nearest-neighbour clustering
//   lbvh  the same sort, then Karras' parallel radix-tree construction and a bottom-up refit
// — collapsed to leaves of <= 2 primitives (MPT_LBVH_LEAF = 1..8; the closest-first pipeline tests every primitive of a leaf
// it enters, so small leaves pay there: bunny x20 12.5 / 13.2 / 12.9 / 12.5 Grays/s with 1 / 2 / 3 / 4) and written in the
// REFERENCE's buffer format (SURVEY App. D buf 0 / buf 6):
//   node = (bmin.xyz, bits(leftFirst)) (bmax.xyz, bits(count));  count > 0: leaf, primitiveIndices[leftFirst ..
//   leftFirst+count);  count <= 0: internal, left child = leftFirst, right child = -count;  root = node 0.
// It stands where the reference has Scene::buildBVH / buildBVHRecursive (R/Scene/Scene.h:71-93,195-317: a sequential
// full-sweep SAH over std::sort, 8.2 s for 1 M primitives on one core); primitive boxes follow Scene.h:199-209
// (sphere: centre -+ radius; triangle: min / max of the vertices) and parent boxes are unions of child boxes, so boxes
// nest exactly (which the closest-first pipeline requires).  The tree topology is not a parity target: the
// oracle renders the same image from these arrays as the HIP pipelines do (tests/test_gpu_lbvh.py).
#pragma once
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "mpt_sah.h"

#define MPT_LBVH_LEAF_MAX 8u   // what the reference's builder allows (R/Scene/Scene.h:223) and a device leaf record holds twice over

namespace mpt_lbvh {

__device__ __forceinline__ int f2ord(float f) {  // order-preserving float -> int (for atomicMin / atomicMax)
    int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }

// primitive boxes (Scene.h:199-209) + bounds of the box centres
__global__ void k_boxes(const float4* prims, uint32_t n, float4* blo, float4* bhi, int* cb /* [6]: min xyz, max xyz as ordered ints */) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, c[3] = {0, 0, 0};
    const bool valid = i < n;
    if (valid) {
        const float4 p0 = prims[3 * (size_t)i], p1 = prims[3 * (size_t)i + 1], p2 = prims[3 * (size_t)i + 2];
        if ((int)p0.w == 0) {
            lo[0] = p0.x - p1.x; lo[1] = p0.y - p1.x; lo[2] = p0.z - p1.x;
            hi[0] = p0.x + p1.x; hi[1] = p0.y + p1.x; hi[2] = p0.z + p1.x;
        } else {
            lo[0] = fminf(p0.x, fminf(p1.x, p2.x)); lo[1] = fminf(p0.y, fminf(p1.y, p2.y)); lo[2] = fminf(p0.z, fminf(p1.z, p2.z));
            hi[0] = fmaxf(p0.x, fmaxf(p1.x, p2.x)); hi[1] = fmaxf(p0.y, fmaxf(p1.y, p2.y)); hi[2] = fmaxf(p0.z, fmaxf(p1.z, p2.z));
        }
        // The reference's slab test never enters a box of zero thickness (PathTracing.h:68 rejects tMax <= tMin): a leaf of
        // coplanar, axis-aligned triangles — the two halves of a Cornell-box wall, which these builders' small leaves pair
        // up where the reference's leaves of eight mix orientations — would be invisible.  So no primitive box is thinner
        // than 2^-16 of its own size / position (host binned builder: the same rule, Scene.cpp).
        {
            const float ext = fmaxf(hi[0] - lo[0], fmaxf(hi[1] - lo[1], hi[2] - lo[2]));
            for (int a = 0; a < 3; ++a) {
                const float pad = fmaxf(fmaxf(fabsf(lo[a]), fabsf(hi[a])), ext) * 1.52587890625e-05f;
                if (hi[a] - lo[a] < pad) {
                    lo[a] -= pad;
                    hi[a] += pad;
                }
            }
        }
        blo[i] = make_float4(lo[0], lo[1], lo[2], 0.0f);
        bhi[i] = make_float4(hi[0], hi[1], hi[2], 0.0f);
        for (int a = 0; a < 3; ++a) c[a] = 0.5f * lo[a] + 0.5f * hi[a];
    }
    if (!cb) return;   // (the top-down builder has no use for the bounds of the centroids: 94 k atomics on six words, 0.9 ms at 1 M)
    // wave reduction, then one atomic per wave and component
    for (int a = 0; a < 3; ++a) {
        float mn = valid && isfinite(c[a]) ? c[a] : INFINITY, mx = valid && isfinite(c[a]) ? c[a] : -INFINITY;
        for (int off = 32; off > 0; off >>= 1) {
            mn = fminf(mn, __shfl_xor(mn, off));
            mx = fmaxf(mx, __shfl_xor(mx, off));
        }
        if ((threadIdx.x & 63u) == 0) {
            atomicMin(&cb[a], f2ord(mn));
            atomicMax(&cb[3 + a], f2ord(mx));
        }
    }
}

__device__ __forceinline__ unsigned long long spread21(unsigned long long v) {  // 21 bits -> every third bit
    v &= 0x1FFFFFull;
    v = (v | v << 32) & 0x1F00000000FFFFull;
    v = (v | v << 16) & 0x1F0000FF0000FFull;
    v = (v | v << 8) & 0x100F00F00F00F00Full;
    v = (v | v << 4) & 0x10C30C30C30C30C3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}
__global__ void k_morton(const float4* blo, const float4* bhi, uint32_t n, const int* cb, unsigned long long* keys, uint32_t* vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long code = 0;
    // cubic cells: the same scale on all three axes (the largest extent of the centroid bounds).  Per-axis scaling made
    // the cells of scene.xml-like scenes — a ground sphere centred 10^4 below everything else — 125 times taller than
    // wide, so that Morton neighbours were columns of triangles far apart in y (leaves of 3x the area, 3x the tests).
    const float ext = fmaxf(ord2f(cb[3]) - ord2f(cb[0]), fmaxf(ord2f(cb[4]) - ord2f(cb[1]), ord2f(cb[5]) - ord2f(cb[2])));
    for (int a = 0; a < 3; ++a) {
        const float mn = ord2f(cb[a]);
        const float lo = a == 0 ? blo[i].x : a == 1 ? blo[i].y : blo[i].z, hi = a == 0 ? bhi[i].x : a == 1 ? bhi[i].y : bhi[i].z;
        const float c = 0.5f * lo + 0.5f * hi;
        float u = ext > 0.0f ? (c - mn) / ext : 0.0f;
        u = isfinite(u) ? fminf(fmaxf(u, 0.0f), 1.0f) : 0.0f;
        const unsigned long long q = (unsigned long long)fminf(u * 2097152.0f, 2097151.0f);
        code |= spread21(q) << (2 - a);
    }
    keys[i] = code;
    vals[i] = i;
}

// Karras 2012: internal node i of the radix tree over the sorted keys; ties are broken by position
__device__ __forceinline__ int delta(const unsigned long long* keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const unsigned long long a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((unsigned)i ^ (unsigned)j);
    return __clzll((long long)(a ^ b));
}
// node ids: internal k in [0, n-1), leaf (single primitive) p -> (n-1) + p
__global__ void k_hierarchy(const unsigned long long* keys, int n, int2* child, int* parent, int2* range) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const int left = lo == gamma ? (n - 1) + gamma : gamma;
    const int right = hi == gamma + 1 ? (n - 1) + gamma + 1 : gamma + 1;
    child[i] = make_int2(left, right);
    range[i] = make_int2(lo, hi);
    parent[left] = i;
    parent[right] = i;
    if (i == 0) parent[0] = -1;
}

// What one thread hands to another inside a kernel (k_refit here, k_own_tree in mpt_devbuild.h) goes through agent-scope
// (sc1) stores and loads — past the L1 and the XCD's L2 — and handoff_release() orders them in front of the arrival counter:
//   * the ORDERING MECHANISM is the explicit `s_waitcnt vmcnt(0)`: on gfx9 vmcnt covers stores too, so every sc1 store this
//     thread has issued has been acknowledged at the coherence point before the relaxed agent-scope atomicAdd is issued;
//   * the workgroup-scope release fence in front of it is there for the COMPILER only (it may not sink the stores below
//     it); in hardware it is just `s_waitcnt lgkmcnt(0)` and synchronises with nobody outside the workgroup;
//   * the consumer is the thread whose atomicAdd returns 1: its sc1 loads are issued after that value has come back (the
//     branch needs it), and sc1 loads do not hit in the non-coherent L1 / L2 lines.
// An agent-scope release (or __threadfence()) would be the textbook form, but it writes the XCD's whole L2 back — per
// thread, per level: 5.2 ms for a 1 M-primitive refit against 0.2.  tests/test_build_asm.py checks in the built code
// object that a `s_waitcnt vmcnt(0)` stands between the last store and the atomic of both kernels.
__device__ __forceinline__ void handoff_release() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ float4 ld4(const float4* p) {
    const float* f = (const float*)p;
    return make_float4(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                       __hip_atomic_load(f + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                       __hip_atomic_load(f + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), 0.0f);
}
__device__ __forceinline__ void st_agent(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ld_agent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st4(float4* p, float x, float y, float z) {
    float* f = (float*)p;
    st_agent(f, x);
    st_agent(f + 1, y);
    st_agent(f + 2, z);
    st_agent(f + 3, 0.0f);
}
// bottom-up boxes: the second thread to arrive at a node computes it
__global__ void k_refit(const uint32_t* vals, const float4* blo, const float4* bhi, int n, const int2* child, const int* parent,
                        float4* nlo, float4* nhi, int* arrived) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t prim = vals[p];
    const float4 l = blo[prim], h = bhi[prim];
    st4(nlo + (n - 1) + p, l.x, l.y, l.z);
    st4(nhi + (n - 1) + p, h.x, h.y, h.z);
    int node = parent[(n - 1) + p];
    while (node >= 0) {
        handoff_release();
        if (atomicAdd(&arrived[node], 1) == 0) return;  // the sibling subtree is not finished yet
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int2 c = child[node];
        const float4 a0 = ld4(nlo + c.x), a1 = ld4(nhi + c.x), b0 = ld4(nlo + c.y), b1 = ld4(nhi + c.y);
        st4(nlo + node, fminf(a0.x, b0.x), fminf(a0.y, b0.y), fminf(a0.z, b0.z));
        st4(nhi + node, fmaxf(a1.x, b1.x), fmaxf(a1.y, b1.y), fmaxf(a1.z, b1.z));
        node = parent[node];
    }
}

// which radix-tree nodes become nodes of the output: an internal node spanning more than LEAF primitives stays internal;
// a node (internal or single primitive) spanning <= LEAF primitives whose parent spans more becomes a leaf
__device__ __forceinline__ int span(const int2* range, int n, int node) { return node >= n - 1 ? 1 : range[node].y - range[node].x + 1; }
__global__ void k_mark(int n, int leaf_max, const int* parent, const int2* range, uint32_t* keep) {
    const int node = blockIdx.x * blockDim.x + threadIdx.x;
    if (node >= 2 * n - 1) return;
    const int sz = span(range, n, node), par = parent[node];
    const bool k = sz > leaf_max || par < 0 || span(range, n, par) > leaf_max;
    keep[node] = k ? 1u : 0u;
}
__global__ void k_emit(int n, int leaf_max, const int2* child, const int2* range, const uint32_t* keep, const uint32_t* index, const float4* nlo,
                       const float4* nhi, const uint32_t* vals, float4* bvh_out, int* prim_idx_out) {
    const int node = blockIdx.x * blockDim.x + threadIdx.x;
    if (node < n) prim_idx_out[node] = (int)vals[node];  // primitiveIndices = the sorted order
    if (node >= 2 * n - 1 || !keep[node]) return;
    const uint32_t at = index[node];
    const int sz = span(range, n, node);
    int lf, cnt;
    if (sz > leaf_max) {
        lf = (int)index[child[node].x];
        cnt = -(int)index[child[node].y];
    } else {
        lf = node >= n - 1 ? node - (n - 1) : range[node].x;
        cnt = sz;
    }
    const float4 lo = nlo[node], hi = nhi[node];
    bvh_out[2 * (size_t)at] = make_float4(lo.x, lo.y, lo.z, __int_as_float(lf));
    bvh_out[2 * (size_t)at + 1] = make_float4(hi.x, hi.y, hi.z, __int_as_float(cnt));
}


// ---- PLOC: parallel locally-ordered clustering (Meister & Bittner 2018) ------------------------------------------------------
// The quality pass of the GPU build.  The Karras tree above splits wherever the Morton codes first differ; PLOC instead
// builds the tree bottom-up by merging, round after round, every pair of clusters that are each other's best partner —
// the one with the smallest merged box among the MPT_PLOC_RADIUS neighbours on either side in Morton order.  The
// result is a binary tree of agglomerative-clustering quality (measured: DESIGN.md 5) in ~30 rounds of three small
// kernels.  Its subtrees are not contiguous in Morton order, so the primitives are re-ordered depth-first afterwards
// and the node ids renumbered (root = internal node 0, leaf at depth-first position p = (n-1) + p): from there on the
// tree looks exactly like a Karras tree to everything downstream (k_mark, k_emit, mpt_devbuild.h).
#ifndef MPT_PLOC_RADIUS
#define MPT_PLOC_RADIUS 16
#endif
struct PlocState {
    uint32_t n_clusters;     // clusters alive in the current round
    uint32_t next_internal;  // internal nodes created so far (creation order: 0, 1, ...)
    uint32_t merges;         // of the current round
};
__device__ __forceinline__ float union_half_area(float4 al, float4 ah, float4 bl, float4 bh) {
    const float dx = fmaxf(ah.x, bh.x) - fminf(al.x, bl.x), dy = fmaxf(ah.y, bh.y) - fminf(al.y, bl.y), dz = fmaxf(ah.z, bh.z) - fminf(al.z, bl.z);
    return dx * dy + dy * dz + dz * dx;
}
__global__ void k_ploc_nn(const PlocState* st, const uint32_t* C, const float4* nlo, const float4* nhi, uint32_t* nn) {
    const uint32_t N = st->n_clusters, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N || N < 2u) return;
    const float4 al = nlo[C[i]], ah = nhi[C[i]];
    const uint32_t lo = i > MPT_PLOC_RADIUS ? i - MPT_PLOC_RADIUS : 0u, hi = i + MPT_PLOC_RADIUS < N - 1u ? i + MPT_PLOC_RADIUS : N - 1u;
    float best = INFINITY;
    uint32_t bj = i == lo ? i + 1u : lo;
    for (uint32_t j = lo; j <= hi; ++j) {
        if (j == i) continue;
        float a = union_half_area(al, ah, nlo[C[j]], nhi[C[j]]);
        if (!(a < 3.0e38f)) a = 3.0e38f;   // (a box with a NaN or an infinite side: still mergeable, last)
        if (a < best) best = a, bj = j;   // ties: the smaller j, on both sides of a pair
    }
    nn[i] = bj;
}
__global__ void k_ploc_flags(const PlocState* st, const uint32_t* nn, uint32_t* merged, uint32_t* valid) {
    const uint32_t N = st->n_clusters, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    uint32_t m = 0, v = 1;
    if (N >= 2u) {
        const uint32_t j = nn[i];
        if (nn[j] == i) {   // mutual: the lower index carries the new node, the higher one disappears
            m = i < j ? 1u : 0u;
            v = i < j ? 1u : 0u;
        }
    }
    merged[i] = m;
    valid[i] = v;
}
__global__ void k_ploc_merge(PlocState* st, const uint32_t* C, const uint32_t* nn, const uint32_t* merged, const uint32_t* valid, const uint32_t* mslot,
                             const uint32_t* vpos, int n, uint32_t* Cout, int2* child, int* parent, uint32_t* size, float4* nlo, float4* nhi) {
    const uint32_t N = st->n_clusters, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    if (valid[i]) {
        uint32_t id = C[i];
        if (merged[i]) {
            const uint32_t a = C[i], b = C[nn[i]];
            id = st->next_internal + mslot[i];
            child[id] = make_int2((int)a, (int)b);
            parent[id] = -1;
            parent[a] = (int)id;
            parent[b] = (int)id;
            size[id] = size[a] + size[b];
            const float4 al = nlo[a], ah = nhi[a], bl = nlo[b], bh = nhi[b];
            nlo[id] = make_float4(fminf(al.x, bl.x), fminf(al.y, bl.y), fminf(al.z, bl.z), 0.0f);
            nhi[id] = make_float4(fmaxf(ah.x, bh.x), fmaxf(ah.y, bh.y), fmaxf(ah.z, bh.z), 0.0f);
        }
        Cout[vpos[i]] = id;
    }
    (void)n;
}
__global__ void k_ploc_advance(PlocState* st, const uint32_t* merged, const uint32_t* mslot, const uint32_t* valid, const uint32_t* vpos) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const uint32_t N = st->n_clusters;
    if (N < 2u) return;
    const uint32_t nm = mslot[N - 1] + merged[N - 1], nv = vpos[N - 1] + valid[N - 1];
    st->next_internal += nm;
    st->n_clusters = nv;
    st->merges = nm;
}
__global__ void k_ploc_init(int n, const uint32_t* vals, const float4* blo, const float4* bhi, uint32_t* C, uint32_t* size, float4* nlo, float4* nhi, int* parent,
                            PlocState* st) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p == 0) {
        st->n_clusters = (uint32_t)n;
        st->next_internal = 0u;
        st->merges = 0u;
    }
    if (p >= n) return;
    const uint32_t id = (uint32_t)(n - 1 + p);   // during the clustering: internal ids in creation order, leaf = (n-1) + Morton position
    C[p] = id;
    size[id] = 1u;
    nlo[id] = blo[vals[p]];
    nhi[id] = bhi[vals[p]];
    parent[id] = -1;
}
// depth-first position of every node's first primitive: walking up, every time the node hangs on the right the left
// sibling's primitives come first
__global__ void k_ploc_first(int n, const int2* child, const int* parent, const uint32_t* size, uint32_t* first) {
    const int node = blockIdx.x * blockDim.x + threadIdx.x;
    if (node >= 2 * n - 1) return;
    uint32_t f = 0;
    for (int y = node, p = parent[node]; p >= 0; y = p, p = parent[p])
        if (y == child[p].y) f += size[child[p].x];
    first[node] = f;
}
// renumber: leaf (n-1) + m -> (n-1) + first.  Internal id k -> (n-2) - k when the tree was made bottom-up (creation order, the
// root last).  The top-down builder numbers its nodes by atomics, in no reproducible order: there an inner node takes the
// position of the gap between two primitives that it splits at (first + primitives of its left child - 1: one gap per inner
// node), with the root's number and 0 exchanged — the same tree gets the same numbers on every run.  Node 0 is the root either way.
template <bool TOP_DOWN>
__device__ __forceinline__ int ploc_new_id(int n, int id, const uint32_t* first, const int2* child, const uint32_t* size) {
    if (id >= n - 1) return (n - 1) + (int)first[id];
    if (!TOP_DOWN) return (n - 2) - id;
    const int s = (int)(first[id] + size[child[id].x]) - 1, r = (int)size[child[0].x] - 1;
    return s == r ? 0 : (s == 0 ? r : s);
}
template <bool TOP_DOWN>
__global__ void k_ploc_renumber(int n, const int2* child, const int* parent, const uint32_t* size, const uint32_t* first, const float4* nlo, const float4* nhi,
                                const uint32_t* vals, int2* child2, int* parent2, int2* range2, float4* nlo2, float4* nhi2, uint32_t* vals2) {
    const int node = blockIdx.x * blockDim.x + threadIdx.x;
    if (node >= 2 * n - 1) return;
    const int id2 = ploc_new_id<TOP_DOWN>(n, node, first, child, size);
    parent2[id2] = parent[node] < 0 ? -1 : ploc_new_id<TOP_DOWN>(n, parent[node], first, child, size);
    nlo2[id2] = nlo[node];
    nhi2[id2] = nhi[node];
    if (node < n - 1) {
        child2[id2] = make_int2(ploc_new_id<TOP_DOWN>(n, child[node].x, first, child, size), ploc_new_id<TOP_DOWN>(n, child[node].y, first, child, size));
        range2[id2] = make_int2((int)first[node], (int)(first[node] + size[node] - 1u));
    } else {
        vals2[first[node]] = vals[node - (n - 1)];
    }
}

// builder "sah": the primitives as items of mpt_sah.h (id = the leaf (n-1) + p, one primitive each) ...
__global__ void k_prim_items(int n, const float4* blo, const float4* bhi, float4* it_lo, float4* it_hi, uint32_t* vals) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const float4 l = blo[p], h = bhi[p];
    it_lo[p] = make_float4(l.x, l.y, l.z, __int_as_float(n - 1 + p));
    it_hi[p] = make_float4(h.x, h.y, h.z, __int_as_float(1));
    vals[p] = (uint32_t)p;
}
// ... and its tree (inner node k = SAH node k, made top-down: parents before children, the root first) in the arrays the
// clustering leaves: children, parents, primitives below every node, boxes
// first[node]: the depth-first position of the node's first primitive (k_ploc_first), here straight from the SAH, which knows where every
// node's items start in its final item order (mpt_sah.h SahFirst)
__global__ void k_sah_to_radix(int n, const int2* s_child, const float4* s_lo, const float4* s_hi, const float4* blo, const float4* bhi, const uint32_t* first_inner,
                               const uint32_t* first_item, int2* child, int* parent, uint32_t* size, float4* nlo, float4* nhi, uint32_t* first) {
    const int node = blockIdx.x * blockDim.x + threadIdx.x, top = 2 * n - 1;
    if (node >= top) return;
    if (node == 0) parent[0] = -1;
    first[node] = node < n - 1 ? first_inner[node] : first_item[node];
    if (node < n - 1) {
        const int2 c = s_child[node];
        const int x = c.x >= top ? c.x - top : c.x, y = c.y >= top ? c.y - top : c.y;
        child[node] = make_int2(x, y);
        parent[x] = node;
        parent[y] = node;
        const float4 l = s_lo[node], h = s_hi[node];
        size[node] = (uint32_t)__float_as_int(l.w);
        nlo[node] = make_float4(l.x, l.y, l.z, 0.0f);
        nhi[node] = make_float4(h.x, h.y, h.z, 0.0f);
    } else {
        size[node] = 1u;
        nlo[node] = blo[node - (n - 1)];
        nhi[node] = bhi[node - (n - 1)];
    }
}

// The same with the spheres kept OUT of the SAH.  A ground sphere of radius 1e4 makes the root's box a million times the
// scene's, every probability the SAH computes is then relative to that box, and its top splits are about the spheres, not
// about where rays go: scene.xml renders 4 % faster (18.9 vs 19.7 ms) when the tree is root -> (the spheres, the SAH tree of
// the triangles) than on the "optimal" tree, and the closest-first kernel keeps its spheres on a list anyway.  So when there
// are 1..MPT_LBVH_HOIST_MAX spheres (and >= 3 triangles) the SAH runs over the triangles alone and the spheres hang under a new
// root as a chain: ids 0 = the root, 1 .. ns-1 = the chain, ns + k = SAH node k.
// Round 4: the same goes for ANY item whose box dwarfs the rest — a ground quad of two 10^4-unit triangles stretches the root box
// exactly as the ground sphere does.  An item is "huge" when the binary exponent of its box's largest extent is >= 11 above
// the MEDIAN item's (>= 2^10 x the median extent, by binades: a 256-bin histogram of the exponents, no sort); up to
// MPT_LBVH_HOIST_MAX spheres + huge triangles hang under the root (if there are more, the spheres alone do, as before).
#define MPT_LBVH_HOIST_MAX 16u
#define MPT_LBVH_HUGE_BINADES 11
struct HoistState {
    uint32_t hist[256];   // exponent of the largest box extent, per item
    uint32_t n_sph, n_huge, mode, thresh;   // mode: 0 nothing hangs under the root, 1 the spheres, 2 spheres and huge triangles; thresh: exponent from which a box is huge
};
__device__ __forceinline__ uint32_t box_exponent(float4 l, float4 h) {
    const float e = fmaxf(fmaxf(h.x - l.x, h.y - l.y), h.z - l.z);
    return e > 0.0f && e < INFINITY ? (__float_as_uint(e) >> 23) & 255u : 0u;   // (empty, NaN and infinite boxes: bin 0, never huge)
}
__global__ void k_hoist_hist(int n, const float4* prims, const float4* blo, const float4* bhi, HoistState* st) {
    __shared__ uint32_t h[256];
    for (uint32_t q = threadIdx.x; q < 256u; q += blockDim.x) h[q] = 0u;
    __syncthreads();
    // (grid-stride over a bounded grid: nearly all items share two or three exponents, and a workgroup per 256 items made 12 k same-address
    //  atomics of the merge below — 47 us for 1 M items)
    uint32_t nsph = 0;
    for (int base = blockIdx.x * blockDim.x; base < n; base += gridDim.x * blockDim.x) {
        const int p = base + (int)threadIdx.x;
        bool sphere = false;
        if (p < n) {
            sphere = (int)prims[3 * (size_t)p].w != 1;
            atomicAdd(&h[box_exponent(blo[p], bhi[p])], 1u);
        }
        nsph += (uint32_t)__popcll(__ballot(sphere));
    }
    if ((threadIdx.x & 63u) == 0u && nsph != 0u) atomicAdd(&st->n_sph, nsph);
    __syncthreads();
    for (uint32_t q = threadIdx.x; q < 256u; q += blockDim.x)
        if (h[q]) atomicAdd(&st->hist[q], h[q]);
}
__global__ void k_hoist_median(int n, HoistState* st) {   // one thread: the median exponent -> the threshold
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t run = 0, med = 0;
    for (uint32_t q = 0; q < 256u; ++q) {
        run += st->hist[q];
        if (2u * run >= (uint32_t)n) {
            med = q;
            break;
        }
    }
    st->thresh = med == 0u ? 256u : med + MPT_LBVH_HUGE_BINADES;   // (median box empty or degenerate: nothing is huge)
    uint32_t huge = 0;
    for (uint32_t q = st->thresh; q < 256u; ++q) huge += st->hist[q];
    st->n_huge = huge;   // (counts huge spheres too: an upper bound of the huge triangles, exact when the spheres are small)
}
// flag = 1: the item takes part in the SAH; 0: it hangs under the root
__global__ void k_tri_flags(int n, const float4* prims, const float4* blo, const float4* bhi, HoistState* st, uint32_t* flag, uint32_t* n_sph_out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const bool with_huge = st->n_sph + st->n_huge <= MPT_LBVH_HOIST_MAX && st->n_huge != 0u;
    if (p < n) {
        const bool tri = (int)prims[3 * (size_t)p].w == 1;
        const bool huge = with_huge && box_exponent(blo[p], bhi[p]) >= st->thresh;
        flag[p] = tri && !huge ? 1u : 0u;
    }
    if (p == n) {
        flag[n] = 0u;
        st->mode = with_huge ? 2u : 1u;
        *n_sph_out = st->n_sph;   // (read back with the item count: the caller need not count the spheres on the host)
    }
}
__global__ void k_prim_items_hoisted(int n, const float4* blo, const float4* bhi, const uint32_t* flag, const uint32_t* rank, float4* it_lo, float4* it_hi,
                                     uint32_t* vals, uint32_t* sph) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    vals[p] = (uint32_t)p;
    if (flag[p]) {
        const float4 l = blo[p], h = bhi[p];
        it_lo[rank[p]] = make_float4(l.x, l.y, l.z, __int_as_float(n - 1 + p));
        it_hi[rank[p]] = make_float4(h.x, h.y, h.z, __int_as_float(1));
    } else {
        const uint32_t j = (uint32_t)p - rank[p];   // (spheres before this one)
        if (j < MPT_LBVH_HOIST_MAX) sph[j] = (uint32_t)p;
    }
}
__global__ void k_sah_to_radix_hoisted(int n, int ns, const int2* s_child, const float4* s_lo, const float4* s_hi, const float4* blo, const float4* bhi,
                                       const uint32_t* sph, const uint32_t* flag, const uint32_t* rank, const uint32_t* first_inner, const uint32_t* first_item,
                                       int2* child, int* parent, uint32_t* size, float4* nlo, float4* nhi, uint32_t* first) {
    const int node = blockIdx.x * blockDim.x + threadIdx.x, top = 2 * n - 1;
    if (node >= top) return;
    if (node == 0) parent[0] = -1;
    // depth-first order: the hoisted items (the chain under the root, in their order), then the SAH's items in its final item order
    if (node >= n - 1) {
        const int p = node - (n - 1);
        first[node] = flag[p] ? (uint32_t)ns + first_item[node] : (uint32_t)p - rank[p];
    } else {
        first[node] = node >= ns ? (uint32_t)ns + first_inner[node - ns] : node == 0 ? 0u : (uint32_t)(node - 1);
    }
    if (node >= n - 1) {   // a primitive
        size[node] = 1u;
        nlo[node] = blo[node - (n - 1)];
        nhi[node] = bhi[node - (n - 1)];
        return;
    }
    int x, y;
    float4 lo, hi;
    uint32_t sz;
    if (node >= ns) {      // SAH node k = node - ns
        const int2 c = s_child[node - ns];
        x = c.x >= top ? c.x - top + ns : c.x;
        y = c.y >= top ? c.y - top + ns : c.y;
        lo = s_lo[node - ns];
        hi = s_hi[node - ns];
        sz = (uint32_t)__float_as_int(lo.w);
    } else {               // the root (0) or a node of the chain (j: spheres j-1 .. ns-1)
        const int j0 = node == 0 ? 0 : node - 1;
        lo = make_float4(INFINITY, INFINITY, INFINITY, 0.0f);
        hi = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.0f);
        for (int j = j0; j < ns; ++j) {
            const float4 a = blo[sph[j]], b = bhi[sph[j]];
            lo = make_float4(fminf(lo.x, a.x), fminf(lo.y, a.y), fminf(lo.z, a.z), 0.0f);
            hi = make_float4(fmaxf(hi.x, b.x), fmaxf(hi.y, b.y), fmaxf(hi.z, b.z), 0.0f);
        }
        if (node == 0) {
            const float4 a = s_lo[0], b = s_hi[0];
            lo = make_float4(fminf(lo.x, a.x), fminf(lo.y, a.y), fminf(lo.z, a.z), 0.0f);
            hi = make_float4(fmaxf(hi.x, b.x), fmaxf(hi.y, b.y), fmaxf(hi.z, b.z), 0.0f);
            x = ns == 1 ? n - 1 + (int)sph[0] : 1;
            y = ns;        // the SAH's root
            sz = (uint32_t)n;
        } else {
            x = n - 1 + (int)sph[node - 1];
            y = node == ns - 1 ? n - 1 + (int)sph[ns - 1] : node + 1;
            sz = (uint32_t)(ns - (node - 1));
        }
    }
    child[node] = make_int2(x, y);
    parent[x] = node;
    parent[y] = node;
    size[node] = sz;
    nlo[node] = make_float4(lo.x, lo.y, lo.z, 0.0f);
    nhi[node] = make_float4(hi.x, hi.y, hi.z, 0.0f);
}

// The radix tree as it stands on the device after build_radix (all arrays owned by the Scratch passed in):
//   node ids: internal k in [0, n-1), single primitive at sorted position p -> (n-1) + p;  2n - 1 ids in all
//   an OUTPUT node is one with keep[id] != 0: internal nodes spanning more than leaf_max primitives, and the nodes below
//   them spanning <= leaf_max (the leaves: primitives [first, first + count) of the SORTED order, vals[] = original ids)
struct Radix {
    uint32_t n = 0;
    int leaf_max = 2;
    float4 *prims = nullptr;             // the caller's primitive array (3 float4 each), on the device
    float4 *blo = nullptr, *bhi = nullptr;   // primitive boxes (original order)
    float4 *nlo = nullptr, *nhi = nullptr;   // node boxes by id
    int2 *child = nullptr, *range = nullptr; // internal nodes
    int *parent = nullptr;                   // by id (-1 at the root)
    uint32_t *vals = nullptr;                // sorted position -> original primitive
    uint32_t *keep = nullptr, *index = nullptr;  // output flag / compact output index by id; index[2n-1] = number of output nodes
    int *cb = nullptr;                       // centroid bounds (ordered ints)
    uint32_t n_out = 0;                      // output nodes (read back)
    bool spheres_hoisted = false;            // builder "sah": the spheres hang under the root, the SAH nodes hold triangles only
    uint32_t n_spheres = 0xFFFFFFFFu;        // builder "sah": primitives that are not triangles, counted on the device (other builders: not known here)
    uint32_t n_hoisted = 0;                  // ... how many items hang there (spheres and huge triangles: a chain of n_hoisted - 1 nodes under the root)
};

// the output nodes of a finished tree: flags, compact index, count (one stream synchronisation for the count)
static hipError_t finish_radix(hipStream_t stream, Scratch& sc, Radix& R, uint32_t* pin) {
    const uint32_t n = R.n;
    const int leaf_max = R.leaf_max;
    const size_t nn = 2 * (size_t)n - 1;
    const uint32_t B = 256, gnn = (uint32_t)((nn + B - 1) / B);
    hipLaunchKernelGGL(k_mark, dim3(gnn), dim3(B), 0, stream, (int)n, leaf_max, (const int*)R.parent, (const int2*)R.range, R.keep);
    size_t scan_bytes = 0;
    MPT_LB(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, R.keep, R.index, (int)nn + 1, stream));
    char* tmp2;
    MPT_LB(sc.alloc(&tmp2, scan_bytes));
    MPT_LB(hipMemsetAsync(R.keep + nn, 0, 4, stream));
    MPT_LB(hipcub::DeviceScan::ExclusiveSum(tmp2, scan_bytes, R.keep, R.index, (int)nn + 1, stream));
    MPT_LB(hipGetLastError());
    MPT_LB(hipMemcpyAsync(pin, R.index + nn, 4, hipMemcpyDeviceToHost, stream));
    MPT_LB(hipStreamSynchronize(stream));
    R.n_out = pin[0];
    return hipSuccess;
}

// d_prims: device, 3 float4 per primitive.  Leaves everything of Radix on the device; synchronises the stream once to
// read the output node count (and once per round / level of the clustering or the SAH).
enum { BUILDER_KARRAS = 0, BUILDER_PLOC = 1, BUILDER_SAH = 2 };
static hipError_t build_radix(hipStream_t stream, Scratch& sc, float4* d_prims, uint32_t n, int leaf_max, int builder, Radix& R) {
    int *arrived;
    unsigned long long *keys, *keys2;
    uint32_t *vals0, *vals_sorted;
    PinnedWords pinned;   // the few words read back per round go through pinned memory (a pageable target costs ~0.3 ms per copy)
    MPT_LB(pinned.get(sc.pool, 1));   // (words 64.. : the level slots of mpt_sah::run_sah)
    const size_t nn = 2 * (size_t)n - 1;
    R.n = n;
    R.leaf_max = leaf_max;
    R.prims = d_prims;
    MPT_LB(sc.alloc(&R.blo, n));
    MPT_LB(sc.alloc(&R.bhi, n));
    MPT_LB(sc.alloc(&R.nlo, nn));
    MPT_LB(sc.alloc(&R.nhi, nn));
    MPT_LB(sc.alloc(&R.cb, 6));
    MPT_LB(sc.alloc(&R.parent, nn));
    MPT_LB(sc.alloc(&arrived, n));
    MPT_LB(sc.alloc(&R.child, n));
    MPT_LB(sc.alloc(&R.range, n));
    MPT_LB(sc.alloc(&keys, n));
    MPT_LB(sc.alloc(&keys2, n));
    MPT_LB(sc.alloc(&vals0, n));
    MPT_LB(sc.alloc(&R.vals, n));
    MPT_LB(sc.alloc(&vals_sorted, n));
    MPT_LB(sc.alloc(&R.keep, nn + 1));
    MPT_LB(sc.alloc(&R.index, nn + 1));
    const int init[6] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF, (int)0x80000000, (int)0x80000000, (int)0x80000000};
    const bool top_down = n > 2 && builder == BUILDER_SAH;   // (needs neither the bounds of the centroids nor the refit's arrival counters)
    if (!top_down) {
        MPT_LB(hipMemcpyAsync(R.cb, init, sizeof init, hipMemcpyHostToDevice, stream));
        MPT_LB(hipMemsetAsync(arrived, 0, (size_t)n * 4, stream));
    }
    const uint32_t B = 256, gn = (n + B - 1) / B, gnn = (uint32_t)((nn + B - 1) / B);
    hipLaunchKernelGGL(k_boxes, dim3(gn), dim3(B), 0, stream, (const float4*)d_prims, n, R.blo, R.bhi, top_down ? (int*)nullptr : R.cb);
    if (top_down) {
        // top-down binned SAH over the primitives (mpt_sah.h), then renumbered like the clustering's tree
        float4 *it_lo, *it_hi, *nlo0, *nhi0;
        int2* child0;
        int* parent0;
        uint32_t *size, *first;
        MPT_LB(sc.alloc(&it_lo, n));
        MPT_LB(sc.alloc(&it_hi, n));
        MPT_LB(sc.alloc(&child0, n));
        MPT_LB(sc.alloc(&parent0, nn));
        MPT_LB(sc.alloc(&nlo0, nn));
        MPT_LB(sc.alloc(&nhi0, nn));
        MPT_LB(sc.alloc(&size, nn));
        MPT_LB(sc.alloc(&first, nn));
        uint32_t *flag, *rank, *sph;
        HoistState* hoist;
        MPT_LB(sc.alloc(&flag, (size_t)n + 1));
        MPT_LB(sc.alloc(&rank, (size_t)n + 2));   // (+ 1 word behind the scan's output: the sphere count, k_tri_flags)
        MPT_LB(sc.alloc(&sph, MPT_LBVH_HOIST_MAX));
        MPT_LB(sc.alloc(&hoist, 1));
        MPT_LB(hipMemsetAsync(hoist, 0, sizeof(HoistState), stream));
        hipLaunchKernelGGL(k_hoist_hist, dim3(std::min(gn, 512u)), dim3(B), 0, stream, (int)n, (const float4*)d_prims, (const float4*)R.blo, (const float4*)R.bhi, hoist);
        hipLaunchKernelGGL(k_hoist_median, dim3(1), dim3(64), 0, stream, (int)n, hoist);
        hipLaunchKernelGGL(k_tri_flags, dim3((n + 1 + B - 1) / B), dim3(B), 0, stream, (int)n, (const float4*)d_prims, (const float4*)R.blo, (const float4*)R.bhi, hoist,
                           flag, rank + n + 1);
        {
            size_t sb = 0;
            MPT_LB(hipcub::DeviceScan::ExclusiveSum(nullptr, sb, flag, rank, (int)n + 1, stream));
            char* stmp;
            MPT_LB(sc.alloc(&stmp, sb));
            MPT_LB(hipcub::DeviceScan::ExclusiveSum(stmp, sb, flag, rank, (int)n + 1, stream));
        }
        MPT_LB(hipMemcpyAsync(pinned.p, rank + n, 8, hipMemcpyDeviceToHost, stream));
        MPT_LB(hipStreamSynchronize(stream));
        const uint32_t nt = pinned.p[0], ns = n - nt;
        R.n_spheres = pinned.p[1];
        mpt_sah::SahTree T;
        uint32_t *first_inner, *first_item;
        MPT_LB(sc.alloc(&first_inner, n));
        MPT_LB(sc.alloc(&first_item, nn));
        T.first = mpt_sah::SahFirst{first_inner, first_item};
        if (ns >= 1u && ns <= MPT_LBVH_HOIST_MAX && nt >= 3u && getenv("MPT_SAH_KEEP_SPHERES") == nullptr) {
            R.spheres_hoisted = true;
            R.n_hoisted = ns;
            hipLaunchKernelGGL(k_prim_items_hoisted, dim3(gn), dim3(B), 0, stream, (int)n, (const float4*)R.blo, (const float4*)R.bhi, (const uint32_t*)flag,
                               (const uint32_t*)rank, it_lo, it_hi, vals0, sph);
            MPT_LB(mpt_sah::run_sah(stream, sc, pinned.p, (int)nn, nullptr, nt, nt, it_lo, it_hi, T));
            hipLaunchKernelGGL(k_sah_to_radix_hoisted, dim3(gnn), dim3(B), 0, stream, (int)n, (int)ns, (const int2*)T.child, (const float4*)T.lo, (const float4*)T.hi,
                               (const float4*)R.blo, (const float4*)R.bhi, (const uint32_t*)sph, (const uint32_t*)flag, (const uint32_t*)rank, (const uint32_t*)first_inner,
                               (const uint32_t*)first_item, child0, parent0, size, nlo0, nhi0, first);
        } else {
            hipLaunchKernelGGL(k_prim_items, dim3(gn), dim3(B), 0, stream, (int)n, (const float4*)R.blo, (const float4*)R.bhi, it_lo, it_hi, vals0);
            MPT_LB(mpt_sah::run_sah(stream, sc, pinned.p, (int)nn, nullptr, n, n, it_lo, it_hi, T));
            hipLaunchKernelGGL(k_sah_to_radix, dim3(gnn), dim3(B), 0, stream, (int)n, (const int2*)T.child, (const float4*)T.lo, (const float4*)T.hi, (const float4*)R.blo,
                               (const float4*)R.bhi, (const uint32_t*)first_inner, (const uint32_t*)first_item, child0, parent0, size, nlo0, nhi0, first);
        }
        hipLaunchKernelGGL(k_ploc_renumber<true>, dim3(gnn), dim3(B), 0, stream, (int)n, (const int2*)child0, (const int*)parent0, (const uint32_t*)size,
                           (const uint32_t*)first, (const float4*)nlo0, (const float4*)nhi0, (const uint32_t*)vals0, R.child, R.parent, R.range, R.nlo, R.nhi, R.vals);
        return finish_radix(stream, sc, R, pinned.p);
    }
    hipLaunchKernelGGL(k_morton, dim3(gn), dim3(B), 0, stream, (const float4*)R.blo, (const float4*)R.bhi, n, (const int*)R.cb, keys, vals0);
    size_t tmp_bytes = 0;
    MPT_LB(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys, keys2, vals0, vals_sorted, (int)n, 0, 63, stream));
    char* tmp;
    MPT_LB(sc.alloc(&tmp, tmp_bytes));
    MPT_LB(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, keys, keys2, vals0, vals_sorted, (int)n, 0, 63, stream));
    if (n > 2 && builder == BUILDER_PLOC) {
        // clustering on scratch arrays in creation order, then the renumbered tree into R's arrays
        int2 *child0;
        int *parent0;
        float4 *nlo0, *nhi0;
        uint32_t *C0, *C1, *size, *nn_, *merged, *valid, *mslot, *vpos, *first;
        PlocState* st;
        MPT_LB(sc.alloc(&child0, n));
        MPT_LB(sc.alloc(&parent0, nn));
        MPT_LB(sc.alloc(&nlo0, nn));
        MPT_LB(sc.alloc(&nhi0, nn));
        MPT_LB(sc.alloc(&C0, n));
        MPT_LB(sc.alloc(&C1, n));
        MPT_LB(sc.alloc(&size, nn));
        MPT_LB(sc.alloc(&nn_, n));
        MPT_LB(sc.alloc(&merged, n));
        MPT_LB(sc.alloc(&valid, n));
        MPT_LB(sc.alloc(&mslot, n));
        MPT_LB(sc.alloc(&vpos, n));
        MPT_LB(sc.alloc(&first, nn));
        MPT_LB(sc.alloc(&st, 1));
        size_t sb = 0;
        MPT_LB(hipcub::DeviceScan::ExclusiveSum(nullptr, sb, merged, mslot, (int)n, stream));
        char* stmp;
        MPT_LB(sc.alloc(&stmp, sb));
        hipLaunchKernelGGL(k_ploc_init, dim3(gn), dim3(B), 0, stream, (int)n, (const uint32_t*)vals_sorted, (const float4*)R.blo, (const float4*)R.bhi, C0, size,
                           nlo0, nhi0, parent0, st);
        uint32_t alive = n;          // upper bound of the clusters alive (the exact count lives on the device)
        for (int round = 0; round < 4096 && alive > 1u; ++round) {
            const uint32_t g = (alive + B - 1) / B;
            hipLaunchKernelGGL(k_ploc_nn, dim3(g), dim3(B), 0, stream, (const PlocState*)st, (const uint32_t*)C0, (const float4*)nlo0, (const float4*)nhi0, nn_);
            hipLaunchKernelGGL(k_ploc_flags, dim3(g), dim3(B), 0, stream, (const PlocState*)st, (const uint32_t*)nn_, merged, valid);
            MPT_LB(hipcub::DeviceScan::ExclusiveSum(stmp, sb, merged, mslot, (int)alive, stream));
            MPT_LB(hipcub::DeviceScan::ExclusiveSum(stmp, sb, valid, vpos, (int)alive, stream));
            hipLaunchKernelGGL(k_ploc_merge, dim3(g), dim3(B), 0, stream, st, (const uint32_t*)C0, (const uint32_t*)nn_, (const uint32_t*)merged, (const uint32_t*)valid,
                               (const uint32_t*)mslot, (const uint32_t*)vpos, (int)n, C1, child0, parent0, size, nlo0, nhi0);
            hipLaunchKernelGGL(k_ploc_advance, dim3(1), dim3(64), 0, stream, st, (const uint32_t*)merged, (const uint32_t*)mslot, (const uint32_t*)valid, (const uint32_t*)vpos);
            std::swap(C0, C1);
            // every round merges at least one pair; in practice the count falls by ~40 %: read it back every few rounds
            if ((round & 3) == 3 || alive <= 4096u) {
                PlocState& h = *(PlocState*)pinned.p;
                MPT_LB(hipMemcpyAsync(&h, st, sizeof h, hipMemcpyDeviceToHost, stream));
                MPT_LB(hipStreamSynchronize(stream));
                alive = h.n_clusters;
            }
        }
        if (alive != 1u) return hipErrorUnknown;
        hipLaunchKernelGGL(k_ploc_first, dim3(gnn), dim3(B), 0, stream, (int)n, (const int2*)child0, (const int*)parent0, (const uint32_t*)size, first);
        hipLaunchKernelGGL(k_ploc_renumber<false>, dim3(gnn), dim3(B), 0, stream, (int)n, (const int2*)child0, (const int*)parent0, (const uint32_t*)size, (const uint32_t*)first,
                           (const float4*)nlo0, (const float4*)nhi0, (const uint32_t*)vals_sorted, R.child, R.parent, R.range, R.nlo, R.nhi, R.vals);
    } else {
        MPT_LB(hipMemcpyAsync(R.vals, vals_sorted, (size_t)n * 4, hipMemcpyDeviceToDevice, stream));
        if (n > 1) {
            hipLaunchKernelGGL(k_hierarchy, dim3(gn), dim3(B), 0, stream, (const unsigned long long*)keys2, (int)n, R.child, R.parent, R.range);
        } else {
            const int minus1 = -1;
            MPT_LB(hipMemcpyAsync(R.parent, &minus1, 4, hipMemcpyHostToDevice, stream));
        }
        hipLaunchKernelGGL(k_refit, dim3(gn), dim3(B), 0, stream, (const uint32_t*)R.vals, (const float4*)R.blo, (const float4*)R.bhi, (int)n,
                           (const int2*)R.child, (const int*)R.parent, R.nlo, R.nhi, arrived);
    }
    return finish_radix(stream, sc, R, pinned.p);
}

// the output tree in the reference's buffer format, on the device: d_bvh (2 float4 per output node), d_idx (n ints)
static hipError_t emit_reference_format(hipStream_t stream, const Radix& R, float4* d_bvh, int* d_idx) {
    const size_t nn = 2 * (size_t)R.n - 1;
    hipLaunchKernelGGL(k_emit, dim3((uint32_t)((nn + 255) / 256)), dim3(256), 0, stream, (int)R.n, R.leaf_max, (const int2*)R.child, (const int2*)R.range,
                       (const uint32_t*)R.keep, (const uint32_t*)R.index, (const float4*)R.nlo, (const float4*)R.nhi, (const uint32_t*)R.vals, d_bvh, d_idx);
    return hipGetLastError();
}

// prims: host, 12 floats per primitive (Scene::createTransformsBuffer).  bvh_out: host, room for 8 * (2n - 1) floats;
// prim_idx_out: host, n ints.  Returns hipSuccess and the node count, or the failing HIP status.
static hipError_t build(hipStream_t stream, const float* prims, uint32_t n, int leaf_max, int builder, float* bvh_out, uint64_t* n_nodes_out,
                        int32_t* prim_idx_out, float* ms_out, ScratchPool* pool = nullptr) {
    Scratch sc(pool);
    MPT_LB(sc.reserve((size_t)n * 560 + ((size_t)8 << 20)));
    float4 *d_prims, *d_bvh;
    int* d_idx;
    MPT_LB(sc.alloc(&d_prims, 3 * (size_t)n));
    MPT_LB(sc.alloc(&d_bvh, 2 * (2 * (size_t)n - 1)));
    MPT_LB(sc.alloc(&d_idx, n));
    MPT_LB(hipMemcpyAsync(d_prims, prims, (size_t)n * 48, hipMemcpyHostToDevice, stream));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    MPT_LB(hipEventCreate(&e0));
    hipError_t rc = hipEventCreate(&e1);
    if (rc != hipSuccess) {
        hipEventDestroy(e0);
        return rc;
    }
    auto body = [&]() -> hipError_t {
        MPT_LB(hipEventRecord(e0, stream));
        Radix R;
        MPT_LB(build_radix(stream, sc, d_prims, n, leaf_max, builder, R));
        MPT_LB(emit_reference_format(stream, R, d_bvh, d_idx));
        MPT_LB(hipEventRecord(e1, stream));
        MPT_LB(hipStreamSynchronize(stream));
        MPT_LB(hipMemcpy(bvh_out, d_bvh, (size_t)R.n_out * 32, hipMemcpyDeviceToHost));
        MPT_LB(hipMemcpy(prim_idx_out, d_idx, (size_t)n * 4, hipMemcpyDeviceToHost));
        *n_nodes_out = R.n_out;
        if (ms_out) MPT_LB(hipEventElapsedTime(ms_out, e0, e1));
        return hipSuccess;
    };
    rc = body();
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return rc;
}

}  // namespace mpt_lbvh
