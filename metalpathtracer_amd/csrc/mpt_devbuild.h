// mpt_devbuild.h — build -> render without the host: everything mpt_upload_scene derives on the host from
// the reference's flat arrays (R/Scene/Scene.h:71-93,195-317 build, :99-167 packers) is derived here ON THE DEVICE from the
// radix tree of mpt_lbvh.h, and stays there:
//   * the threaded reference-order tree (mpt_device.h: hit link = the child the reference pops first, the RIGHT one,
//     PathTracing.h:188-193; miss link = the node after the subtree), breadth-first so that its top can be staged in LDS
//   * the primitive records in leaf order (48 B: v0, e1, e2 + reference leaf, material, original id), the de-duplicated
//     material table, the reference leaf boxes
//   * the product's own 4-wide tree (mpt_accel.h format) as a collapse of the SAME binary tree — the child with the largest
//     box is opened until four — with the spheres on the always list and kept out of the boxes
// The same tree in the reference's buffer format is kept on the device as well (mpt_download_bvh: what the oracle, or the
// reference's own shader, would walk), so parity is checked as for every other builder: oracle(downloaded arrays) == HIP.
#pragma once
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include "mpt_accel.h"
#include "mpt_device.h"
#include "mpt_lbvh.h"

namespace mpt_devbuild {
using mpt_lbvh::Radix;
using mpt_lbvh::Scratch;

struct Scalars {             // device-side results the host reads back once, at the end
    uint32_t n_spheres;      // spheres found (the first 32 positions are recorded)
    uint32_t sphere_pos[32]; // their positions in the device primitive array
    uint32_t tri_extent;     // bits of the largest finite |coordinate| of a triangle vertex
    uint32_t n_mats;
    uint32_t n_leaves;
    uint32_t n_acc_nodes, acc_depth;
};

__device__ __forceinline__ int span_of(const int2* range, int n, int node) { return node >= n - 1 ? 1 : range[node].y - range[node].x + 1; }
__device__ __forceinline__ int first_of(const int2* range, int n, int node) { return node >= n - 1 ? node - (n - 1) : range[node].x; }

// ---- materials: sort by a 64-bit hash, mark the runs, number them ---------------------------------------------------------
__global__ void k_mat_hash(const float4* mats, uint32_t n, unsigned long long* keys, uint32_t* ids) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 a = ((const uint4*)mats)[2 * (size_t)i], b = ((const uint4*)mats)[2 * (size_t)i + 1];
    unsigned long long h = 0xcbf29ce484222325ull;
    const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    for (int k = 0; k < 8; ++k) {
        h ^= w[k];
        h *= 0x100000001b3ull;
        h ^= h >> 29;
    }
    keys[i] = h;
    ids[i] = i;
}
__global__ void k_mat_heads(const float4* mats, const uint32_t* ids, uint32_t n, uint32_t* head) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    bool h = j == 0;
    if (!h) {
        const uint4* m = (const uint4*)mats;
        const uint4 a0 = m[2 * (size_t)ids[j]], a1 = m[2 * (size_t)ids[j] + 1], b0 = m[2 * (size_t)ids[j - 1]], b1 = m[2 * (size_t)ids[j - 1] + 1];
        h = a0.x != b0.x || a0.y != b0.y || a0.z != b0.z || a0.w != b0.w || a1.x != b1.x || a1.y != b1.y || a1.z != b1.z || a1.w != b1.w;
    }
    head[j] = h ? 1u : 0u;
}
__global__ void k_mat_scatter(const float4* mats, const uint32_t* ids, const uint32_t* head, const uint32_t* rank /* inclusive scan of head */,
                              uint32_t n, uint32_t* mat_of_prim, float4* table, Scalars* sc) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t id = rank[j] - 1u;
    mat_of_prim[ids[j]] = id;
    if (head[j]) {
        table[2 * (size_t)id] = mats[2 * (size_t)ids[j]];
        table[2 * (size_t)id + 1] = mats[2 * (size_t)ids[j] + 1];
    }
    if (j == n - 1) sc->n_mats = rank[j];
}

// ---- triangle extent (for the box padding) --------------------------------------------------------------------------------
__global__ void k_tri_extent(const float4* prims, uint32_t n, Scalars* sc) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    float m = 0.0f;
    if (i < n) {
        const float4 p0 = prims[3 * (size_t)i], p1 = prims[3 * (size_t)i + 1], p2 = prims[3 * (size_t)i + 2];
        if ((int)p0.w == 1) {
            const float v[9] = {p0.x, p0.y, p0.z, p1.x, p1.y, p1.z, p2.x, p2.y, p2.z};
            for (int k = 0; k < 9; ++k)
                if (isfinite(v[k])) m = fmaxf(m, fabsf(v[k]));
        }
    }
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63u) == 0 && m > 0.0f) atomicMax(&sc->tri_extent, __float_as_uint(m));
}

// ---- leaves: number them, lay their primitives out, their reference boxes, their own (sphere-free) boxes ----------------------
__global__ void k_leaf_flags(int n, int leaf_max, const int2* range, const uint32_t* keep, uint32_t* is_leaf) {
    const int node = blockIdx.x * blockDim.x + threadIdx.x;
    if (node >= 2 * n - 1) return;
    is_leaf[node] = keep[node] && span_of(range, n, node) <= leaf_max ? 1u : 0u;
}
// one thread per output leaf.  own box of the leaf BEFORE the final padding: the reference leaf box; for a leaf that holds a
// sphere, a box around its triangles only (+5 % of their extent + pad, clipped to the leaf box) — empty if it has none
__global__ void k_leaves(int n, int leaf_max, const int2* range, const uint32_t* is_leaf, const uint32_t* leaf_id, const uint32_t* vals,
                         const float4* prims, const uint32_t* mat_of_prim, const float4* nlo, const float4* nhi, float4* dprims,
                         float4* refleaf, float4* olo, float4* ohi, Scalars* sc, int use_always_hint) {
    const int node = blockIdx.x * blockDim.x + threadIdx.x;
    if (node >= 2 * n - 1 || !is_leaf[node]) return;
    const uint32_t leaf = leaf_id[node];
    const int first = first_of(range, n, node), count = span_of(range, n, node);
    const float4 lo = nlo[node], hi = nhi[node];
    refleaf[2 * (size_t)leaf] = make_float4(lo.x, lo.y, lo.z, 0.0f);
    refleaf[2 * (size_t)leaf + 1] = make_float4(hi.x, hi.y, hi.z, 0.0f);
    float tl[3] = {INFINITY, INFINITY, INFINITY}, th[3] = {-INFINITY, -INFINITY, -INFINITY};
    int ntri = 0, nsph = 0;
    for (int k = 0; k < count; ++k) {
        const uint32_t pid = vals[first + k];
        const float4 p0 = prims[3 * (size_t)pid], p1 = prims[3 * (size_t)pid + 1], p2 = prims[3 * (size_t)pid + 2];
        const int type = (int)p0.w;
        float4 r0, r1, r2;
        const uint32_t tag = (leaf << 1) | (type == 1 ? 1u : 0u);
        if (type == 1) {  // triangle: v0, e1 = v1 - v0, e2 = v2 - v0 (PathTracing.h:149-150), the subtraction the shader performs
            r0 = make_float4(p0.x, p0.y, p0.z, __uint_as_float(tag));
            r1 = make_float4(p1.x - p0.x, p1.y - p0.y, p1.z - p0.z, __uint_as_float(mat_of_prim[pid]));
            r2 = make_float4(p2.x - p0.x, p2.y - p0.y, p2.z - p0.z, __uint_as_float(pid));
            const float v[3][3] = {{p0.x, p0.y, p0.z}, {p0.x + r1.x, p0.y + r1.y, p0.z + r1.z}, {p0.x + r2.x, p0.y + r2.y, p0.z + r2.z}};
            for (int q = 0; q < 3; ++q)
                for (int a = 0; a < 3; ++a) {
                    tl[a] = fminf(tl[a], v[q][a]);
                    th[a] = fmaxf(th[a], v[q][a]);
                }
            ntri++;
        } else {  // sphere; anything that is neither is never hit (PathTracing.h:120,143): kept as a sphere of radius NaN
            r0 = make_float4(p0.x, p0.y, p0.z, __uint_as_float(tag));
            r1 = make_float4(type == 0 ? p1.x : NAN, 0.0f, 0.0f, __uint_as_float(mat_of_prim[pid]));
            r2 = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(pid));
            nsph++;
            const uint32_t s = atomicAdd(&sc->n_spheres, 1u);
            if (s < 32u) sc->sphere_pos[s] = (uint32_t)(first + k);
        }
        dprims[3 * (size_t)(first + k)] = r0;
        dprims[3 * (size_t)(first + k) + 1] = r1;
        dprims[3 * (size_t)(first + k) + 2] = r2;
    }
    float4 bl = make_float4(lo.x, lo.y, lo.z, 0.0f), bh = make_float4(hi.x, hi.y, hi.z, 0.0f);
    if (nsph != 0 && use_always_hint) {
        if (ntri == 0) {
            bl = make_float4(INFINITY, INFINITY, INFINITY, 0.0f);
            bh = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.0f);
        } else {
            const float pad = fmaxf(__uint_as_float(sc->tri_extent), 1e-6f) * 6.103515625e-05f;
            const float ext = fmaxf(th[0] - tl[0], fmaxf(th[1] - tl[1], th[2] - tl[2]));
            const float g = 0.05f * ext + pad;
            bl = make_float4(fmaxf(lo.x, tl[0] - g), fmaxf(lo.y, tl[1] - g), fmaxf(lo.z, tl[2] - g), 0.0f);
            bh = make_float4(fminf(hi.x, th[0] + g), fminf(hi.y, th[1] + g), fminf(hi.z, th[2] + g), 0.0f);
        }
    }
    olo[node] = bl;
    ohi[node] = bh;
}
// own boxes of the inner nodes: unions of the children's own boxes, bottom-up (the second thread to arrive computes)
__global__ void k_own_refit(int n, const uint32_t* is_leaf, const int2* child, const int* parent, float4* olo, float4* ohi, int* arrived) {
    const int leaf = blockIdx.x * blockDim.x + threadIdx.x;
    if (leaf >= 2 * n - 1 || !is_leaf[leaf]) return;
    __threadfence();
    int node = parent[leaf];
    while (node >= 0) {
        if (atomicAdd(&arrived[node], 1) == 0) return;
        __threadfence();
        const int2 c = child[node];
        const float4 a0 = mpt_lbvh::ld4(olo + c.x), a1 = mpt_lbvh::ld4(ohi + c.x), b0 = mpt_lbvh::ld4(olo + c.y), b1 = mpt_lbvh::ld4(ohi + c.y);
        olo[node] = make_float4(fminf(a0.x, b0.x), fminf(a0.y, b0.y), fminf(a0.z, b0.z), 0.0f);
        ohi[node] = make_float4(fmaxf(a1.x, b1.x), fmaxf(a1.y, b1.y), fmaxf(a1.z, b1.z), 0.0f);
        __threadfence();
        node = parent[node];
    }
}

// ---- the threaded reference-order tree --------------------------------------------------------------------------------------
// depth of every output node and its skip link: the node the reference visits after this node's subtree.  The reference
// pops the right child first (PathTracing.h:188-193): after a RIGHT child's subtree comes its left sibling; a left child
// inherits its parent's skip link.
__global__ void k_depth_skip(int n, const uint32_t* keep, const uint32_t* index, const int2* child, const int* parent, uint32_t* depth_c /* by compact index */,
                             uint32_t* id_c, int* skip /* by id: node id or -1 = the end */) {
    const int node = blockIdx.x * blockDim.x + threadIdx.x;
    if (node >= 2 * n - 1 || !keep[node]) return;
    uint32_t d = 0;
    int sk = -2;
    for (int y = node, p = parent[node]; p >= 0; y = p, p = parent[p]) {
        if (sk == -2 && y == child[p].y) sk = child[p].x;
        ++d;
    }
    skip[node] = sk == -2 ? -1 : sk;
    depth_c[index[node]] = d < 255u ? d : 255u;
    id_c[index[node]] = (uint32_t)node;
}
__global__ void k_positions(uint32_t n_out, const uint32_t* order /* breadth-first position -> id */, uint32_t* tpos /* by id */) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_out) tpos[order[i]] = i;
}
__global__ void k_emit_threaded(uint32_t n_out, int n, const uint32_t* order, const uint32_t* tpos, const uint32_t* is_leaf, const int2* child,
                                const int2* range, const int* skip, const float4* nlo, const float4* nhi, float4* nodes) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    const int id = (int)order[i];
    const float4 lo = nlo[id], hi = nhi[id];
    const uint32_t next = skip[id] < 0 ? n_out : tpos[skip[id]];
    uint32_t A, B = next;
    if (is_leaf[id]) A = MPT_NODE_HOLD | ((uint32_t)first_of(range, n, id) * 16u + (uint32_t)(span_of(range, n, id) - 1));
    else A = tpos[child[id].y];   // box hit: the child the reference pops first
    nodes[2 * (size_t)i] = make_float4(lo.x, lo.y, lo.z, __uint_as_float(A));
    nodes[2 * (size_t)i + 1] = make_float4(hi.x, hi.y, hi.z, __uint_as_float(B));
}

// ---- the own 4-wide tree: breadth-first collapse of the binary tree, one workgroup, level by level ----------------------------
__device__ __forceinline__ float half_area4(float4 lo, float4 hi) {
    const float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
    return dx * dy + dy * dz + dz * dx;
}
__device__ __forceinline__ bool empty4(float4 lo, float4 hi) { return !(hi.x >= lo.x && hi.y >= lo.y && hi.z >= lo.z); }
#define MPT_DB_THREADS 1024
__global__ __launch_bounds__(MPT_DB_THREADS) void k_collapse4(int n, int root, const uint32_t* is_leaf, const int2* child, const int2* range,
                                                              const float4* olo, const float4* ohi, uint32_t* wbin /* wide node -> binary node */,
                                                              float4* acc_nodes, uint32_t cap, Scalars* sc) {
    typedef hipcub::BlockScan<uint32_t, MPT_DB_THREADS> Scan;
    __shared__ typename Scan::TempStorage tmp;
    __shared__ uint32_t s_next;
    const float pad = fmaxf(__uint_as_float(sc->tri_extent), 1e-6f) * 6.103515625e-05f;  // 2^-14: covers the rcp / fma slab arithmetic
    if (threadIdx.x == 0) {
        wbin[0] = (uint32_t)root;
        s_next = 1u;
    }
    __syncthreads();
    uint32_t begin = 0, end = 1, depth = 0;
    while (begin < end) {
        ++depth;
        for (uint32_t base = begin; base < end; base += MPT_DB_THREADS) {
            const uint32_t k = base + threadIdx.x;
            const bool valid = k < end;
            int ch[4] = {-1, -1, -1, -1};
            int nc = 0;
            if (valid) {
                const int b = (int)wbin[k];
                if (is_leaf[b]) {  // (only the root can be a leaf here: a tree of one leaf)
                    if (!empty4(olo[b], ohi[b])) ch[nc++] = b;
                } else {
                    const int2 c = child[b];
                    if (!empty4(olo[c.x], ohi[c.x])) ch[nc++] = c.x;
                    if (!empty4(olo[c.y], ohi[c.y])) ch[nc++] = c.y;
                    for (int round = 0; round < 6 && nc < 4; ++round) {   // open the inner child with the largest box
                        int bi = -1;
                        float ba = -1.0f;
                        for (int i = 0; i < nc; ++i)
                            if (!is_leaf[ch[i]]) {
                                const float a = half_area4(olo[ch[i]], ohi[ch[i]]);
                                if (a > ba) ba = a, bi = i;
                            }
                        if (bi < 0) break;
                        const int2 g = child[ch[bi]];
                        const bool e0 = empty4(olo[g.x], ohi[g.x]), e1 = empty4(olo[g.y], ohi[g.y]);
                        if (!e0 && !e1) {
                            ch[bi] = g.x;
                            ch[nc++] = g.y;
                        } else {
                            ch[bi] = e0 ? g.y : g.x;   // (one side holds spheres only: the node is just its other child)
                        }
                    }
                }
            }
            uint32_t nint = 0;
            for (int i = 0; i < nc; ++i) nint += is_leaf[ch[i]] ? 0u : 1u;
            uint32_t off = 0, total = 0;
            Scan(tmp).ExclusiveSum(nint, off, total);
            __syncthreads();
            const uint32_t at = s_next + off;
            if (valid && k < cap) {
                float o[4 * MPT_OT_NODE_STRIDE];
                for (uint32_t q = 0; q < 4u * MPT_OT_NODE_STRIDE; ++q) o[q] = 0.0f;
                uint32_t j = 0;
                for (int c = 0; c < 4; ++c) {
                    uint32_t ref = MPT_ACCEL_EMPTY;
                    float lo[3] = {INFINITY, 0.0f, 0.0f}, hi[3] = {INFINITY, 0.0f, 0.0f};   // empty slot: a box no walked ray enters (mpt_accel.h emit)
                    if (c < nc) {
                        const float4 l = olo[ch[c]], h = ohi[ch[c]];
                        lo[0] = l.x - pad; lo[1] = l.y - pad; lo[2] = l.z - pad;
                        hi[0] = h.x + pad; hi[1] = h.y + pad; hi[2] = h.z + pad;
                        if (is_leaf[ch[c]]) {
                            ref = MPT_ACCEL_LEAF | ((uint32_t)(span_of(range, n, ch[c]) - 1) << 27) | (uint32_t)first_of(range, n, ch[c]);
                        } else {
                            ref = at + j;
                            if (at + j < cap) wbin[at + j] = (uint32_t)ch[c];
                            ++j;
                        }
                    }
                    for (int a = 0; a < 3; ++a) {
                        o[4 * a + c] = lo[a];
                        o[12 + 4 * a + c] = hi[a];
                    }
                    o[24 + c] = __uint_as_float(ref);
                }
                float4* dst = acc_nodes + (size_t)k * MPT_OT_NODE_STRIDE;
                for (uint32_t q = 0; q < MPT_OT_NODE_STRIDE; ++q) dst[q] = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
            }
            __syncthreads();
            if (threadIdx.x == 0) s_next += total;
            __syncthreads();
        }
        begin = end;
        end = s_next < cap ? s_next : cap;
        __threadfence_block();   // wbin written above is read by this workgroup's next level
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        sc->n_acc_nodes = end;
        sc->acc_depth = depth;
    }
}

// ---- the always list (<= 16 spheres): 5 float4 each, in the order of the device primitive array ---------------------------------
__global__ void k_always(Scalars* sc, float4* dprims, const float4* refleaf, float4* always) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const uint32_t ns = sc->n_spheres < 32u ? sc->n_spheres : 32u;
    for (uint32_t i = 1; i < ns; ++i) {   // (found in atomic order: sort by position)
        const uint32_t v = sc->sphere_pos[i];
        uint32_t j = i;
        for (; j > 0 && sc->sphere_pos[j - 1] > v; --j) sc->sphere_pos[j] = sc->sphere_pos[j - 1];
        sc->sphere_pos[j] = v;
    }
    if (sc->n_spheres > MPT_ACCEL_MAX_ALWAYS) return;   // too many: the closest-first pipeline is not used for this scene
    for (uint32_t k = 0; k < ns; ++k) {
        const uint32_t i = sc->sphere_pos[k];
        float4 r0 = dprims[3 * (size_t)i], r1 = dprims[3 * (size_t)i + 1];
        const float4 r2 = dprims[3 * (size_t)i + 2];
        const uint32_t leaf = __float_as_uint(r0.w) >> 1;
        r1.z = __uint_as_float(k);             // the primitive record points at its always-list entry ...
        dprims[3 * (size_t)i + 1] = r1;
        r1.y = __uint_as_float(i);             // ... and the entry at the primitive (what the walk reports as the winner)
        always[5 * (size_t)k] = r0;
        always[5 * (size_t)k + 1] = r1;
        always[5 * (size_t)k + 2] = r2;
        always[5 * (size_t)k + 3] = refleaf[2 * (size_t)leaf];
        always[5 * (size_t)k + 4] = refleaf[2 * (size_t)leaf + 1];
    }
}

// Everything above, in order, on one stream.  Inputs: the caller's primitive and material arrays, ALREADY on the device
// (d_prims_in: 3 float4 per primitive, d_mats_in: 2 float4 per primitive).  All outputs are hipMalloc'ed here and handed to
// the caller (who frees them); scratch is freed on return.
struct Built {
    float4 *nodes = nullptr, *prims = nullptr, *mats = nullptr, *acc_nodes = nullptr, *refleaf = nullptr, *always = nullptr;
    float4* ref_bvh = nullptr;   // the same tree in the reference's buffer format (2 float4 per node) ...
    int* ref_idx = nullptr;      // ... and its primitiveIndices
    uint32_t n_nodes = 0, n_prims = 0, n_mats = 0, n_acc_nodes = 0, n_always = 0, n_ref_leaves = 0, acc_depth = 0, n_spheres = 0;
    float tri_extent = 0.0f;
    void release() {
        hipFree(nodes); hipFree(prims); hipFree(mats); hipFree(acc_nodes); hipFree(refleaf); hipFree(always); hipFree(ref_bvh); hipFree(ref_idx);
        *this = Built{};
    }
};

static hipError_t build(hipStream_t stream, float4* d_prims_in, const float4* d_mats_in, uint32_t n, int leaf_max, uint32_t n_spheres_hint, Built& out) {
    Scratch sc;
    Radix R;
    Scalars* d_sc;
    MPT_LB(sc.alloc(&d_sc, 1));
    MPT_LB(hipMemsetAsync(d_sc, 0, sizeof(Scalars), stream));
    const uint32_t B = 256, gn = (n + B - 1) / B;
    hipLaunchKernelGGL(k_tri_extent, dim3(gn), dim3(B), 0, stream, (const float4*)d_prims_in, n, d_sc);
    // materials
    unsigned long long *mk, *mk2;
    uint32_t *mi, *mi2, *mhead, *mrank, *mat_of_prim;
    float4* mtable;
    MPT_LB(sc.alloc(&mk, n));
    MPT_LB(sc.alloc(&mk2, n));
    MPT_LB(sc.alloc(&mi, n));
    MPT_LB(sc.alloc(&mi2, n));
    MPT_LB(sc.alloc(&mhead, n));
    MPT_LB(sc.alloc(&mrank, n));
    MPT_LB(sc.alloc(&mat_of_prim, n));
    MPT_LB(sc.alloc(&mtable, 2 * (size_t)n));
    hipLaunchKernelGGL(k_mat_hash, dim3(gn), dim3(B), 0, stream, d_mats_in, n, mk, mi);
    {
        size_t bytes = 0;
        MPT_LB(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, mk, mk2, mi, mi2, (int)n, 0, 64, stream));
        char* tmp;
        MPT_LB(sc.alloc(&tmp, bytes));
        MPT_LB(hipcub::DeviceRadixSort::SortPairs(tmp, bytes, mk, mk2, mi, mi2, (int)n, 0, 64, stream));
        hipLaunchKernelGGL(k_mat_heads, dim3(gn), dim3(B), 0, stream, d_mats_in, (const uint32_t*)mi2, n, mhead);
        size_t sb = 0;
        MPT_LB(hipcub::DeviceScan::InclusiveSum(nullptr, sb, mhead, mrank, (int)n, stream));
        char* tmp2;
        MPT_LB(sc.alloc(&tmp2, sb));
        MPT_LB(hipcub::DeviceScan::InclusiveSum(tmp2, sb, mhead, mrank, (int)n, stream));
        hipLaunchKernelGGL(k_mat_scatter, dim3(gn), dim3(B), 0, stream, d_mats_in, (const uint32_t*)mi2, (const uint32_t*)mhead, (const uint32_t*)mrank, n,
                           mat_of_prim, mtable, d_sc);
    }
    // the binary tree
    MPT_LB(mpt_lbvh::build_radix(stream, sc, d_prims_in, n, leaf_max, R));
    const uint32_t n_out = R.n_out;
    const size_t nn = 2 * (size_t)n - 1;
    const uint32_t gnn = (uint32_t)((nn + B - 1) / B), go = (n_out + B - 1) / B;
    MPT_LB(hipMalloc(&out.ref_bvh, (size_t)n_out * 32));
    MPT_LB(hipMalloc(&out.ref_idx, (size_t)n * 4));
    MPT_LB(mpt_lbvh::emit_reference_format(stream, R, out.ref_bvh, out.ref_idx));
    // leaves
    uint32_t *is_leaf, *leaf_id;
    float4 *olo, *ohi;
    int* arrived;
    MPT_LB(sc.alloc(&is_leaf, nn + 1));
    MPT_LB(sc.alloc(&leaf_id, nn + 1));
    MPT_LB(sc.alloc(&olo, nn));
    MPT_LB(sc.alloc(&ohi, nn));
    MPT_LB(sc.alloc(&arrived, n));
    MPT_LB(hipMemsetAsync(arrived, 0, (size_t)n * 4, stream));
    hipLaunchKernelGGL(k_leaf_flags, dim3(gnn), dim3(B), 0, stream, (int)n, leaf_max, (const int2*)R.range, (const uint32_t*)R.keep, is_leaf);
    MPT_LB(hipMemsetAsync(is_leaf + nn, 0, 4, stream));
    {
        size_t sb = 0;
        MPT_LB(hipcub::DeviceScan::ExclusiveSum(nullptr, sb, is_leaf, leaf_id, (int)nn + 1, stream));
        char* tmp;
        MPT_LB(sc.alloc(&tmp, sb));
        MPT_LB(hipcub::DeviceScan::ExclusiveSum(tmp, sb, is_leaf, leaf_id, (int)nn + 1, stream));
    }
    MPT_LB(hipMalloc(&out.prims, (size_t)n * 48));
    MPT_LB(hipMalloc(&out.refleaf, (size_t)n_out * 32));   // (leaves <= output nodes)
    const int use_always = n_spheres_hint <= MPT_ACCEL_MAX_ALWAYS ? 1 : 0;
    hipLaunchKernelGGL(k_leaves, dim3(gnn), dim3(B), 0, stream, (int)n, leaf_max, (const int2*)R.range, (const uint32_t*)is_leaf, (const uint32_t*)leaf_id,
                       (const uint32_t*)R.vals, (const float4*)d_prims_in, (const uint32_t*)mat_of_prim, (const float4*)R.nlo, (const float4*)R.nhi, out.prims,
                       out.refleaf, olo, ohi, d_sc, use_always);
    hipLaunchKernelGGL(k_own_refit, dim3(gnn), dim3(B), 0, stream, (int)n, (const uint32_t*)is_leaf, (const int2*)R.child, (const int*)R.parent, olo, ohi, arrived);
    // threaded tree, breadth-first
    uint32_t *depth_c, *depth_s, *id_c, *order, *tpos;
    int* skip;
    MPT_LB(sc.alloc(&depth_c, n_out));
    MPT_LB(sc.alloc(&depth_s, n_out));
    MPT_LB(sc.alloc(&id_c, n_out));
    MPT_LB(sc.alloc(&order, n_out));
    MPT_LB(sc.alloc(&tpos, nn));
    MPT_LB(sc.alloc(&skip, nn));
    hipLaunchKernelGGL(k_depth_skip, dim3(gnn), dim3(B), 0, stream, (int)n, (const uint32_t*)R.keep, (const uint32_t*)R.index, (const int2*)R.child,
                       (const int*)R.parent, depth_c, id_c, skip);
    {
        size_t bytes = 0;
        MPT_LB(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, depth_c, depth_s, id_c, order, (int)n_out, 0, 8, stream));
        char* tmp;
        MPT_LB(sc.alloc(&tmp, bytes));
        MPT_LB(hipcub::DeviceRadixSort::SortPairs(tmp, bytes, depth_c, depth_s, id_c, order, (int)n_out, 0, 8, stream));
    }
    hipLaunchKernelGGL(k_positions, dim3(go), dim3(B), 0, stream, n_out, (const uint32_t*)order, tpos);
    MPT_LB(hipMalloc(&out.nodes, (size_t)n_out * 32));
    hipLaunchKernelGGL(k_emit_threaded, dim3(go), dim3(B), 0, stream, n_out, (int)n, (const uint32_t*)order, (const uint32_t*)tpos, (const uint32_t*)is_leaf,
                       (const int2*)R.child, (const int2*)R.range, (const int*)skip, (const float4*)R.nlo, (const float4*)R.nhi, out.nodes);
    // own tree
    const uint32_t cap = n_out / 2u + 2u;   // inner nodes of a binary tree with n_out nodes
    uint32_t* wbin;
    MPT_LB(sc.alloc(&wbin, cap));
    MPT_LB(hipMalloc(&out.acc_nodes, (size_t)cap * MPT_OT_NODE_STRIDE * 16));
    const int root = n > 1 ? 0 : 0;   // internal node 0 is the root (Karras); a single primitive: node id 0 = (n-1) + 0
    hipLaunchKernelGGL(k_collapse4, dim3(1), dim3(MPT_DB_THREADS), 0, stream, (int)n, root, (const uint32_t*)is_leaf, (const int2*)R.child, (const int2*)R.range,
                       (const float4*)olo, (const float4*)ohi, wbin, out.acc_nodes, cap, d_sc);
    MPT_LB(hipMalloc(&out.always, (size_t)MPT_ACCEL_MAX_ALWAYS * 80));
    hipLaunchKernelGGL(k_always, dim3(1), dim3(64), 0, stream, d_sc, out.prims, (const float4*)out.refleaf, out.always);
    MPT_LB(hipGetLastError());
    Scalars h;
    MPT_LB(hipMemcpyAsync(&h, d_sc, sizeof h, hipMemcpyDeviceToHost, stream));
    uint32_t n_leaves = 0;
    MPT_LB(hipMemcpyAsync(&n_leaves, leaf_id + nn, 4, hipMemcpyDeviceToHost, stream));
    MPT_LB(hipStreamSynchronize(stream));
    MPT_LB(hipMalloc(&out.mats, (size_t)std::max(h.n_mats, 1u) * 32));
    MPT_LB(hipMemcpyAsync(out.mats, mtable, (size_t)h.n_mats * 32, hipMemcpyDeviceToDevice, stream));
    MPT_LB(hipStreamSynchronize(stream));
    out.n_nodes = n_out;
    out.n_prims = n;
    out.n_mats = h.n_mats;
    out.n_acc_nodes = h.n_acc_nodes;
    out.acc_depth = h.acc_depth;
    out.n_spheres = h.n_spheres;
    out.n_always = h.n_spheres <= MPT_ACCEL_MAX_ALWAYS ? h.n_spheres : 0u;
    out.n_ref_leaves = n_leaves;
    memcpy(&out.tri_extent, &h.tri_extent, 4);
    return hipSuccess;
}

}  // namespace mpt_devbuild
