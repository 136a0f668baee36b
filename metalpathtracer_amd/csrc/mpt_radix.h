// mpt_radix.h — a stable least-significant-digit radix sort of (32-bit key, 32-bit value) pairs, 8 bits a pass, for the device builders.
//
// Why not hipcub::DeviceRadixSort: on this toolchain (ROCm 7.2.0, gfx950) rocPRIM sorts the sizes the builders have (10^4 .. 10^7 pairs) by
// MERGING — a block sort and log2(n / block) merge passes of two kernels each, whatever the bit range: 25 launches and ~190 us for the 1 M
// material keys of a build, and the same 25 launches (243 us) for the 1.5 M 8-bit depths of the threaded tree, which is ONE counting pass
// (profiles/r05_devbuild_timeline.txt; it is also the path whose bit range [32, 64) is wrong: tests/experiments/hipcub_partial_bits.hip).
// Here a pass is three launches: per-wave digit counts, one scan of them, the scatter.
//
// Layout: a WAVE owns a tile of MPT_RADIX_TILE consecutive items and walks it in item order, 64 at a time; counts[digit * tiles + tile] —
// digit-major, so that ONE exclusive scan gives every (digit, tile) its first output position.  Within a 64-item step the lanes of equal
// digit find each other by eight ballots (one per digit bit) and take consecutive positions in lane order: the sort is stable.
#pragma once
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include "mpt_sah.h"   // mpt_lbvh::Scratch, MPT_LB

namespace mpt_radix {
#define MPT_RADIX_TILE 1024u   // items per wave
#define MPT_RADIX_WAVES 4u     // waves per workgroup

// the wave's LDS traffic before this point is done before anything after it starts (the hardware runs a wave's LDS instructions in order:
// this only keeps the compiler from moving them)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// lanes of the wave that hold the same digit as this one (valid lanes only), as a mask
__device__ __forceinline__ unsigned long long radix_peers(uint32_t digit, bool valid) {
    unsigned long long m = __ballot(valid);
    for (int b = 0; b < 8; ++b) {
        const bool bit = (digit >> b) & 1u;
        const unsigned long long v = __ballot(bit);
        m &= bit ? v : ~v;
    }
    return m;
}
__global__ __launch_bounds__(64 * MPT_RADIX_WAVES) void k_radix_count(const uint32_t* keys, uint32_t n, uint32_t shift, uint32_t tiles, uint32_t* counts) {
    __shared__ uint32_t h[MPT_RADIX_WAVES][256];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6, tile = blockIdx.x * MPT_RADIX_WAVES + wv;
    for (uint32_t d = lane; d < 256u; d += 64u) h[wv][d] = 0u;
    wave_sync();
    if (tile >= tiles) return;
    const uint32_t b = tile * MPT_RADIX_TILE;
    for (uint32_t base = b; base < b + MPT_RADIX_TILE && base < n; base += 64u) {
        const uint32_t i = base + lane;
        const bool valid = i < n;
        const uint32_t d = valid ? (keys[i] >> shift) & 255u : 0u;
        const unsigned long long peers = radix_peers(d, valid);
        if (valid && (peers & ((1ull << lane) - 1ull)) == 0ull) h[wv][d] += (uint32_t)__popcll(peers);   // (the group's lowest lane: one writer per digit)
        wave_sync();
    }
    for (uint32_t d = lane; d < 256u; d += 64u) counts[(size_t)d * tiles + tile] = h[wv][d];
}
__global__ __launch_bounds__(64 * MPT_RADIX_WAVES) void k_radix_scatter(const uint32_t* keys, const uint32_t* vals, uint32_t n, uint32_t shift, uint32_t tiles,
                                                                        const uint32_t* first /* exclusive scan of counts */, uint32_t* keys_out, uint32_t* vals_out) {
    __shared__ uint32_t run[MPT_RADIX_WAVES][256];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6, tile = blockIdx.x * MPT_RADIX_WAVES + wv;
    if (tile >= tiles) return;
    for (uint32_t d = lane; d < 256u; d += 64u) run[wv][d] = first[(size_t)d * tiles + tile];
    wave_sync();
    const uint32_t b = tile * MPT_RADIX_TILE;
    for (uint32_t base = b; base < b + MPT_RADIX_TILE && base < n; base += 64u) {
        const uint32_t i = base + lane;
        const bool valid = i < n;
        const uint32_t k = valid ? keys[i] : 0u, v = valid ? vals[i] : 0u;
        const uint32_t d = (k >> shift) & 255u;
        const unsigned long long peers = radix_peers(d, valid), below = peers & ((1ull << lane) - 1ull);
        uint32_t at = 0;
        if (valid) at = run[wv][d] + (uint32_t)__popcll(below);
        wave_sync();   // (every lane has read its digit's position before the group's lowest lane moves it on)
        if (valid && below == 0ull) run[wv][d] += (uint32_t)__popcll(peers);
        wave_sync();
        if (valid) {
            keys_out[at] = k;
            vals_out[at] = v;
        }
    }
}

struct RadixTemp {
    uint32_t *counts = nullptr, *first = nullptr;
    char* scan_tmp = nullptr;
    size_t scan_bytes = 0;
    uint32_t tiles = 0;
};
static hipError_t radix_reserve(mpt_lbvh::Scratch& sc, uint32_t n, hipStream_t stream, RadixTemp& T) {
    T.tiles = (n + MPT_RADIX_TILE - 1u) / MPT_RADIX_TILE;
    const size_t m = (size_t)256 * T.tiles;
    MPT_LB(sc.alloc(&T.counts, m));
    MPT_LB(sc.alloc(&T.first, m));
    MPT_LB(hipcub::DeviceScan::ExclusiveSum(nullptr, T.scan_bytes, T.counts, T.first, (int)m, stream));
    MPT_LB(sc.alloc(&T.scan_tmp, T.scan_bytes));
    return hipSuccess;
}
// Sorts on key bits [0, 8 * passes).  The pairs ping-pong between (keys, vals) and (keys2, vals2); *in_second tells where they ended.
static hipError_t radix_sort_pairs(hipStream_t stream, const RadixTemp& T, uint32_t* keys, uint32_t* vals, uint32_t* keys2, uint32_t* vals2, uint32_t n, int passes,
                                   bool* in_second) {
    const uint32_t g = (T.tiles + MPT_RADIX_WAVES - 1u) / MPT_RADIX_WAVES;
    bool second = false;
    for (int p = 0; p < passes; ++p) {
        uint32_t *ki = second ? keys2 : keys, *vi = second ? vals2 : vals, *ko = second ? keys : keys2, *vo = second ? vals : vals2;
        hipLaunchKernelGGL(k_radix_count, dim3(g), dim3(64 * MPT_RADIX_WAVES), 0, stream, (const uint32_t*)ki, n, 8u * (uint32_t)p, T.tiles, T.counts);
        size_t sb = T.scan_bytes;
        MPT_LB(hipcub::DeviceScan::ExclusiveSum(T.scan_tmp, sb, T.counts, T.first, (int)(256u * T.tiles), stream));
        hipLaunchKernelGGL(k_radix_scatter, dim3(g), dim3(64 * MPT_RADIX_WAVES), 0, stream, (const uint32_t*)ki, (const uint32_t*)vi, n, 8u * (uint32_t)p, T.tiles,
                           (const uint32_t*)T.first, ko, vo);
        second = !second;
    }
    *in_second = second;
    return hipGetLastError();
}
}  // namespace mpt_radix
