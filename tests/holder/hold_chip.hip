// hold_chip.hip — a FOREIGN persistent kernel for tests (tests/test_gpu_parity.py: the residency gate's fallback).  Test infrastructure only.
//
// hold_chip_start(device, ms) launches, on a stream of its own, two workgroups per compute unit that each take 80 KB of LDS — so that no
// workgroup of a trace kernel (57-78 KB of LDS) fits beside them anywhere — and spin on the 100 MHz clock until `ms` milliseconds have
// passed since the kernel's first wave started: what another process's long kernel looks like to mpt_render_async.  The exit condition is
// a clock every wave reads (no inter-wave dependency): the grid always drains.  hold_chip_wait() synchronises that stream.
//
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -fPIC -shared tests/holder/hold_chip.hip -o tests/holder/_build/libholdchip.so
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
hipStream_t g_stream = nullptr;
unsigned long long* g_t0 = nullptr;

__global__ __launch_bounds__(256) void k_hold(unsigned long long* t0, unsigned long long ticks) {
    extern __shared__ float lds[];
    if (threadIdx.x == 0) {
        lds[0] = 0.0f;   // (the allocation is what matters)
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        atomicCAS(t0, 0ull, now);   // the first wave to arrive starts the clock for everybody
    }
    __syncthreads();
    const unsigned long long start = __hip_atomic_load(t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__builtin_amdgcn_s_memrealtime() - start < ticks) __builtin_amdgcn_s_sleep(64);
}
}  // namespace

extern "C" int hold_chip_start(int device, int ms) {
    if (ms < 1 || ms > 5000) return -1;
    if (hipSetDevice(device) != hipSuccess) return -2;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return -2;
    if (!g_stream && hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return -3;
    if (!g_t0 && hipMalloc(&g_t0, 8) != hipSuccess) return -3;
    if (hipMemsetAsync(g_t0, 0, 8, g_stream) != hipSuccess) return -3;
    const size_t lds = 80 * 1024;
    if (hipFuncSetAttribute((const void*)k_hold, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -4;
    hipLaunchKernelGGL(k_hold, dim3(2 * prop.multiProcessorCount), dim3(256), lds, g_stream, g_t0, (unsigned long long)ms * 100000ull);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}
extern "C" int hold_chip_wait() { return g_stream && hipStreamSynchronize(g_stream) == hipSuccess ? 0 : -1; }
