// A kernel that does nothing but hold CUs: `wgs` workgroups of 256 threads, each with 64 KB of LDS (so it cannot share a
// CU with the 160 KB workgroups of the split GEMMs), spinning until `usec` microseconds of the 100 MHz real-time clock
// have passed.  Stands in for a collective's (RCCL) kernels on a one-GPU box: scripts/bench_contention.py launches it
// on a second stream and times the GEMMs beside it.
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC scripts/micro/cu_hog.hip -o scripts/micro/build/libcu_hog.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void hog_kernel(long ticks, int* sink) {
    __shared__ int pad[16384];
    pad[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const uint64_t t0 = __builtin_readcyclecounter();
    uint64_t t = t0;
    const uint64_t start = __builtin_amdgcn_s_memrealtime();
    while ((long)(__builtin_amdgcn_s_memrealtime() - start) < ticks) __builtin_amdgcn_s_sleep(32);
    t = __builtin_readcyclecounter();
    if (pad[(threadIdx.x + 1) & 255] == -1) sink[0] = (int)(t - t0);
}

extern "C" int cu_hog(int wgs, int usec, void* stream) {
    static int* sink = nullptr;
    if (!sink && hipMalloc(&sink, 64) != hipSuccess) return -1;
    if (wgs <= 0) return 0;
    hipLaunchKernelGGL(hog_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, (long)usec * 100, sink);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
