// How long does a CU need to push one 256 x 256 fp32 C tile (256 KB) with the split GEMM's store pattern (each wave
// instruction = 16 rows x 64 B), when 1 / 8 / 64 / 256 CUs do it at the same time?  Answers whether the GEMM's per-tile
// write-out time (~18 us) is a per-CU limit or chip-wide burst contention.
//   hipcc -O3 --offload-arch=gfx950 store_burst.hip -o store_burst && ./store_burst
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// PAT: lanes per row segment = 4 (16 rows x 64 B per instruction, the GEMM's), 8 (8 rows x 128 B), 16 (4 x 256 B), 64 (1 x 1 KB)
template <int LPR>
__global__ __launch_bounds__(512) void burst(float* C, int ldc, int tiles_per_wg, long long* t_out, int reps) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    constexpr int RPI = 64 / LPR;                       // rows per instruction
    const int rr = lane / LPR, cc = lane % LPR;
    f32x4 v = {1.f * tid, 2.f, 3.f, 4.f};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r)
        for (int t = 0; t < tiles_per_wg; ++t) {
            const long tile = (long)blockIdx.x * tiles_per_wg + t;
            float* base = C + tile * 256 * (long)ldc;               // tiles stacked along rows, full-width rows of ldc floats
            // the wave owns a 128 x 64 strip (32 KB) = 32 instructions of 1 KB
            for (int k = 0; k < 32; ++k) {
                int row, col;
                if (LPR <= 16) { const int per_row = 16 / LPR; row = (k / per_row) * RPI + rr; col = (k % per_row) * LPR * 4 + cc * 4; }
                else { row = k * 4 + (lane >> 4); col = (lane & 15) * 4; }            // LPR 64: 4 rows x 256 B = the strip's full width
                float* c = base + (long)(wm * 128 + row) * ldc + wn * 64 + col;
                *reinterpret_cast<f32x4*>(c) = v;
            }
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) t_out[blockIdx.x] = t1 - t0;
}

int main() {
    const int ldc = 1024;
    float* C; long long* T;
    hipMalloc(&C, (size_t)256 * 8 * 256 * ldc * 4);
    hipMalloc(&T, 256 * 8);
    for (int pat : {4, 8, 16})
    for (int ncu : {1, 64, 256}) {
        for (int tiles : {4}) {
            auto launch = [&]() {
                if (pat == 4) hipLaunchKernelGGL(burst<4>, dim3(ncu), dim3(512), 0, 0, C, ldc, tiles, T, 1);
                else if (pat == 8) hipLaunchKernelGGL(burst<8>, dim3(ncu), dim3(512), 0, 0, C, ldc, tiles, T, 1);
                else hipLaunchKernelGGL(burst<16>, dim3(ncu), dim3(512), 0, 0, C, ldc, tiles, T, 1);
            };
            launch();
            hipDeviceSynchronize();
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a);
            launch();
            hipEventRecord(b); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, a, b);
            std::vector<long long> h(ncu);
            hipMemcpy(h.data(), T, ncu * 8, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            printf("lanes/row-segment %2d  CUs %3d  tiles/CU %d : kernel %.1f us, in-kernel ticks median %lld max %lld  (%.1f KB per CU) -> %.1f GB/s per CU, %.2f TB/s chip\n",
                   pat, ncu, tiles, ms * 1e3, h[ncu / 2], h[ncu - 1], 256.0 * tiles, 256e3 * tiles / (ms * 1e-3) / 1e9, ncu * 256e3 * tiles / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
