// Register-only MFMA rate probe for gfx950: how many bf16 32x32x16 MFMA FLOP/s the chip sustains
// with nothing else in the way (no LDS, no memory), as a function of run length and waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 scripts/micro/mfma_peak.hip -o gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    f32x4 ra = {1.0f * threadIdx.x, 2.0f, 3.0f, 4.0f}, rb = {0.5f, 0.25f, 1.0f * blockIdx.x, 2.0f};
    bf16x8 a = __builtin_bit_cast(bf16x8, ra), b = __builtin_bit_cast(bf16x8, rb);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 123.456f) out[threadIdx.x] = s;
}

// Same loop with eight different pseudo-random operand pairs cycling through the MFMAs (data toggling
// raises power; a constant-operand loop can overstate what real data sustains).
__device__ __forceinline__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__global__ __launch_bounds__(512) void mfma_loop_rand(float* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 a[8], b[8];
    for (int k = 0; k < 8; ++k) {
        unsigned w[8];
        for (int q = 0; q < 8; ++q) {
            // two bf16 per word, exponents kept near 1.0 so nothing overflows
            unsigned r = mix(threadIdx.x * 977u + blockIdx.x * 131u + k * 17u + q);
            w[q] = (r & 0x807f807fu) | 0x3f003f00u;
        }
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 ua = {w[0], w[1], w[2], w[3]}, ub = {w[4], w[5], w[6], w[7]};
        a[k] = __builtin_bit_cast(bf16x8, ua); b[k] = __builtin_bit_cast(bf16x8, ub);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(r * 4 + i) & 7], b[(r * 3 + i) & 7], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 123.456f) out[threadIdx.x] = s;
}

// 16x16x32 shape on the same random operands and the same output tile per wave (4 x 32x32 = 16 x 16x16).
__global__ __launch_bounds__(512) void mfma_loop_rand16(float* out, int iters) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i)
        for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
    bf16x8 a[8], b[8];
    for (int k = 0; k < 8; ++k) {
        unsigned w[8];
        for (int q = 0; q < 8; ++q) {
            unsigned r = mix(threadIdx.x * 977u + blockIdx.x * 131u + k * 17u + q);
            w[q] = (r & 0x807f807fu) | 0x3f003f00u;
        }
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 ua = {w[0], w[1], w[2], w[3]}, ub = {w[4], w[5], w[6], w[7]};
        a[k] = __builtin_bit_cast(bf16x8, ua); b[k] = __builtin_bit_cast(bf16x8, ub);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(r * 4 + i) & 7], b[(r * 3 + (i >> 2)) & 7], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i)
        for (int e = 0; e < 4; ++e) s += acc[i][e];
    if (s == 123.456f) out[threadIdx.x] = s;
}

static void run_rand16(int iters, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop_rand16, dim3(256), dim3(512), 0, 0, out, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop_rand16, dim3(256), dim3(512), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 256.0 * 8 * iters * 32.0 * 16384.0;
    printf("%-28s threads=512 blocks=256 iters=%d  %8.3f ms  %7.1f TFLOP/s\n", "random operands, 16x16x32", iters, ms, flop / ms / 1e9);
}

static void run_rand(int iters, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop_rand, dim3(256), dim3(512), 0, 0, out, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop_rand, dim3(256), dim3(512), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 256.0 * 8 * iters * 16.0 * 32768.0;
    printf("%-28s threads=512 blocks=256 iters=%d  %8.3f ms  %7.1f TFLOP/s\n", "random operands, 2 waves/SIMD", iters, ms, flop / ms / 1e9);
}

// LDS-fed 16x16x32 loops on random data: per k32 slice a wave re-reads its fragments from LDS like the split GEMM does.
//   TILE 0: 128x64 per wave (8 x 4 accumulators): 16 A + 8 B ds_read_b128 per 96 MFMAs, 8 waves per CU (2 per SIMD)
//   TILE 1: 128x128 per wave (8 x 8 accumulators): 16 A + 16 B reads per 192 MFMAs, 4 waves per CU (1 per SIMD)
template <int TILE>
__global__ __launch_bounds__(TILE ? 256 : 512) void mfma_lds_loop(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[32768];             // 128 KB
    constexpr int NJ = TILE ? 8 : 4;
    for (int i = threadIdx.x; i < 32768; i += blockDim.x) {
        unsigned r = mix(i * 2654435761u + blockIdx.x), q = mix(r);
        unsigned w = ((r & 0x807f807fu) | 0x3f003f00u);
        lds[i] = __builtin_bit_cast(float, w ^ (q & 0x00010001u));
    }
    __syncthreads();
    f32x4 acc[8][NJ];
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* base = lds + wave * 2048 + lane * 4;                      // conflict-free: 16 B per lane, contiguous
    for (int it = 0; it < iters; ++it) {
        const float* p = base + (it & 3) * 256;
        f32x4 bh[NJ], bl[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) { bh[j] = *(const f32x4*)(p + j * 512); bl[j] = *(const f32x4*)(p + j * 512 + 4096); }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x4 ah = *(const f32x4*)(p + 8192 + i * 512), al = *(const f32x4*)(p + 12288 + i * 512);
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bh[j]), __builtin_bit_cast(bf16x8, al), acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bl[j]), __builtin_bit_cast(bf16x8, ah), acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bh[j]), __builtin_bit_cast(bf16x8, ah), acc[i][j], 0, 0, 0);
        }
    }
    float sacc = 0.f;
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < NJ; ++j)
            for (int e = 0; e < 4; ++e) sacc += acc[i][j][e];
    if (sacc == 123.456f) out[threadIdx.x] = sacc;
}

template <int TILE>
static void run_lds(int iters, float* out) {
    const int threads = TILE ? 256 : 512;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_lds_loop<TILE>, dim3(256), dim3(threads), 0, 0, out, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_lds_loop<TILE>, dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 256.0 * (threads / 64) * iters * (TILE ? 192.0 : 96.0) * 16384.0;
    printf("%-28s threads=%d blocks=256 iters=%d  %8.3f ms  %7.1f TFLOP/s\n",
           TILE ? "LDS-fed 128x128/wave, 1/SIMD" : "LDS-fed 128x64/wave, 2/SIMD", threads, iters, ms, flop / ms / 1e9);
}

template <int NACC>
static void run(const char* name, int threads, int blocks, int iters, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(threads), 0, 0, out, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * (threads / 64) * iters * 4.0 * NACC * 32768.0;
    printf("%-28s threads=%d blocks=%d iters=%d  %8.3f ms  %7.1f TFLOP/s\n", name, threads, blocks, iters, ms, flop / ms / 1e9);
}

int main() {
    float* out; hipMalloc(&out, 4096);
    for (int rep = 0; rep < 2; ++rep) {
        run<4>("4 acc, 1 wave/SIMD", 256, 256, 2000, out);
        run<4>("4 acc, 2 waves/SIMD", 512, 256, 2000, out);
        run<4>("4 acc, 2 waves/SIMD, long", 512, 256, 40000, out);
        run<2>("2 acc, 2 waves/SIMD", 512, 256, 4000, out);
        run<1>("1 acc (dependent chain)", 512, 256, 8000, out);
        run<8>("8 acc, 2 waves/SIMD", 512, 256, 1000, out);
        run<4>("4 acc, 2 WG/CU x 4 waves", 256, 512, 2000, out);
        run_rand(2000, out);
        run_rand(40000, out);
        run_rand16(2000, out);
        run_rand16(40000, out);
        run_lds<0>(4000, out);
        run_lds<1>(4000, out);
    }
    return 0;
}
