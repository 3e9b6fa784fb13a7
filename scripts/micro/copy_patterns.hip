// What does a one-wave-per-row pass (the structure of ln_prep / ln_act_bwd: the row in registers, reduced, written back)
// reach as a plain copy, and which knobs move it?  512 MB / 1 GB read + the same written, rows of D floats.
//   pattern   0 = lane i owns bytes 16 i .. 16 i + 15 of each 1 KB;  1 = a lane owns 8 consecutive floats (two accesses
//             32 B apart: the sx8 group), loads and stores;  2 = loads as 1, stores as 0
//   nt        non-temporal loads / stores (bit 0 / bit 1)
//   persist   1 = grid of `grid` workgroups walking rows with the next row prefetched;  0 = one row per wave, R / 4 workgroups
// torch's own elementwise kernel (scripts/micro/torch_copy.py: add(x, 1, out=y)) is the yardstick: 6.2 TB/s.
// Build: hipcc -O3 --offload-arch=gfx950 scripts/micro/copy_patterns.hip -o scripts/micro/build/copy_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NT> __device__ __forceinline__ f32x4 ld(const float* p) {
    return (NT & 1) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)) : *reinterpret_cast<const f32x4*>(p);
}
template <int NT> __device__ __forceinline__ void st(float* p, f32x4 v) {
    if (NT & 2) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p)); else *reinterpret_cast<f32x4*>(p) = v;
}

template <int PAT, int NT, int PERSIST, int NS>
__global__ __launch_bounds__(256) void copy_rows(const float* __restrict__ src, float* __restrict__ dst, int R, int D) {
    const int lane = threadIdx.x & 63;
    const int rstep = gridDim.x * 4;
    f32x4 a[NS], b[NS];
    auto load = [&](int row) {
        const float* p = src + (size_t)min(row, R - 1) * D;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            if (PAT == 0) { a[i] = ld<NT>(p + 512 * i + lane * 4); b[i] = ld<NT>(p + 512 * i + 256 + lane * 4); }
            else          { a[i] = ld<NT>(p + 512 * i + lane * 8); b[i] = ld<NT>(p + 512 * i + lane * 8 + 4); }
        }
    };
    int row = blockIdx.x * 4 + threadIdx.x / 64;
    load(row);
    for (; row < R; row += rstep) {
        f32x4 va[NS], vb[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) { va[i] = a[i] * 1.0001f; vb[i] = b[i] * 0.9999f; }
        if (PERSIST) load(row + rstep);
        float* q = dst + (size_t)row * D;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            if (PAT == 1) { st<NT>(q + 512 * i + lane * 8, va[i]); st<NT>(q + 512 * i + lane * 8 + 4, vb[i]); }
            else          { st<NT>(q + 512 * i + lane * 4, va[i]); st<NT>(q + 512 * i + 256 + lane * 4, vb[i]); }
        }
        if (!PERSIST) break;
    }
}

// persistent, lane-contiguous, non-temporal, with the next row's loads as inline asm (the compiler does not track them) and a
// COUNTED wait at the top of the loop: the compiler's own placement is `s_waitcnt vmcnt(0)` right after issuing the
// prefetch (it cannot count across the loop's back edge), which waits for the row just requested and for the previous
// row's stores.
template <int NS>
__global__ __launch_bounds__(256) void copy_rows_counted(const float* __restrict__ src, float* __restrict__ dst, int R, int D) {
    const int lane = threadIdx.x & 63;
    const int rstep = gridDim.x * 4;
    f32x4 a[NS], b[NS];
    auto load = [&](int row) {
        const float* p = src + (size_t)min(row, R - 1) * D + lane * 4;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(a[i]) : "v"(p + 512 * i) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(b[i]) : "v"(p + 512 * i + 256) : "memory");
        }
    };
    int row = blockIdx.x * 4 + threadIdx.x / 64;
    load(row);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (; row < R; row += rstep) {
        f32x4 va[NS], vb[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) { asm volatile("" : "+v"(a[i]), "+v"(b[i])); va[i] = a[i] * 1.0001f; vb[i] = b[i] * 0.9999f; }
        load(row + rstep);
        float* q = dst + (size_t)row * D + lane * 4;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(q + 512 * i), "v"(va[i]) : "memory");
            asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(q + 512 * i + 256), "v"(vb[i]) : "memory");
        }
        // the 2 NS stores are younger than the 2 NS loads of the next row: those have landed, the stores may still fly
        if (NS == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
}

template <int NS>
static void run_counted(const float* src, float* dst, int R, int grid) {
    const int D = 512 * NS;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((copy_rows_counted<NS>), dim3(grid), dim3(256), 0, 0, src, dst, R, D);
    (void)hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((copy_rows_counted<NS>), dim3(grid), dim3(256), 0, 0, src, dst, R, D);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("counted waits, asm nt loads/stores, persistent  D=%4d grid=%6d  %7.1f us  %6.0f GB/s\n", D, grid, ms / reps * 1e3,
           2.0 * R * D * 4 / (ms / reps * 1e-3) / 1e9);
}

template <int PAT, int NT, int PERSIST, int NS>
static void run(const float* src, float* dst, int R, int grid) {
    const int D = 512 * NS;
    if (!PERSIST) grid = R / 4;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((copy_rows<PAT, NT, PERSIST, NS>), dim3(grid), dim3(256), 0, 0, src, dst, R, D);
    (void)hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((copy_rows<PAT, NT, PERSIST, NS>), dim3(grid), dim3(256), 0, 0, src, dst, R, D);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 2.0 * R * D * 4;
    printf("pattern %d  nt %d  persist %d  D=%4d grid=%6d  %7.1f us  %6.0f GB/s\n", PAT, NT, PERSIST, D, grid, ms / reps * 1e3,
           bytes / (ms / reps * 1e-3) / 1e9);
}

int main() {
    const int R = 131072;
    float *src, *dst;
    (void)hipMalloc(&src, (size_t)R * 2048 * 4); (void)hipMalloc(&dst, (size_t)R * 2048 * 4);
    (void)hipMemset(src, 1, (size_t)R * 2048 * 4);
    run<0, 0, 1, 4>(src, dst, R, 4096); run<1, 0, 1, 4>(src, dst, R, 4096); run<2, 0, 1, 4>(src, dst, R, 4096);
    run<0, 1, 1, 4>(src, dst, R, 4096); run<0, 2, 1, 4>(src, dst, R, 4096); run<0, 3, 1, 4>(src, dst, R, 4096);
    run<1, 3, 1, 4>(src, dst, R, 4096); run<2, 3, 1, 4>(src, dst, R, 4096);
    run<0, 0, 0, 4>(src, dst, R, 0); run<0, 3, 0, 4>(src, dst, R, 0); run<1, 3, 0, 4>(src, dst, R, 0); run<2, 3, 0, 4>(src, dst, R, 0);
    run<0, 3, 1, 4>(src, dst, R, 1024); run<0, 3, 1, 4>(src, dst, R, 2048); run<0, 3, 1, 4>(src, dst, R, 8192); run<0, 3, 1, 4>(src, dst, R, 16384);
    run<0, 0, 1, 2>(src, dst, R, 4096); run<1, 0, 1, 2>(src, dst, R, 4096); run<0, 3, 1, 2>(src, dst, R, 4096); run<1, 3, 1, 2>(src, dst, R, 4096);
    run<2, 3, 1, 2>(src, dst, R, 4096); run<0, 3, 0, 2>(src, dst, R, 0); run<2, 3, 0, 2>(src, dst, R, 0);
    for (int g : {1024, 2048, 4096, 8192}) { run_counted<4>(src, dst, R, g); run_counted<2>(src, dst, R, g); }
    run<0, 3, 1, 4>(src, dst, R, 4096); run<0, 3, 0, 4>(src, dst, R, 0); run<0, 3, 1, 2>(src, dst, R, 4096); run<0, 3, 0, 2>(src, dst, R, 0);
    return 0;
}
