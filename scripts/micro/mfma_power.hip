// What the 1,400 W board cap leaves a bf16 MFMA loop that touches no memory at all: v_mfma_f32_16x16x32_bf16 on
// register-resident operands, 2 waves per SIMD on every CU, accumulators restarted every 64 products (K = 2048),
// with (0) one constant operand pair, (1) normal-random operands (eight A and eight B fragments per wave, cycling),
// (2) the same with half of the A elements zero (post-ReLU activations).  Board power and shader clock are sampled from
// the card's hwmon files while each loop runs.  The split GEMMs' 1.42-1.53 PFLOP/s are to be read against (1) / (2).
// Build: hipcc -O3 --offload-arch=gfx950 scripts/micro/mfma_power.hip -o scripts/micro/build/mfma_power
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dirent.h>
#include <random>
#include <string>
#include <thread>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NOPS>   // distinct operand fragments per side that cycle through the MFMAs (1 = constant operands)
__global__ __launch_bounds__(512) void mfma_loop(const u32x4* __restrict__ ops, float* out, int iters) {
    bf16x8 a[NOPS], b[NOPS];
    const size_t base = ((size_t)blockIdx.x * 512 + threadIdx.x) * 16;
#pragma unroll
    for (int k = 0; k < NOPS; ++k) {
        a[k] = __builtin_bit_cast(bf16x8, ops[base + k]);
        b[k] = __builtin_bit_cast(bf16x8, ops[base + 8 + k]);
    }
    f32x4 acc[4][4], keep = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int kk = 0; kk < 64; ++kk) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(i + kk) % NOPS], b[(j + 3 * kk) % NOPS], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) keep += acc[i][j];
    }
    if (keep[0] + keep[1] + keep[2] + keep[3] == 123.456f) out[threadIdx.x] = keep[0];
}

static std::string hwmon_of_device() {
    char id[64] = {0};
    if (hipDeviceGetPCIBusId(id, sizeof id, 0) != hipSuccess) return "";
    for (char* c = id; *c; ++c) *c = (char)tolower(*c);
    const std::string d = std::string("/sys/bus/pci/devices/") + id + "/hwmon";
    if (DIR* dir = opendir(d.c_str())) {
        while (dirent* e = readdir(dir))
            if (strncmp(e->d_name, "hwmon", 5) == 0) { std::string r = d + "/" + e->d_name; closedir(dir); return r; }
        closedir(dir);
    }
    return "";
}
static long read_long(const std::string& p) {
    FILE* f = fopen(p.c_str(), "r");
    if (!f) return -1;
    long v = -1;
    if (fscanf(f, "%ld", &v) != 1) v = -1;
    fclose(f);
    return v;
}

template <typename F>
static void run(const char* name, F launch, const std::string& hw, double flop_per_launch) {
    launch();
    hipDeviceSynchronize();
    std::atomic<bool> stop{false};
    std::vector<long> pw, fq;
    std::thread th([&] {
        while (!stop.load()) {
            if (!hw.empty()) { pw.push_back(read_long(hw + "/power1_input")); fq.push_back(read_long(hw + "/freq1_input")); }
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
    });
    const auto t0 = std::chrono::steady_clock::now();
    int n = 0;
    double el = 0;
    do {
        for (int i = 0; i < 10; ++i) launch();
        hipDeviceSynchronize();
        n += 10;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (el < 3.0);
    stop = true;
    th.join();
    double w = 0, g = 0; int c = 0;
    for (size_t i = pw.size() / 4; i < pw.size(); ++i) if (pw[i] > 0 && fq[i] > 0) { w += pw[i]; g += fq[i]; ++c; }
    printf("%-46s %8.1f us/launch  %6.0f W  %5.2f GHz  %7.1f TFLOP/s\n", name, el / n * 1e6, c ? w / c / 1e6 : NAN, c ? g / c / 1e9 : NAN,
           flop_per_launch * n / el / 1e12);
    fflush(stdout);
}

int main() {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const std::string hw = hwmon_of_device();
    printf("%d CUs, sensor %s, cap %.0f W\n", cus, hw.c_str(), hw.empty() ? NAN : read_long(hw + "/power1_cap") / 1e6);
    const int grid = cus, iters = 40;
    const size_t nfrag = (size_t)grid * 512 * 16;
    std::vector<unsigned> h(nfrag * 4), hz;
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    auto bf16 = [](float f) { unsigned u; memcpy(&u, &f, 4); return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16; };
    for (size_t i = 0; i < h.size(); ++i) h[i] = bf16(nd(rng)) | (bf16(nd(rng)) << 16);
    hz = h;                                         // variant: half of the A elements zero (fragments 0..7 of each lane)
    for (size_t l = 0; l < nfrag / 16; ++l)
        for (int k = 0; k < 8; ++k)
            for (int w = 0; w < 4; ++w) {
                unsigned& u = hz[(l * 16 + k) * 4 + w];
                const unsigned r = rng();
                if (r & 1) u &= 0xffff0000u;
                if (r & 2) u &= 0x0000ffffu;
            }
    u32x4 *d, *dz; float* out;
    hipMalloc(&d, h.size() * 4); hipMalloc(&dz, h.size() * 4); hipMalloc(&out, 4096);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dz, hz.data(), hz.size() * 4, hipMemcpyHostToDevice);
    const double fl = 2.0 * 16 * 16 * 32 * 16 * 64 * (double)iters * 8 * grid;     // per launch: 16 tiles x 64 products x iters x 8 waves x grid
    run("constant operand pair", [&] { hipLaunchKernelGGL(mfma_loop<1>, dim3(grid), dim3(512), 0, 0, d, out, iters); }, hw, fl);
    run("normal-random operands (8 + 8 fragments)", [&] { hipLaunchKernelGGL(mfma_loop<8>, dim3(grid), dim3(512), 0, 0, d, out, iters); }, hw, fl);
    run("normal-random, half of the A elements zero", [&] { hipLaunchKernelGGL(mfma_loop<8>, dim3(grid), dim3(512), 0, 0, dz, out, iters); }, hw, fl);
    return 0;
}
