// What the matrix pipe itself sustains on this chip, with nothing to feed: register-resident v_mfma_f32_32x32x16_bf16 chains, two waves per SIMD on
// every CU, operands that change from instruction to instruction.  Run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_ceiling_probe.cpp -o tools/mfma_ceiling_probe && tools/mfma_ceiling_probe
// Prints, per operand data set (random normal bf16 / zeros), the achieved dense bf16 TFLOP/s, the shader clock the chip held (s_memtime cycles
// over s_memrealtime's 100 MHz wall clock, sampled by one wave per workgroup) and the fraction of the 2.5 PFLOP/s datasheet peak.  The fp32-grade
// "x3" GEMM needs six of these MFMAs per 16-deep k step: its ceiling is this number / 6.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(256) chain(const bf16x8* __restrict__ src, float* __restrict__ sink, long long* __restrict__ stamps, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] = src[(t * 8 + j) & 0xffff]; b[j] = src[(t * 8 + 4 + j) & 0xffff]; }
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    long long c0 = 0, w0 = 0;
    if ((threadIdx.x & 63) == 0) { c0 = clock64(); w0 = wall_clock64(); }
    for (int it = 0; it < iters; ++it) {
        // 16 MFMAs per trip: every accumulator sees a different (a, b) pair each time, no two consecutive MFMAs share an accumulator
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(j + s) & 3], b[(j + 2 * s + 1) & 3], acc[j], 0, 0, 0);
    }
    if ((threadIdx.x & 63) == 0) {
        stamps[2 * (blockIdx.x * 4 + (threadIdx.x >> 6))] = clock64() - c0;
        stamps[2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = wall_clock64() - w0;
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    sink[t] = s;
}

static unsigned short to_bf16(float f) { unsigned u; std::memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const int blocks_per_cu = argc > 2 ? atoi(argv[2]) : 2;          // 2 workgroups x 4 waves = two waves per SIMD
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * blocks_per_cu;
    std::vector<unsigned short> h(8 * 65536);
    bf16x8* src; float* sink; long long* stamps;
    hipMalloc(&src, h.size() * 2); hipMalloc(&sink, (size_t)blocks * 256 * 4); hipMalloc(&stamps, (size_t)blocks * 4 * 16);
    std::vector<long long> hs((size_t)blocks * 8);
    printf("%d CUs, %d workgroups of 4 waves, %d trips x 16 MFMAs per wave\n", prop.multiProcessorCount, blocks, iters);
    for (int mode = 0; mode < 3; ++mode) {
        srand(1);
        for (auto& v : h) {
            float x = 0.f;
            if (mode == 0) { float u1 = (rand() + 1.f) / (RAND_MAX + 2.f), u2 = rand() / (float)RAND_MAX; x = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2); }
            else if (mode == 1) x = (rand() & 1) ? 0.f : fabsf((rand() % 1000) * 1e-3f);          // ReLU-like: half zeros, positive rest
            v = to_bf16(x);
        }
        hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) {                                  // the third repetition is reported (clocks settled)
            hipEventRecord(e0);
            hipLaunchKernelGGL(chain, dim3(blocks), dim3(256), 0, 0, src, sink, stamps, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost);
        double cyc = 0, wall = 0;
        for (size_t i = 0; i < hs.size() / 2; ++i) { cyc += hs[2 * i]; wall += hs[2 * i + 1]; }
        const double ghz = cyc / wall * 0.1;                                    // s_memrealtime ticks at 100 MHz
        const double flops = (double)blocks * 4 * iters * 16 * 2.0 * 32 * 32 * 16;
        const double tf = flops / (ms * 1e-3) * 1e-12;
        const char* name = mode == 0 ? "random normal" : mode == 1 ? "half zeros (ReLU-like)" : "all zeros";
        printf("%-24s %8.2f ms  %8.1f TFLOP/s  = %.3f of 2500   shader clock %.2f GHz   pipe issue rate %.3f of (clock x 1024 SIMDs x 1 MFMA / 32 cycles)\n", name, ms, tf,
               tf / 2500.0, ghz, tf * 1e12 / (ghz * 1e9 * 1024 * (2.0 * 32 * 32 * 16 / 32.0)));
    }
    return 0;
}
