// Stand-alone probe of the round-3 finding behind pulse_amd/csrc/build.py's "no packed fp32" rule:
//
//   on gfx950 (MI355X, ROCm 7.2) packed-fp32 VALU arithmetic (v_pk_mul_f32 / v_pk_add_f32) can return wrong values while ANOTHER
//   kernel's waves issue MFMAs on the same SIMD.
//
// Build and run on the GPU box (tools/_run.sh has the hipcc line):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pk_f32_probe.cpp -o tools/pk_f32_probe && tools/pk_f32_probe
//
// What it does: kernel ``chain`` runs, per lane, a 256-step recurrence v <- v * c + d twice -- once on float2 values (the compiler
// emits v_pk_mul_f32 + v_pk_add_f32; checked by grepping the ISA in tools/_run.sh) and once on two independent scalars (v_mul_f32 +
// v_add_f32) -- and writes both.  The host computes the same recurrence in fp32 (no contraction: both device forms round exactly
// like it) and counts lanes whose PACKED / SCALAR result differs, (a) alone, (b) while a second stream of the same process runs an
// MFMA-only kernel on every CU, (c) while it runs a VALU-only competitor.  Mismatches are tallied per 16-lane quarter of the wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <csignal>
#include <unistd.h>
#include <sys/wait.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int STEPS = 256;

__global__ void __launch_bounds__(64) chain(const float* __restrict__ in, float* __restrict__ out_pk, float* __restrict__ out_sc, int n) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const float a = in[4 * i], b = in[4 * i + 1], c = in[4 * i + 2], d = in[4 * i + 3];
    f32x2 v = {a, b};
    const f32x2 cc = {c, c}, dd = {d, d};
    float s0 = a, s1 = b;
#pragma unroll 8
    for (int k = 0; k < STEPS; ++k) {
        v = v * cc;                 // v_pk_mul_f32
        v = v + dd;                 // v_pk_add_f32
        asm volatile("" : "+v"(s0), "+v"(s1));          // keep the scalar chain out of the SLP vectoriser's sight
        s0 = s0 * c; s1 = s1 * c;
        asm volatile("" : "+v"(s0), "+v"(s1));
        s0 = s0 + d; s1 = s1 + d;
    }
    out_pk[2 * i] = v.x; out_pk[2 * i + 1] = v.y;
    out_sc[2 * i] = s0; out_sc[2 * i + 1] = s1;
}

// competitor 1: MFMA only (four independent accumulators, bf16 32x32x16), runs until ``iters`` are done
__global__ void __launch_bounds__(256) mfma_hammer(float* sink, int iters) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x - e)); }
    f32x16 acc[4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    if (s == 123.456f) sink[0] = s;
}

// competitor 2: plain VALU only
__global__ void __launch_bounds__(256) valu_hammer(float* sink, int iters) {
    float x = 1.0f + 1e-3f * threadIdx.x, y = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) { x = x * 1.0001f + y; y = y * 0.9999f + 1e-4f; }
    }
    if (x + y == 123.456f) sink[0] = x;
}

static void cpu_ref(const std::vector<float>& in, std::vector<float>& out, int n) {
    for (int i = 0; i < n; ++i) {
        volatile float a = in[4 * i], b = in[4 * i + 1];
        const float c = in[4 * i + 2], d = in[4 * i + 3];
        for (int k = 0; k < STEPS; ++k) {
            volatile float ta = a * c; a = ta + d;
            volatile float tb = b * c; b = tb + d;
        }
        out[2 * i] = a; out[2 * i + 1] = b;
    }
}

// the competitor as ANOTHER PROCESS (forked before this process touches HIP): back-to-back MFMA kernels until killed
static void hammer_process() {
    float* sink;
    CK(hipMalloc(&sink, 64));
    for (;;) {
        for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(mfma_hammer, dim3(1024), dim3(256), 0, 0, sink, 20000);
        CK(hipDeviceSynchronize());
    }
}

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 200;
    const bool other_process = argc > 2 && !strcmp(argv[2], "--process");
    pid_t child = -1;
    if (other_process) {
        child = fork();
        if (child == 0) { hammer_process(); return 0; }
        sleep(3);                                 // let the competitor come up
    }
    const int n = 64 * 2048;                     // 2048 single-wave workgroups: every SIMD gets several
    std::vector<float> in(4 * n), ref(2 * n), pk(2 * n), sc(2 * n);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.0f; };
    for (int i = 0; i < n; ++i) { in[4 * i] = rnd() - 0.5f; in[4 * i + 1] = rnd() - 0.5f; in[4 * i + 2] = 0.9f + 0.2f * rnd(); in[4 * i + 3] = 0.1f * (rnd() - 0.5f); }
    cpu_ref(in, ref, n);
    float *d_in, *d_pk, *d_sc, *d_sink;
    CK(hipMalloc(&d_in, in.size() * 4)); CK(hipMalloc(&d_pk, pk.size() * 4)); CK(hipMalloc(&d_sc, sc.size() * 4)); CK(hipMalloc(&d_sink, 64));
    CK(hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice));
    hipStream_t s_main, s_side;
    CK(hipStreamCreateWithFlags(&s_main, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s_side, hipStreamNonBlocking));

    const char* names[3] = {other_process ? "beside an MFMA kernel of ANOTHER PROCESS" : "alone", "beside an MFMA kernel (second stream)", "beside a VALU kernel (second stream)"};
    int total_bad_pk = 0;
    for (int mode = 0; mode < (other_process ? 1 : 3); ++mode) {
        long bad_pk = 0, bad_sc = 0, bad_launch = 0;
        long quarter[4] = {0, 0, 0, 0};
        for (int l = 0; l < launches; ++l) {
            if (mode == 1) hipLaunchKernelGGL(mfma_hammer, dim3(1024), dim3(256), 0, s_side, d_sink, 20000);
            if (mode == 2) hipLaunchKernelGGL(valu_hammer, dim3(1024), dim3(256), 0, s_side, d_sink, 20000);
            CK(hipMemsetAsync(d_pk, 0, pk.size() * 4, s_main)); CK(hipMemsetAsync(d_sc, 0, sc.size() * 4, s_main));
            hipLaunchKernelGGL(chain, dim3(n / 64), dim3(64), 0, s_main, d_in, d_pk, d_sc, n);
            CK(hipMemcpyAsync(pk.data(), d_pk, pk.size() * 4, hipMemcpyDeviceToHost, s_main));
            CK(hipMemcpyAsync(sc.data(), d_sc, sc.size() * 4, hipMemcpyDeviceToHost, s_main));
            CK(hipStreamSynchronize(s_main));
            long b0 = bad_pk;
            for (int i = 0; i < n; ++i) {
                const bool p = memcmp(&pk[2 * i], &ref[2 * i], 8) != 0, q = memcmp(&sc[2 * i], &ref[2 * i], 8) != 0;
                if (p) { ++bad_pk; ++quarter[(i & 63) >> 4]; }
                if (q) ++bad_sc;
            }
            if (bad_pk != b0) ++bad_launch;
            CK(hipStreamSynchronize(s_side));
        }
        printf("%-40s: packed-fp32 lanes wrong %ld (in %ld of %d launches; by wave quarter %ld %ld %ld %ld), scalar-fp32 lanes wrong %ld\n", names[mode], bad_pk,
               bad_launch, launches, quarter[0], quarter[1], quarter[2], quarter[3], bad_sc);
        if (mode == 1) total_bad_pk = (int)(bad_pk > 0);
    }
    if (child > 0) { kill(child, SIGKILL); waitpid(child, nullptr, 0); }
    return 0;
}
