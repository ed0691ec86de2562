// Where do the matrix-pipe cycles of the x3 GEMM go?  (round-3 verdict, missing #4.)  tools/mfma_ceiling_probe.cpp showed the bare pipe:
// 0.97 issued at 1.72 GHz on random operands.  The GEMM holds 0.65-0.69 at the same clock.  This probe adds the GEMM's feeding work to the
// same register-resident MFMA chain, one arm at a time, at the GEMM's own per-MFMA rates:
//   lds   : one ds_read_b128 per two MFMAs (gemm_x3_kernel: 12 fragment reads per 24 MFMAs), the fragments read ARE the next operands;
//   dma   : LDS-DMA fill (buffer_load_dwordx4 ... lds, 1 KB per wave instruction) streaming from a 1 GB buffer, D instructions per 16 MFMAs
//           (D = 3: 192 B per MFMA = the 128 x 128 x 16 fp32 tile's 170 B per MFMA, about 1.9-2.4 TB/s chip-wide at the GEMM's MFMA rate);
//   valu  : four dependent-free VALU ops per MFMA (the in-kernel three-way split costs 3.7);
//   all combinations.  Per arm: TFLOP/s, shader clock (s_memtime / s_memrealtime), pipe issue rate, and busy x clock / 2.4 GHz.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/mfma_feed_probe.cpp -o tools/mfma_feed_probe && tools/mfma_feed_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void_t;

constexpr int LDS_BYTES = 64 * 1024;           // two workgroups per CU, like gemm_x3_kernel

template <bool LDS, int DMA, int VALU>
__global__ void __launch_bounds__(256) chain(const bf16x8* __restrict__ src, const char* __restrict__ stream, long long stream_bytes_per_wg,
                                             float* __restrict__ sink, long long* __restrict__ stamps, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // random bf16 data in LDS (what the fragment reads return)
    for (int i = threadIdx.x; i < LDS_BYTES / 16; i += 256) reinterpret_cast<bf16x8*>(smem)[i] = src[(blockIdx.x * 977 + i) & 0xffff];
    __syncthreads();
    bf16x8 a[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] = src[(t * 8 + j) & 0xffff]; b[j] = src[(t * 8 + 4 + j) & 0xffff]; }
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float v0 = (float)lane, v1 = 1.0001f, v2 = 0.5f, v3 = 3.f;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(stream) + (long long)blockIdx.x * stream_bytes_per_wg, 0,
                                                                  (unsigned)stream_bytes_per_wg, 0x00020000u);
    // the DMA lands in the upper half of the LDS image, the fragment reads take the lower half (the real kernel double-buffers the same way)
    const int dma_lds = LDS_BYTES / 2 + wave * 4096;
    unsigned so = 0;
    long long c0 = 0, w0 = 0;
    if (lane == 0) { c0 = clock64(); w0 = wall_clock64(); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(j + s) & 3], b[(j + 2 * s + 1) & 3], acc[j], 0, 0, 0);
                if constexpr (VALU >= 4) {                               // 4 VALU per MFMA, independent of the MFMA chain
                    v0 = v0 * v1 + v2; v1 = v1 * 0.99999f + 1e-6f; v2 = v2 - v3 * 1e-7f; v3 = v3 + v0 * 1e-9f;
                } else if constexpr (VALU == 2) {
                    v0 = v0 * v1 + v2; v1 = v1 * 0.99999f + 1e-6f;
                } else if constexpr (VALU == 1) {
                    v0 = v0 * v1 + v2;
                }
                if constexpr (LDS) {
                    if (j & 1) {                                         // one 16-byte-per-lane read per two MFMAs; it replaces an operand used 3-4 MFMAs later
                        const int slot = (s * 2 + (j >> 1)) & 7;
                        const int addr = (lane * 16 + ((it * 8 + slot) & 31) * 1024) & (LDS_BYTES / 2 - 1);
                        const bf16x8 f = *reinterpret_cast<const bf16x8*>(smem + addr);
                        if (slot < 4) a[slot] = f; else b[slot - 4] = f;
                    }
                }
            }
            if constexpr (DMA > 0) {
                if (s < DMA) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(smem + dma_lds + (s & 3) * 1024), 16, lane * 16 + wave * 1024, so, 0, 0);
                    so += 4096;                                          // the four waves of the workgroup stream 4 KB per step together
                    if (so + 8192 > (unsigned)stream_bytes_per_wg) so = 0;
                }
            }
        }
        if constexpr (DMA > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DMA) : "memory");   // two trips of loads in flight
    }
    if (lane == 0) {
        stamps[2 * (blockIdx.x * 4 + wave)] = clock64() - c0;
        stamps[2 * (blockIdx.x * 4 + wave) + 1] = wall_clock64() - w0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = v0 + v1 + v2 + v3;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    sink[t] = s + (float)smem[dma_lds + lane];
}

static unsigned short to_bf16(float f) { unsigned u; std::memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }

template <bool LDS, int DMA, int VALU>
static void run(const char* name, int blocks, int iters, const bf16x8* src, const char* stream, long long per_wg, float* sink, long long* stamps) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(chain<LDS, DMA, VALU>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((chain<LDS, DMA, VALU>), dim3(blocks), dim3(256), LDS_BYTES, 0, src, stream, per_wg, sink, stamps, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    if (hipGetLastError() != hipSuccess) { printf("%-40s launch failed\n", name); return; }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> hs((size_t)blocks * 8);
    hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (size_t i = 0; i < hs.size() / 2; ++i) { cyc += hs[2 * i]; wall += hs[2 * i + 1]; }
    const double ghz = cyc / wall * 0.1;
    const double flops = (double)blocks * 4 * iters * 16 * 2.0 * 32 * 32 * 16;
    const double tf = flops / (ms * 1e-3) * 1e-12;
    const double issue = tf * 1e12 / (ghz * 1e9 * 1024 * (2.0 * 32 * 32 * 16 / 32.0));
    const double dma_tbs = DMA > 0 ? (double)blocks * 4 * iters * DMA * 1024.0 / (ms * 1e-3) * 1e-12 : 0.0;
    printf("%-40s %8.2f ms %8.1f TFLOP/s (x3-equivalent %6.1f)  clock %.2f GHz  pipe issue %.3f  busy x clock / 2.4 = %.3f  DMA %.2f TB/s\n", name, ms, tf, tf / 6, ghz,
           issue, issue * ghz / 2.4, dma_tbs);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 8000;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * 2;                      // two workgroups x 4 waves per CU = two waves per SIMD
    std::vector<unsigned short> h(8 * 65536);
    srand(1);
    for (auto& v : h) { float u1 = (rand() + 1.f) / (RAND_MAX + 2.f), u2 = rand() / (float)RAND_MAX; v = to_bf16(sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2)); }
    bf16x8* src; float* sink; long long* stamps; char* stream;
    const long long per_wg = 2LL << 20;                                    // 2 MB per workgroup = 1 GB streamed region (beyond the 256 MB Infinity Cache)
    hipMalloc(&src, h.size() * 2); hipMalloc(&sink, (size_t)blocks * 256 * 4); hipMalloc(&stamps, (size_t)blocks * 4 * 16); hipMalloc(&stream, per_wg * blocks);
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (long long o = 0; o < per_wg * blocks; o += (long long)h.size() * 2) hipMemcpy(stream + o, src, h.size() * 2, hipMemcpyDeviceToDevice);
    printf("%d CUs, %d workgroups of 4 waves, %d trips x 16 MFMAs per wave, random normal bf16 operands\n", prop.multiProcessorCount, blocks, iters);
    const long long l2_wg = 32LL << 10;     // 32 KB per workgroup: 64 workgroups per XCD x 32 KB = 2 MB, inside the XCD's 4 MB L2
    const long long mall_wg = 256LL << 10;  // 256 KB per workgroup = 128 MB: misses L2, inside the 256 MB Infinity Cache
    run<false, 0, 0>("mfma only", blocks, iters, src, stream, per_wg, sink, stamps);
    run<true, 0, 0>("+ ds_read_b128 per 2 MFMAs", blocks, iters, src, stream, per_wg, sink, stamps);
    run<false, 3, 0>("+ LDS-DMA 192 B / MFMA, L2-resident", blocks, iters, src, stream, l2_wg, sink, stamps);
    run<false, 3, 0>("+ LDS-DMA 192 B / MFMA, MALL-resident", blocks, iters, src, stream, mall_wg, sink, stamps);
    run<false, 3, 0>("+ LDS-DMA 192 B / MFMA, from HBM", blocks, iters, src, stream, per_wg, sink, stamps);
    run<false, 2, 0>("+ LDS-DMA 128 B / MFMA, from HBM", blocks, iters, src, stream, per_wg, sink, stamps);
    run<false, 0, 1>("+ 1 VALU / MFMA", blocks, iters, src, stream, per_wg, sink, stamps);
    run<false, 0, 2>("+ 2 VALU / MFMA", blocks, iters, src, stream, per_wg, sink, stamps);
    run<false, 0, 4>("+ 4 VALU / MFMA", blocks, iters, src, stream, per_wg, sink, stamps);
    run<true, 3, 0>("+ ds_read + LDS-DMA 192 (L2)", blocks, iters, src, stream, l2_wg, sink, stamps);
    run<true, 0, 4>("+ ds_read + 4 VALU", blocks, iters, src, stream, per_wg, sink, stamps);
    run<true, 3, 4>("+ ds_read + LDS-DMA 192 (L2) + 4 VALU", blocks, iters, src, stream, l2_wg, sink, stamps);
    run<true, 3, 4>("+ ds_read + LDS-DMA 192 (HBM) + 4 VALU", blocks, iters, src, stream, per_wg, sink, stamps);
    run<false, 0, 0>("mfma only (again)", blocks, iters, src, stream, per_wg, sink, stamps);
    return 0;
}
