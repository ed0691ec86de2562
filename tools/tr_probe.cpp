// Probe of ds_read_b64_tr_b16 (gfx950 LDS transposing read) semantics, run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/tr_probe.cpp -o tools/tr_probe && tools/tr_probe
// LDS holds u16 element i = i.  Test 1: lane l reads with byte address 8 l (lane-linear); prints what each lane of the first 16-lane group
// got.  Test 2: a [k][n] row-major tile (pitch 128 elements): lanes 4 j + q of a group address row j, elements 4 q .. 4 q + 3 of 16 outs;
// the hypothesis (cdna_hip_programming.md, "ds_read_b64_tr_b16") is that lane i then receives column i of the 4 x 16 block: rows 0..3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(unsigned short* out, int mode) {
    extern __shared__ __attribute__((aligned(16))) unsigned short sm[];
    for (int i = threadIdx.x; i < 4096; i += 64) sm[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    int byte_addr;
    if (mode == 0) byte_addr = 8 * l;
    else {
        const int g = l >> 4, j = (l >> 2) & 3, q = l & 3;           // group g: outs 16 g .., row j, piece q
        byte_addr = (j * 128 + 16 * g + 4 * q) * 2;
    }
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)((char*)sm + byte_addr));
    out[4 * l] = v.x; out[4 * l + 1] = v.y; out[4 * l + 2] = v.z; out[4 * l + 3] = v.w;
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    std::vector<unsigned short> h(256);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 8192, 0, d, mode);
        hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) { printf("lane %2d: %4d %4d %4d %4d\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]); if (l == 19 && mode == 0) { printf("...\n"); l = 47; } }
        bool ok = true;
        if (mode == 1) for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) ok &= h[4 * l + r] == r * 128 + 16 * (l >> 4) + (l & 15);
        if (mode == 1) printf("hypothesis (lane i of group g gets rows 0..3 of column 16 g + i): %s\n", ok ? "CONFIRMED" : "REFUTED");
    }
    return 0;
}
