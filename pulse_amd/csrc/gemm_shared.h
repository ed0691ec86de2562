// Pieces shared by the fp32-storage GEMM translation units (gemm_f32.hip: fp32 MFMA, bf16 MFMA over fp32 storage, the 128 x 128 x3 kernel;
// gemm_x3w.hip: the 256 x 256 x3 kernel): the launch argument block, the LDS slot permutation, raw-buffer and LDS access helpers.
#pragma once
#include "common.h"

namespace pulse {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned RSRC_FLAGS = 0x00020000u;         // raw buffer, 32-bit data format (gfx90a+ / gfx950)

struct GemmArgs {
    const float* A; const float* B; float* C; float* C2; const float* bias; const float* aux;
    int M, N, K;
    int lda, ldb, ldc, ldc2, ldaux;
    long long sA, sB, sC, sC2, sBias, sAux;   // batch strides (floats)
    int batch, splitk, kchunk;
    long long sSplit;                          // C slab stride per k-split (floats)
    int act;                                   // 0 none, 1 relu, 2 silu (EPI 0 only)
    int epi;                                   // 0 bias+act, 1 relu-grad mask, 2 silu-grad
    int tiles_m, tiles_n;
    int vec_epi;                               // all epilogue pointers / pitches are 16-byte aligned
    float* rowsum; long long sRowsum;          // <MC,MC> only: per-slab sums over k of A(k, m)  (bias gradient)
    long long* dbg;                            // optional per-workgroup clock stamps (tools/gemm_bench --clocks)
    int round_bf16;                            // outputs rounded to bf16-representable values (bf16 autocast semantics)
    unsigned* mask; int ldmask; long long sMask;   // ReLU bit mask (pulse_gemm_desc.relu_mask): written by EPI 0 + relu, read by EPI 1 when aux is null
};

// word of the ReLU bit mask that holds output row r (within its 64-row block: rows r, r + 8, .., r + 56 share a word), column group cg = col / 4
__device__ __forceinline__ long long mask_word(int r, int cg, int ldmask) { return (long long)((r >> 6) * 8 + (r & 7)) * ldmask + cg; }

// launch_gemm_x3w's answer when the device refuses the 256 x 256 tile's dynamic-LDS request: not an error, the launcher falls back (internal code)
constexpr int kWideTileUnavailable = -1000;

__device__ __forceinline__ int slot_of(int out) { return out ^ ((out >> 3) & 7); }

__device__ __forceinline__ f32x4 buf_load(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_store(f32x4 v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0);
}
__device__ __forceinline__ f32x4 lds_read(int byte_addr) {
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    return *reinterpret_cast<const f32x4*>(smem_c + byte_addr);
}
__device__ __forceinline__ void lds_write(int byte_addr, f32x4 v) {
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    *reinterpret_cast<f32x4*>(smem_c + byte_addr) = v;
}

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned fbits(float v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ float bitsf(unsigned v) { return __builtin_bit_cast(float, v); }
// (lo_elem, hi_elem) rounded to nearest-even bf16 and packed {hi, lo}: one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pack_rn(float lo_elem, float hi_elem) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){lo_elem, hi_elem}, bf16x2));
}

// gemm_x3w.hip: the x3 arithmetic on a 256 x 256 x 16 tile (one wave per SIMD, accumulators in AGPRs).  ``g.tiles_m`` / ``g.tiles_n`` are
// recomputed for the 256-wide tiling by the callee.  Returns a PULSE_* code.
int launch_gemm_x3w(const GemmArgs& g, bool akc, bool bkc, hipStream_t stream);
// gemm_x3s.hip: the x3 arithmetic for skinny outputs (N <= 96, A reduction-contiguous, plain / ReLU epilogue): 128 rows x all columns per
// workgroup.  Returns a PULSE_* code, or kWideTileUnavailable when the device refuses its dynamic LDS.
int launch_gemm_x3s(const GemmArgs& g, bool bkc, hipStream_t stream);

}  // namespace pulse
