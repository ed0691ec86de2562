// FP32 MFMA GEMM for the actor / critic / VAE MLPs (forward, dX and dW passes) on gfx950.
//
// Replaces every nn.Linear call (+ its autograd backward) on the PPO update path:
//   phc/learning/network_builder.py:105-124,245-261 (actor_mlp / critic_mlp / mu / value),
//   phc/learning/amp_network_builder.py:127-148,206-211 (eval_actor / eval_critic),
//   phc/learning/amp_network_z_builder.py:469-580 (PULSE VAE encoder / prior / decoder MLPs).
//
// Arithmetic: v_mfma_f32_32x32x2_f32 -- f32 inputs, f32 accumulate, bit-identical to an fmaf chain,
// 64 FLOP/clk/SIMD (157 TFLOP/s chip peak; gfx950 has no TF32/xf32 path).  The reference trains in
// fp32 (mixed_precision: False, learning/im.yaml:50), so fp32 is kept end to end.
//
// One kernel, three operand-layout instantiations, C[m][n] = sum_k A(m,k) * B(n,k):
//   <KC,KC>  forward   Y = X W^T      X [M][K],  W [N][K]      (both reduction-contiguous)
//   <KC,MC>  dX        dX = dY W      dY [M][N], W [N][K]      (B stored [red][out])
//   <MC,MC>  dW        dW = dY^T X    dY [M][N], X [M][K]      (both stored [red][out], split-K over M)
// Tiling: 128x128x32 block tile, 4 waves (2x2), each wave 2x2 MFMA tiles of 32x32 (64 accumulator
// VGPRs).  Operands are staged global -> registers -> LDS with 16-byte loads, double-buffered in LDS
// (one barrier per k-tile, next tile's global loads in flight during the 64 MFMAs of the current one).
// LDS images: reduction-contiguous operands as [out][k] with a 36-float pitch (conflict-free
// ds_read_b128: 16-lane groups land on 16 distinct 16-byte slots); [red][out] operands as [k][out]
// read with ds_read_b32 (32 consecutive floats per half-wave).  Because k is a pure reduction index
// the two wave halves take k-offsets {0..3} and {4..7} of every 8-k step, for A and B alike, which is
// what lets a single ds_read_b128 feed four consecutive MFMAs.
// 256 CUs / 8 XCDs: the 1-D grid is remapped so each XCD owns a contiguous band of m-tiles (A panels
// stay in that XCD's L2; the small weight matrix is shared by all).
#include <cstdlib>
#include <type_traits>
#include "common.h"

namespace pulse {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int PITCH_KC = BK + 4;    // [out][k] image
constexpr int PITCH_MC = BM + 4;    // [k][out] image
constexpr int TILE_FLOATS = BM * PITCH_KC;  // 4608 >= BK * PITCH_MC (4224)

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
};

// ---- global -> register staging -------------------------------------------------------------
// Loads are BRANCH-FREE (addresses are clamped to valid memory instead of predicated) so that nothing
// consumes a loaded register before the MFMA block: the reduction-tail / out-of-range zeroing happens in
// store_tile, after the compute of the current tile, when the data must have landed anyway.  (A first
// version masked inside the load and the compiler had to put s_waitcnt vmcnt(0) right behind every load,
// exposing the full HBM latency once per k-tile.)
template <bool KC>
__device__ __forceinline__ void load_tile(float4 (&r)[4], const float* __restrict__ P, int ld, int out0, int ext,
                                          int k0, int kbeg, int kend, int tid) {
    if constexpr (KC) {
        int k = k0 + (tid & 7) * 4;
        k = k < kend ? k : kbeg;                           // fully past the end: read something valid, masked later
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int row = out0 + (tid >> 3) + 32 * i;
            row = row < ext ? row : ext - 1;              // rows beyond the extent are never stored
            r[i] = *reinterpret_cast<const float4*>(P + (long long)row * ld + k);
        }
    } else {
        int m = out0 + (tid & 31) * 4;
        m = m < ext ? m : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int k = k0 + (tid >> 5) + 8 * i;
            k = k < kend ? k : kend - 1;
            r[i] = *reinterpret_cast<const float4*>(P + (long long)k * ld + m);
        }
    }
}

template <bool KC>
__device__ __forceinline__ void store_tile(float* __restrict__ s, const float4 (&r)[4], int k0, int kend, int tid) {
    if constexpr (KC) {
        const int k = k0 + (tid & 7) * 4;
        const bool full = k + 3 < kend;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 v = r[i];
            if (!full) {                                   // reduction tail: zero the lanes past K
                if (k >= kend) v.x = 0.f;
                if (k + 1 >= kend) v.y = 0.f;
                if (k + 2 >= kend) v.z = 0.f;
                v.w = 0.f;
            }
            *reinterpret_cast<float4*>(s + ((tid >> 3) + 32 * i) * PITCH_KC + (tid & 7) * 4) = v;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + (tid >> 5) + 8 * i;
            float4 v = r[i];
            if (k >= kend) v = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(s + ((tid >> 5) + 8 * i) * PITCH_MC + (tid & 31) * 4) = v;
        }
    }
}

// ---- per-unit forms (one 16-byte access each) for the interleaved main loop -------------------------------
template <bool KC>
__device__ __forceinline__ void load_unit(float4& r, const float* __restrict__ P, int ld, int out0, int ext, int k0, int kbeg, int kend,
                                          int tid, int i) {
    if constexpr (KC) {
        int k = k0 + (tid & 7) * 4;
        k = k < kend ? k : kbeg;
        int row = out0 + (tid >> 3) + 32 * i;
        row = row < ext ? row : ext - 1;
        r = *reinterpret_cast<const float4*>(P + (long long)row * ld + k);
    } else {
        int m = out0 + (tid & 31) * 4;
        m = m < ext ? m : 0;
        int k = k0 + (tid >> 5) + 8 * i;
        k = k < kend ? k : kend - 1;
        r = *reinterpret_cast<const float4*>(P + (long long)k * ld + m);
    }
}

template <bool KC>
__device__ __forceinline__ void store_unit(float* __restrict__ s, const float4& r, int k0, int kend, int tid, int i) {
    if constexpr (KC) {
        const int k = k0 + (tid & 7) * 4;
        float4 v = r;
        if (k + 3 >= kend) {
            if (k >= kend) v.x = 0.f;
            if (k + 1 >= kend) v.y = 0.f;
            if (k + 2 >= kend) v.z = 0.f;
            v.w = 0.f;
        }
        *reinterpret_cast<float4*>(s + ((tid >> 3) + 32 * i) * PITCH_KC + (tid & 7) * 4) = v;
    } else {
        const int k = k0 + (tid >> 5) + 8 * i;
        float4 v = r;
        if (k >= kend) v = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(s + ((tid >> 5) + 8 * i) * PITCH_MC + (tid & 31) * 4) = v;
    }
}

// fragment for one 32-wide tile: the 4 k-values this lane feeds to 4 consecutive MFMAs
template <bool KC>
__device__ __forceinline__ float4 load_frag(const float* __restrict__ s, int out_in_tile, int kk, int half) {
    if constexpr (KC) {
        return *reinterpret_cast<const float4*>(s + out_in_tile * PITCH_KC + kk + 4 * half);
    } else {
        const float* p = s + (kk + 4 * half) * PITCH_MC + out_in_tile;
        return make_float4(p[0], p[PITCH_MC], p[2 * PITCH_MC], p[3 * PITCH_MC]);
    }
}

template <bool AKC, bool BKC>
__global__ void __launch_bounds__(256) gemm_f32_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // stage s: A image at smem + s*2*TILE_FLOATS, B image right after it

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5;
    const int l31 = lane & 31;

    // XCD-aware bijective remap: block b runs on XCD b % 8; give each XCD a contiguous range of tiles
    const int ntile = g.tiles_m * g.tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3;
    const int q = ntile >> 3, rr = ntile & 7;
    const int id = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + loc;
    const int tm = id / g.tiles_n, tn = id - tm * g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const int z = blockIdx.y;
    const int bz = z / g.splitk, sp = z - bz * g.splitk;
    const int kbeg = sp * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);

    const float* A = g.A + bz * g.sA;
    const float* B = g.B + bz * g.sB;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[4], rb[4];
    float4 fa[2][2], fb[2][2];                                   // two fragment sets: one feeding MFMAs, one in flight from LDS
    float rsum[2] = {0.f, 0.f};
    const int nkt = (kend - kbeg + BK - 1) / BK;
    const int arow = wm * 64 + l31, brow = wn * 64 + l31;

    // fragment unit u of set SET: order fa0, fb0, fb1, fa1 = the order the MFMA pairs consume them
    auto frag_unit = [&](int set, int u, const float* a_s, const float* b_s, int kk) {
        if (u == 0) fa[set][0] = load_frag<AKC>(a_s, arow, kk, half);
        else if (u == 1) fb[set][0] = load_frag<BKC>(b_s, brow, kk, half);
        else if (u == 2) fb[set][1] = load_frag<BKC>(b_s, brow + 32, kk, half);
        else fa[set][1] = load_frag<AKC>(a_s, arow + 32, kk, half);
    };
    auto comp = [](const float4& v, int c) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; };

    // One k-tile = 64 MFMAs issued as 32 PAIRS (two accumulator chains alternate, so no MFMA waits on its predecessor).
    // With ONE wave per SIMD doing all the work, everything else has to ride in the issue shadow of those MFMAs
    // (each occupies the matrix pipe for 64 cycles): after every pair exactly one small unit of side work is issued --
    //   pairs  0-3   fragment reads for pairs  8-15   (set 1, k +8)
    //   pairs  8-11  fragment reads for pairs 16-23   (set 0, k +16)
    //   pairs 16-19  fragment reads for pairs 24-31   (set 1, k +24)
    //   pairs 19-26  the 8 register->LDS spills of tile t+1 (its global loads were issued a tile ago)
    //   after pair 27  the ONE barrier of the tile
    //   pairs 24-31  the 8 global loads of tile t+2 (load k re-uses the registers spill k just drained)
    //   pairs 28-31  fragment reads for pairs 0-7 of tile t+1 (set 0, other LDS stage)
    // sched_barrier(0) after every pair pins that order.
    auto tile = [&](auto last_tag, int t) {
        constexpr bool LAST = decltype(last_tag)::value;
        const int cur = t & 1;
        const float* a_s = smem + cur * 2 * TILE_FLOATS;
        const float* b_s = a_s + TILE_FLOATS;
        float* a_o = smem + (cur ^ 1) * 2 * TILE_FLOATS;
        float* b_o = a_o + TILE_FLOATS;
        const int k_next = kbeg + (t + 1) * BK, k_next2 = kbeg + (t + 2) * BK;
#pragma unroll
        for (int p = 0; p < 32; ++p) {
            const int grp = p >> 3, set = grp & 1, q = p & 7, i = q >> 2, c = q & 3;
            if constexpr (!AKC) {                  // dW pass: the A fragments are dY -- their k-sums are the bias gradient
                if (q == 0) {
                    rsum[0] += (fa[set][0].x + fa[set][0].y) + (fa[set][0].z + fa[set][0].w);
                    rsum[1] += (fa[set][1].x + fa[set][1].y) + (fa[set][1].z + fa[set][1].w);
                }
            }
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(fa[set][i], c), comp(fb[set][0], c), acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(fa[set][i], c), comp(fb[set][1], c), acc[i][1], 0, 0, 0);
            if (p < 4) frag_unit(1, p, a_s, b_s, 8);
            else if (p >= 8 && p < 12) frag_unit(0, p - 8, a_s, b_s, 16);
            else if (p >= 16 && p < 20) frag_unit(1, p - 16, a_s, b_s, 24);
            if constexpr (!LAST) {
                if (p >= 19 && p < 27) {
                    const int u = p - 19;
                    if (u < 4) store_unit<AKC>(a_o, ra[u], k_next, kend, tid, u);
                    else store_unit<BKC>(b_o, rb[u - 4], k_next, kend, tid, u - 4);
                }
                if (p == 27) {
                    __builtin_amdgcn_sched_barrier(0);
                    __syncthreads();
                }
                if (p >= 24) {              // unconditional: past the last tile the clamped addresses re-read valid memory, never stored
                    const int u = p - 24;
                    if (u < 4) load_unit<AKC>(ra[u], A, g.lda, m0, g.M, k_next2, kbeg, kend, tid, u);
                    else load_unit<BKC>(rb[u - 4], B, g.ldb, n0, g.N, k_next2, kbeg, kend, tid, u - 4);
                }
                if (p >= 28) frag_unit(0, p - 28, a_o, b_o, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if (nkt > 0) {
        load_tile<AKC>(ra, A, g.lda, m0, g.M, kbeg, kbeg, kend, tid);
        load_tile<BKC>(rb, B, g.ldb, n0, g.N, kbeg, kbeg, kend, tid);
        store_tile<AKC>(smem, ra, kbeg, kend, tid);
        store_tile<BKC>(smem + TILE_FLOATS, rb, kbeg, kend, tid);
    }
    __syncthreads();
    if (nkt > 1) {
        load_tile<AKC>(ra, A, g.lda, m0, g.M, kbeg + BK, kbeg, kend, tid);
        load_tile<BKC>(rb, B, g.ldb, n0, g.N, kbeg + BK, kbeg, kend, tid);
    }
    if (nkt > 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) frag_unit(0, u, smem, smem + TILE_FLOATS, 0);
    }
    for (int t = 0; t + 1 < nkt; ++t) tile(std::false_type{}, t);
    if (nkt > 0) tile(std::true_type{}, nkt - 1);
    if constexpr (!AKC) {
        if (g.rowsum && tn == 0 && wn == 0) {
            // this lane summed the k-offsets of its half; the other half's lane holds the rest of the same row
            const float r0 = rsum[0] + __shfl_xor(rsum[0], 32, 64);
            const float r1 = rsum[1] + __shfl_xor(rsum[1], 32, 64);
            if (half == 0) {
                float* rs = g.rowsum + bz * g.sRowsum + sp * g.sSplit;
                const int row = m0 + wm * 64 + l31;
                if (row < g.M) rs[row] = r0;
                if (row + 32 < g.M) rs[row + 32] = r1;
            }
        }
    }
    __syncthreads();                                              // the epilogue reuses the staging buffers
#undef PULSE_LOAD_FRAGS
#undef PULSE_MFMA_GROUP

    // ---- epilogue -----------------------------------------------------------------------------------
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
    float* C = g.C + bz * g.sC + sp * g.sSplit;
    float* C2 = g.C2 ? g.C2 + bz * g.sC2 : nullptr;
    const float* bias = g.bias ? g.bias + bz * g.sBias : nullptr;
    const float* aux = g.aux ? g.aux + bz * g.sAux : nullptr;

    if (g.vec_epi) {
        // Wide path: the accumulators are transposed through LDS (the staging buffers are free after the main
        // loop) so every global access of the epilogue -- C stores, aux loads, pre-activation stores -- is a
        // 16-byte access covering 512 contiguous bytes of one row per half-wave, instead of 64 dword stores and
        // 64 dword aux loads per lane.
        constexpr int CP = BN + 4;                     // 132-float pitch: ds_read_b128 lane groups stay conflict-free
        float* sC = smem;                              // 128 x 132 x 4 B = 67,584 B <= 73,728 B
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    sC[(wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * CP + wn * 64 + j * 32 + l31] = acc[i][j][r];
        __syncthreads();
        const int c4 = (tid & 31) * 4;
        const int col = n0 + c4;
        if (col < g.N) {
            const bool full = col + 3 < g.N;
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bias) {
                if (full) bv = *reinterpret_cast<const float4*>(bias + col);
                else { bv.x = bias[col]; if (col + 1 < g.N) bv.y = bias[col + 1]; if (col + 2 < g.N) bv.z = bias[col + 2]; }
            }
#pragma unroll 4
            for (int q = 0; q < 16; ++q) {
                const int rl = (tid >> 5) + 8 * q;
                const int row = m0 + rl;
                if (row >= g.M) continue;
                float4 v = *reinterpret_cast<const float4*>(sC + rl * CP + c4);
                float o[4] = {v.x, v.y, v.z, v.w};
                if (g.epi == 0) {
                    o[0] += bv.x; o[1] += bv.y; o[2] += bv.z; o[3] += bv.w;
                    if (g.act == 1) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) o[k] = fmaxf(o[k], 0.f);
                    } else if (g.act == 2) {
                        if (C2) {
                            float* p2 = C2 + (long long)row * g.ldc2 + col;
                            if (full) *reinterpret_cast<float4*>(p2) = make_float4(o[0], o[1], o[2], o[3]);
                            else for (int k = 0; k < 4 && col + k < g.N; ++k) p2[k] = o[k];
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) o[k] = o[k] / (1.f + __expf(-o[k]));
                    }
                } else {
                    const float* pa = aux + (long long)row * g.ldaux + col;
                    float a4[4] = {0.f, 0.f, 0.f, 0.f};
                    if (full) { const float4 t = *reinterpret_cast<const float4*>(pa); a4[0] = t.x; a4[1] = t.y; a4[2] = t.z; a4[3] = t.w; }
                    else for (int k = 0; k < 4 && col + k < g.N; ++k) a4[k] = pa[k];
                    if (g.epi == 1) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) o[k] = a4[k] > 0.f ? o[k] : 0.f;
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float sg = 1.f / (1.f + __expf(-a4[k]));
                            o[k] *= sg * (1.f + a4[k] * (1.f - sg));
                        }
                    }
                }
                float* pc = C + (long long)row * g.ldc + col;
                if (full) *reinterpret_cast<float4*>(pc) = make_float4(o[0], o[1], o[2], o[3]);
                else for (int k = 0; k < 4 && col + k < g.N; ++k) pc[k] = o[k];
            }
        }
        return;
    }

    // Scalar path (unaligned C / aux pitches): one dword per lane per accumulator register.
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + l31;
        if (col >= g.N) continue;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row >= g.M) continue;
                float v = acc[i][j][r];
                if (g.epi == 0) {
                    v += bv;
                    if (g.act == 1) {
                        v = fmaxf(v, 0.f);
                    } else if (g.act == 2) {
                        if (C2) C2[(long long)row * g.ldc2 + col] = v;   // keep the pre-activation for backward
                        v = v / (1.f + __expf(-v));
                    }
                } else if (g.epi == 1) {
                    v = aux[(long long)row * g.ldaux + col] > 0.f ? v : 0.f;
                } else {
                    const float zz = aux[(long long)row * g.ldaux + col];
                    const float sg = 1.f / (1.f + __expf(-zz));
                    v *= sg * (1.f + zz * (1.f - sg));
                }
                C[(long long)row * g.ldc + col] = v;
            }
        }
    }
}

// ---- deterministic reduction of split-K slabs (and of column-sum partials) ---------------------
__global__ void __launch_bounds__(256) reduce_slabs_kernel(const float* __restrict__ slabs, int nslab, long long slab_stride,
                                                          long long count, float* __restrict__ out, float scale) {
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < count; i += (long long)gridDim.x * blockDim.x * 4) {
        if (i + 3 < count) {
            float4 s = *reinterpret_cast<const float4*>(slabs + i);
            for (int k = 1; k < nslab; ++k) {
                const float4 v = *reinterpret_cast<const float4*>(slabs + k * slab_stride + i);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            s.x *= scale; s.y *= scale; s.z *= scale; s.w *= scale;
            *reinterpret_cast<float4*>(out + i) = s;
        } else {
            for (long long e = i; e < count; ++e) {
                float s = slabs[e];
                for (int k = 1; k < nslab; ++k) s += slabs[k * slab_stride + e];
                out[e] = s * scale;
            }
        }
    }
}

// ---- column sums (bias gradients): partial[chunk][n] = sum over the chunk's rows of X[m][n] ------
// HBM-bound (each dZ element read once).  256 threads = 64 column groups (one float4 = 4 columns each,
// so a row segment of 1 KiB is read per 64 lanes) x 4 row lanes; 4 independent loads in flight per thread.
__global__ void __launch_bounds__(256) colsum_partial_kernel(const float* __restrict__ X, int M, int N, int ld, int rows_per_chunk,
                                                            float* __restrict__ partial, long long ldp) {
    __shared__ float4 red[4][64];
    const int cg = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 256 + cg * 4;
    const int r0 = blockIdx.y * rows_per_chunk;
    const int r1 = min(M, r0 + rows_per_chunk);
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
    if (c < N) {
        const float* p = X + c;
        int r = r0 + rl;
        for (; r + 12 < r1; r += 16) {
            const float4 a = *reinterpret_cast<const float4*>(p + (long long)r * ld);
            const float4 b = *reinterpret_cast<const float4*>(p + (long long)(r + 4) * ld);
            const float4 cc = *reinterpret_cast<const float4*>(p + (long long)(r + 8) * ld);
            const float4 d = *reinterpret_cast<const float4*>(p + (long long)(r + 12) * ld);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
            s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
            s2.x += cc.x; s2.y += cc.y; s2.z += cc.z; s2.w += cc.w;
            s3.x += d.x; s3.y += d.y; s3.z += d.z; s3.w += d.w;
        }
        for (; r < r1; r += 4) {
            const float4 a = *reinterpret_cast<const float4*>(p + (long long)r * ld);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
        }
    }
    s0.x += s1.x + s2.x + s3.x; s0.y += s1.y + s2.y + s3.y; s0.z += s1.z + s2.z + s3.z; s0.w += s1.w + s2.w + s3.w;
    red[rl][cg] = s0;
    __syncthreads();
    if (rl == 0 && c < N) {
        const float4 a = red[0][cg], b = red[1][cg], cc = red[2][cg], d = red[3][cg];
        float* o = partial + (long long)blockIdx.y * ldp + c;
        o[0] = a.x + b.x + cc.x + d.x;
        if (c + 1 < N) o[1] = a.y + b.y + cc.y + d.y;
        if (c + 2 < N) o[2] = a.z + b.z + cc.z + d.z;
        if (c + 3 < N) o[3] = a.w + b.w + cc.w + d.w;
    }
}

}  // namespace pulse

using namespace pulse;

extern "C" {

int pulse_sizeof_gemm_desc(void) { return (int)sizeof(pulse_gemm_desc); }

int pulse_gemm_f32(const pulse_gemm_desc* d, pulse_stream_t s) {
    PULSE_REQUIRE(d != nullptr, "pulse_gemm_f32: null descriptor");
    PULSE_REQUIRE(d->M >= 0 && d->N >= 0 && d->K >= 0, "pulse_gemm_f32: negative size");
    if (d->M == 0 || d->N == 0 || d->batch == 0) return PULSE_OK;
    PULSE_REQUIRE(d->A && d->B && d->C, "pulse_gemm_f32: null operand");
    PULSE_REQUIRE(d->batch >= 1 && d->split_k >= 1, "pulse_gemm_f32: batch / split_k must be >= 1");
    PULSE_REQUIRE((d->lda % 4) == 0 && (d->ldb % 4) == 0, "pulse_gemm_f32: lda / ldb must be multiples of 4 floats");
    PULSE_REQUIRE((reinterpret_cast<uintptr_t>(d->A) & 15) == 0 && (reinterpret_cast<uintptr_t>(d->B) & 15) == 0,
                  "pulse_gemm_f32: A / B must be 16-byte aligned");
    PULSE_REQUIRE((d->stride_a % 4) == 0 && (d->stride_b % 4) == 0, "pulse_gemm_f32: batch strides must be multiples of 4 floats");
    const bool akc = d->a_layout == PULSE_GEMM_RED_CONTIG, bkc = d->b_layout == PULSE_GEMM_RED_CONTIG;
    PULSE_REQUIRE(!(!akc && bkc), "pulse_gemm_f32: layout combination (A out-contiguous, B reduction-contiguous) unsupported");
    // pitches must cover the float4 reads: reduction-contiguous rows up to roundup4(K), others up to roundup4(extent)
    const int k4 = (d->K + 3) & ~3;
    PULSE_REQUIRE(akc ? d->lda >= k4 : d->lda >= ((d->M + 3) & ~3), "pulse_gemm_f32: lda too small");
    PULSE_REQUIRE(bkc ? d->ldb >= k4 : d->ldb >= ((d->N + 3) & ~3), "pulse_gemm_f32: ldb too small");
    PULSE_REQUIRE(d->ldc >= d->N, "pulse_gemm_f32: ldc too small");
    PULSE_REQUIRE(d->epilogue >= 0 && d->epilogue <= 2 && d->activation >= 0 && d->activation <= 2, "pulse_gemm_f32: bad epilogue / activation");
    PULSE_REQUIRE(d->epilogue == 0 || d->aux != nullptr, "pulse_gemm_f32: gradient epilogue needs aux");
    PULSE_REQUIRE(d->rowsum == nullptr || (!akc && !bkc), "pulse_gemm_f32: rowsum needs the (OUT, OUT) layouts (dW pass)");
    PULSE_REQUIRE(d->split_k == 1 || (d->epilogue == 0 && d->activation == 0 && d->bias == nullptr),
                  "pulse_gemm_f32: split-K slabs carry no epilogue");

    GemmArgs g;
    g.A = d->A; g.B = d->B; g.C = d->C; g.C2 = d->C2; g.bias = d->bias; g.aux = d->aux;
    g.M = d->M; g.N = d->N; g.K = d->K;
    g.lda = d->lda; g.ldb = d->ldb; g.ldc = d->ldc; g.ldc2 = d->ldc2; g.ldaux = d->ldaux;
    g.sA = d->stride_a; g.sB = d->stride_b; g.sC = d->stride_c; g.sC2 = d->stride_c2; g.sBias = d->stride_bias; g.sAux = d->stride_aux;
    g.batch = d->batch; g.splitk = d->split_k;
    int kchunk = (d->K + d->split_k - 1) / d->split_k;
    kchunk = ((kchunk + BK - 1) / BK) * BK;
    g.kchunk = kchunk > 0 ? kchunk : BK;
    g.sSplit = d->split_stride;
    g.act = d->activation; g.epi = d->epilogue;
    g.rowsum = d->rowsum; g.sRowsum = d->stride_rowsum;
    g.tiles_m = (d->M + BM - 1) / BM; g.tiles_n = (d->N + BN - 1) / BN;
    auto al16 = [](const void* p, long long ld, long long st) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && (ld % 4) == 0 && (st % 4) == 0; };
    g.vec_epi = al16(d->C, d->ldc, d->stride_c) && (d->split_stride % 4) == 0 && (!d->aux || al16(d->aux, d->ldaux, d->stride_aux)) &&
                (!d->C2 || al16(d->C2, d->ldc2, d->stride_c2)) && (!d->bias || al16(d->bias, 4, d->stride_bias));
    static const size_t lds_extra = getenv("PULSE_GEMM_LDS_EXTRA") ? (size_t)atoi(getenv("PULSE_GEMM_LDS_EXTRA")) : 0;   // tuning knob
    const size_t lds = sizeof(float) * 4 * TILE_FLOATS + lds_extra;   // 73,728 B -> two workgroups per CU
    const dim3 grid((unsigned)(g.tiles_m * g.tiles_n), (unsigned)(d->batch * d->split_k));
    // The 72 KiB dynamic-LDS opt-in is a per-function attribute: set it ONCE per instantiation (calling
    // hipFuncSetAttribute on every launch serialises the host against the stream).
    static bool attr_done[3] = {false, false, false};
    hipError_t e = hipSuccess;
#define LAUNCH(IDX, AK, BK_)                                                                                       \
    if (!attr_done[IDX]) {                                                                                        \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_kernel<AK, BK_>),                          \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                            \
        if (e != hipSuccess) return fail(PULSE_ERR_LAUNCH, "pulse_gemm_f32: LDS attribute: %s", hipGetErrorString(e)); \
        attr_done[IDX] = true;                                                                                    \
    }                                                                                                             \
    hipLaunchKernelGGL((gemm_f32_kernel<AK, BK_>), grid, dim3(256), lds, as_stream(s), g)
    if (akc && bkc) { LAUNCH(0, true, true); }
    else if (akc && !bkc) { LAUNCH(1, true, false); }
    else { LAUNCH(2, false, false); }
#undef LAUNCH
    return check_launch("pulse_gemm_f32");
}

int pulse_reduce_slabs(const float* slabs, int32_t num_slabs, int64_t slab_stride, int64_t count, float* out, float scale,
                       pulse_stream_t s) {
    PULSE_REQUIRE(num_slabs >= 1 && count >= 0, "pulse_reduce_slabs: bad sizes");
    if (count == 0) return PULSE_OK;
    PULSE_REQUIRE(slabs && out, "pulse_reduce_slabs: null pointer");
    PULSE_REQUIRE((slab_stride % 4) == 0 && (reinterpret_cast<uintptr_t>(slabs) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0,
                  "pulse_reduce_slabs: 16-byte alignment required");
    long long blocks = (count / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(s), slabs, num_slabs, slab_stride, count, out, scale);
    return check_launch("pulse_reduce_slabs");
}

int pulse_colsum_partial(const float* x, int32_t m, int32_t n, int32_t ld, int32_t num_chunks, float* partial, int64_t ld_partial,
                         pulse_stream_t s) {
    PULSE_REQUIRE(m >= 0 && n >= 0 && num_chunks >= 1, "pulse_colsum_partial: bad sizes");
    if (n == 0) return PULSE_OK;
    PULSE_REQUIRE(x && partial && ld >= ((n + 3) & ~3) && ld_partial >= n, "pulse_colsum_partial: bad pointers / pitches (ld must cover roundup4(n))");
    PULSE_REQUIRE((ld % 4) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "pulse_colsum_partial: x rows must be 16-byte aligned");
    const int rows = (m + num_chunks - 1) / num_chunks;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)num_chunks), dim3(256), 0, as_stream(s), x, m, n, ld,
                       rows > 0 ? rows : 1, partial, (long long)ld_partial);
    return check_launch("pulse_colsum_partial");
}
}
