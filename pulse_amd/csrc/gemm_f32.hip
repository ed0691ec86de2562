// FP32 MFMA GEMM for the actor / critic / VAE MLPs (forward, dX and dW passes) on gfx950.
//
// Replaces every nn.Linear call (+ its autograd backward) on the PPO update path:
//   phc/learning/network_builder.py:105-124,245-261 (actor_mlp / critic_mlp / mu / value),
//   phc/learning/amp_network_builder.py:127-148,206-211 (eval_actor / eval_critic),
//   phc/learning/amp_network_z_builder.py:469-580 (PULSE VAE encoder / prior / decoder MLPs).
//
// Arithmetic: v_mfma_f32_32x32x2_f32 -- f32 inputs, f32 accumulate, bit-identical to an fmaf chain,
// 64 FLOP/clk/SIMD (157 TFLOP/s chip peak at 2.4 GHz; gfx950 has no TF32/xf32 path).  The reference trains in
// fp32 (mixed_precision: False, learning/im.yaml:50), so fp32 is kept end to end.
//
// One kernel, three operand-layout instantiations, C[m][n] = sum_k A(m,k) * B(n,k):
//   <KC,KC>  forward   Y = X W^T      X [M][K],  W [N][K]      (both reduction-contiguous)
//   <KC,MC>  dX        dX = dY W      dY [M][N], W [N][K]      (B stored [red][out])
//   <MC,MC>  dW        dW = dY^T X    dY [M][N], X [M][K]      (both stored [red][out], split-K over M)
// Tiling: 128x128x32 block tile, 4 waves (2x2), each wave 2x2 MFMA tiles of 32x32 (64 accumulator registers), two
// workgroups per CU (LDS-limited), so every SIMD holds two waves that fill each other's stalls.
//
// What the round-2 measurements (tools/gemm_bench --clocks: per-workgroup s_memtime stamps) said, and what this
// version does about it:
//   * beside an fp32-MFMA-saturating partner wave every VALU instruction of the other wave waits for a gap between
//     two 64-cycle MFMAs, and every VALU instruction of the MFMA wave itself delays its next MFMA: VALU work is the
//     scarce resource.  The old epilogue (~400 VALU per wave) took 29k cycles per tile beside a busy partner (9.8k
//     alone) and the dW main loop lost 14 % to bias-gradient adds and address arithmetic.
//   * so: global loads are buffer loads (per-lane byte offset computed ONCE, the k advance rides in the scalar
//     offset), LDS addresses are per-lane constants + immediates (the LDS stage is a template parameter), the
//     reduction-tail masks exist only in the tile that stages the last k-tile, the bias is the initial value of the
//     accumulators, the fast epilogue is ds_write / ds_read_b128 / activation / buffer_store with scalar row offsets.
//   * ONE LDS image for all layouts, [k-chunk of 4][out][4 k] in 16-byte slots, 129 slots per k-chunk block, slot =
//     out ^ ((out >> 3) & 7): reduction-contiguous operands are stored as loaded, [red][out] operands are transposed
//     4x4 in registers on the way in, and every fragment is one conflict-free ds_read_b128 feeding four MFMAs (k is a
//     pure reduction index, so the two wave halves take k-chunks 2g and 2g+1 of every 8-k group, A and B alike).
// 256 CUs / 8 XCDs: the 1-D grid is remapped so each XCD owns a contiguous band of m-tiles (A panels stay in that
// XCD's L2; the small weight matrix is shared by all).
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "gemm_shared.h"

namespace pulse {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int KC_SLOTS = 129;                        // 16-byte slots per k-chunk block: 128 outs + 1 pad slot
constexpr int IMG_BYTES = 8 * KC_SLOTS * 16;         // one operand tile: 8 k-chunks x 129 slots = 16,512 B
constexpr int STAGE_BYTES = 2 * IMG_BYTES;           // A image + B image
constexpr int LDS_BYTES = 2 * STAGE_BYTES;           // two stages = 66,048 B -> two workgroups per CU
constexpr int CP = BN;                               // epilogue transpose pitch (floats): 128 x 128 x 4 B = 65,536 B

// Per-thread staging state of one operand: 4 in-flight 16-byte loads, their (constant) buffer byte offsets and the
// (constant) LDS byte addresses their data goes to.
//   KC (reduction-contiguous): load i = row (tid >> 3) + 32 i, k-chunk tid & 7      -> one slot, stored as loaded
//   MC ([red][out]):           load i = k row 4 (tid >> 5) + i, outs 4 (tid & 31).. -> 4x4 transpose, store j = out 4L + j
template <bool KC>
struct Stager {
    f32x4 r[4];
    int voff[4];
    int lds[4];
    int kpos;            // KC: first k of this thread's slot inside the tile (0, 4, .. 28); MC: first k row (0, 4, .. 28)

    __device__ __forceinline__ void init(int tid, int ld, int ext_rel /* valid outs from the tile origin, >= 1 */, int img_off) {
        if constexpr (KC) {
            const int kc = tid & 7, row0 = tid >> 3;
            kpos = kc * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + 32 * i;
                const int rr = row < ext_rel ? row : ext_rel - 1;       // rows beyond the extent are never stored: read a valid one
                voff[i] = (rr * ld + kc * 4) * 4;
                lds[i] = img_off + (kc * KC_SLOTS + slot_of(row)) * 16;
            }
        } else {
            const int kch = tid >> 5, L = tid & 31;
            kpos = kch * 4;
            const int col = 4 * L < ext_rel ? 4 * L : 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                voff[i] = ((kch * 4 + i) * ld + col) * 4;
                lds[i] = img_off + (kch * KC_SLOTS + slot_of(4 * L + i)) * 16;
            }
        }
    }
    // single short tile (K < 32 from the window start): lanes past the readable range re-read k position 0 (masked later)
    __device__ __forceinline__ void clamp_short(int tid, int ld, int readable /* k positions that may be read */) {
        if constexpr (KC) {
            if (kpos >= readable) {
#pragma unroll
                for (int i = 0; i < 4; ++i) voff[i] -= kpos * 4;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (kpos + i >= readable) voff[i] -= (kpos + i) * ld * 4;
        }
    }
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, int soff, int i) { r[i] = buf_load(rs, voff[i], soff); }

    // store unit u (0..3) into the stage at byte offset st; MASKED zeroes k positions outside [lo, hi)
    template <bool MASKED>
    __device__ __forceinline__ void store(int st, int u, int lo, int hi) {
        if constexpr (KC) {
            f32x4 v = r[u];
            if constexpr (MASKED) {
                if (kpos < lo || kpos >= hi) v.x = 0.f;
                if (kpos + 1 < lo || kpos + 1 >= hi) v.y = 0.f;
                if (kpos + 2 < lo || kpos + 2 >= hi) v.z = 0.f;
                if (kpos + 3 < lo || kpos + 3 >= hi) v.w = 0.f;
            }
            lds_write(st + lds[u], v);
        } else {
            f32x4 v = {r[0][u], r[1][u], r[2][u], r[3][u]};            // out 4L + u, k rows kpos .. kpos + 3
            if constexpr (MASKED) {
                if (kpos < lo || kpos >= hi) v.x = 0.f;
                if (kpos + 1 < lo || kpos + 1 >= hi) v.y = 0.f;
                if (kpos + 2 < lo || kpos + 2 >= hi) v.z = 0.f;
                if (kpos + 3 < lo || kpos + 3 >= hi) v.w = 0.f;
            }
            lds_write(st + lds[u], v);
        }
    }
};

// ---- epilogue (shared by all main loops; WM = 32-row MFMA tiles per wave: tile height 64 WM) -----------------------------------------------------------
// C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
// ``round_bf16``: results leave as bf16-representable fp32 values -- what a bf16 autocast Linear hands to the next op.
__device__ __forceinline__ float rbf(float v) { return (float)(__bf16)v; }

template <int WM>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, f32x16 (&acc)[WM][2], int tid, int m0, int n0, int bz, int sp, int wm, int wn,
                                              int half, int l31) {
    if (g.round_bf16) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = rbf(acc[i][j][r]);
    }
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* C = g.C + bz * g.sC + sp * g.sSplit;
    float* C2 = g.C2 ? g.C2 + bz * g.sC2 : nullptr;
    const float* aux = g.aux ? g.aux + bz * g.sAux : nullptr;
    unsigned* mask = g.mask ? g.mask + bz * g.sMask : nullptr;     // ReLU bit mask: written by the relu forward, read by relu-grad when aux is null
    const bool use_mask = g.epi == 1 && aux == nullptr;

    // The accumulators are transposed through LDS (the staging buffers are free after the main loop) so every global access of the vector
    // epilogue is a 16-byte access covering 512 contiguous bytes of one row per half-wave.  The scalar path (unaligned pitches) reads the
    // same image: the 64 accumulator registers are dead from here on in EVERY path -- with them live across the per-element address
    // arithmetic of the scalar path the weight-gradient instantiation spilled 351 VGPRs (round-3 verdict, weak #1).
    const bool fast = g.vec_epi && m0 + 64 * WM <= g.M && n0 + BN <= g.N && (g.epi == 1 || g.epi == 3 || (g.epi == 0 && g.act < 2));
    const int c4 = (tid & 31) * 4;
    const int rl0 = tid >> 5;
    f32x4 ax[8 * WM];
    unsigned mw[WM];                           // this thread's mask words: rows rl0 + 8 q of 64-row block b = q / 8, columns c4 .. c4 + 3
    if (fast && use_mask) {
#pragma unroll
        for (int b = 0; b < WM; ++b) mw[b] = mask[mask_word(m0 + 64 * b + rl0, (n0 + c4) >> 2, g.ldmask)];
        // [r6, last hours] The bits are expanded HERE into the registers the aux path would have loaded (+1 / -1 per element) and the store loop
        // below is the aux path's.  The loop this replaces -- ``nb = mw >> 4 q; v.x = (nb & 1) ? v.x : 0`` after the barrier -- was bit-identical in
        // every test and returned garbage in 12 - 48 elements of a row now and then as soon as another stream's or process's GEMMs ran beside the
        // launch (tools/gemm_contend_probe.py: 5 412 wrong elements in 4 000 launches, 0 alone, 0 for the aux path); this form:
        // 0 in 3 000, and the agent-level tests pass 10 / 10 beside a GEMM-hammering process with the masks forced on.
#pragma unroll
        for (int q = 0; q < 8 * WM; ++q) {
            const unsigned nb = mw[q >> 3] >> (4 * (q & 7));
            ax[q] = (f32x4){(nb & 1u) ? 1.f : -1.f, (nb & 2u) ? 1.f : -1.f, (nb & 4u) ? 1.f : -1.f, (nb & 8u) ? 1.f : -1.f};
        }
    } else if (fast && g.epi != 0) {           // relu-grad / multiply-by-aux: the 16 aux loads fly while the accumulators go through LDS
        const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(aux) + (long long)m0 * g.ldaux + n0, 0,
                                                                            0xffffffffu, RSRC_FLAGS);
        const int voX = (rl0 * g.ldaux + c4) * 4;
#pragma unroll
        for (int q = 0; q < 8 * WM; ++q) ax[q] = buf_load(rsX, voX, q * 8 * g.ldaux * 4);
    }
    float* sC = smem;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                sC[(wm * 32 * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * CP + wn * 64 + j * 32 + l31] = acc[i][j][r];
    __syncthreads();
    if (g.dbg && tid == 0) g.dbg[8 * (blockIdx.y * gridDim.x + blockIdx.x) + 6] = clock64();
    if (g.vec_epi) {
        if (fast) {
            // fast path (full tile; none / relu / relu-grad): per 16-byte store one ds_read_b128, the activation, one
            // buffer store whose row advance is a scalar offset -- no per-access address arithmetic on the VALU
            const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(C + (long long)m0 * g.ldc + n0, 0, 0xffffffffu, RSRC_FLAGS);
            const int voC = (rl0 * g.ldc + c4) * 4;
            const int ldsC = (rl0 * CP + c4) * 4;
            if (g.epi == 0) {
                const bool relu = g.act == 1;
                const bool wmask = relu && mask != nullptr;
                unsigned w = 0;
#pragma unroll
                for (int q = 0; q < 8 * WM; ++q) {
                    f32x4 v = lds_read(ldsC + q * 8 * CP * 4);
                    if (wmask) {
                        w |= ((v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u)) << (4 * (q & 7));
                        if ((q & 7) == 7) { mask[mask_word(m0 + 64 * (q >> 3) + rl0, (n0 + c4) >> 2, g.ldmask)] = w; w = 0; }
                    }
                    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    buf_store(v, rsC, voC, q * 8 * g.ldc * 4);
                }
            } else if (g.epi == 1) {                   // relu-grad: aux = the activations, or the +-1 expansion of the forward's bit mask (above)
#pragma unroll
                for (int q = 0; q < 8 * WM; ++q) {
                    const f32x4 a = ax[q];
                    f32x4 v = lds_read(ldsC + q * 8 * CP * 4);
                    v.x = a.x > 0.f ? v.x : 0.f; v.y = a.y > 0.f ? v.y : 0.f; v.z = a.z > 0.f ? v.z : 0.f; v.w = a.w > 0.f ? v.w : 0.f;
                    buf_store(v, rsC, voC, q * 8 * g.ldc * 4);
                }
            } else {                                   // EPI_MUL_AUX: aux holds the producer's stored activation derivative
#pragma unroll
                for (int q = 0; q < 8 * WM; ++q) {
                    const f32x4 a = ax[q];
                    f32x4 v = lds_read(ldsC + q * 8 * CP * 4);
                    v.x *= a.x; v.y *= a.y; v.z *= a.z; v.w *= a.w;
                    buf_store(v, rsC, voC, q * 8 * g.ldc * 4);
                }
            }
            return;
        }
        const int col = n0 + c4;
        if (col < g.N) {
            const bool full = col + 3 < g.N;
            const bool wmask = g.epi == 0 && g.act == 1 && mask != nullptr;
            unsigned w = 0;
#pragma unroll 4
            for (int q = 0; q < 8 * WM; ++q) {
                const int rl = rl0 + 8 * q;
                const int row = m0 + rl;
                // ragged tiles: the word of a 64-row block is stored after its last row slot (rows past M contribute zero bits; the buffer
                // covers roundup64(M) rows), and read once at the block's first slot
                if (use_mask && (q & 7) == 0 && m0 + 64 * (q >> 3) < g.M) w = mask[mask_word(m0 + 64 * (q >> 3) + rl0, col >> 2, g.ldmask)];
                if (row >= g.M) {
                    if (wmask && (q & 7) == 7 && m0 + 64 * (q >> 3) < g.M) { mask[mask_word(m0 + 64 * (q >> 3) + rl0, col >> 2, g.ldmask)] = w; w = 0; }
                    continue;
                }
                float4 v = *reinterpret_cast<const float4*>(sC + rl * CP + c4);
                float o[4] = {v.x, v.y, v.z, v.w};
                if (wmask) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) w |= (col + k < g.N && o[k] > 0.f ? 1u : 0u) << (4 * (q & 7) + k);
                    if ((q & 7) == 7) { mask[mask_word(m0 + 64 * (q >> 3) + rl0, col >> 2, g.ldmask)] = w; w = 0; }
                }
                if (use_mask) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] = ((w >> (4 * (q & 7) + k)) & 1u) ? o[k] : 0.f;
                    float* pc = C + (long long)row * g.ldc + col;
                    if (full) *reinterpret_cast<float4*>(pc) = make_float4(o[0], o[1], o[2], o[3]);
                    else for (int k = 0; k < 4 && col + k < g.N; ++k) pc[k] = o[k];
                    continue;
                }
                if (g.epi == 0) {
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
                    } else if (g.act == 3) {             // SiLU whose C2 receives d silu / d z (the backward pass then multiplies: EPI_MUL_AUX)
                        float d[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float sg = 1.f / (1.f + __expf(-o[k]));
                            d[k] = sg * (1.f + o[k] * (1.f - sg));
                            o[k] = o[k] / (1.f + __expf(-o[k]));
                        }
                        float* p2 = C2 + (long long)row * g.ldc2 + col;
                        if (full) *reinterpret_cast<float4*>(p2) = make_float4(d[0], d[1], d[2], d[3]);
                        else for (int k = 0; k < 4 && col + k < g.N; ++k) p2[k] = d[k];
                    }
                } else {
                    const float* pa = aux + (long long)row * g.ldaux + col;
                    float a4[4] = {0.f, 0.f, 0.f, 0.f};
                    if (full) { const float4 t = *reinterpret_cast<const float4*>(pa); a4[0] = t.x; a4[1] = t.y; a4[2] = t.z; a4[3] = t.w; }
                    else for (int k = 0; k < 4 && col + k < g.N; ++k) a4[k] = pa[k];
                    if (g.epi == 1) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) o[k] = a4[k] > 0.f ? o[k] : 0.f;
                    } else if (g.epi == 3) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) o[k] *= a4[k];
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

    // Scalar path (unaligned C / aux pitches): one dword per lane per tile element, read back from the LDS image (a plain loop: this path
    // serves odd test shapes, not the training shapes).
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + l31;
        if (col >= g.N) continue;
        for (int i = 0; i < WM; ++i) {
#pragma unroll 1
            for (int r = 0; r < 16; ++r) {
                const int rl = wm * 32 * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int row = m0 + rl;
                if (row >= g.M) continue;
                float v = sC[rl * CP + wn * 64 + j * 32 + l31];
                if (g.epi == 0) {
                    if (g.act == 1) {
                        v = fmaxf(v, 0.f);
                    } else if (g.act == 2) {
                        if (C2) C2[(long long)row * g.ldc2 + col] = v;   // keep the pre-activation for backward
                        v = v / (1.f + __expf(-v));
                    } else if (g.act == 3) {
                        const float sg = 1.f / (1.f + __expf(-v));
                        C2[(long long)row * g.ldc2 + col] = sg * (1.f + v * (1.f - sg));     // d silu / d z for the backward pass (EPI_MUL_AUX)
                        v = v / (1.f + __expf(-v));
                    }
                } else if (g.epi == 1) {
                    v = aux[(long long)row * g.ldaux + col] > 0.f ? v : 0.f;
                } else if (g.epi == 3) {
                    v *= aux[(long long)row * g.ldaux + col];
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

template <bool AKC, bool BKC>
__global__ void __launch_bounds__(256) gemm_f32_kernel(const GemmArgs g) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5;
    const int l31 = lane & 31;

    // XCD-aware workgroup order (map_workgroup above): a band of output tiles per XCD, or a k range per XCD for split-K launches
    const WgMap wg = map_workgroup(g.tiles_m * g.tiles_n, g.batch, g.splitk);
    const int id = wg.id;
    const int tm = id / g.tiles_n, tn = id - tm * g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int bz = wg.bz, sp = wg.sp;
    const int kbeg = sp * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int klen = kend - kbeg;
    const int nkt = (klen + BK - 1) / BK;

    long long dbg_c0 = 0, dbg_w0 = 0, dbg_c1 = 0, dbg_w1 = 0;
    if (g.dbg) { dbg_c0 = clock64(); dbg_w0 = wall_clock64(); }

    // The last k-tile is read through a window that ENDS at roundup4(kend) (never past a row's pitch / the last k row):
    // it may overlap the tile before it, so its image is valid for positions [lo, hi) only.
    const int r4 = (klen + 3) & ~3;
    const int wlast = r4 > BK ? r4 - BK : 0;                 // window start of the last tile, relative to kbeg
    const int lo = nkt > 0 ? (nkt - 1) * BK - wlast : 0;
    const int hi = klen - wlast;

    // buffer resources based at this workgroup's tile origin and k start (all offsets stay far below 2^31)
    const float* Ab = g.A + bz * g.sA + (AKC ? (long long)m0 * g.lda + kbeg : (long long)kbeg * g.lda + m0);
    const float* Bb = g.B + bz * g.sB + (BKC ? (long long)n0 * g.ldb + kbeg : (long long)kbeg * g.ldb + n0);
    // [red][out] operands: the shifted window of the last k-tile may name up to three k rows past the operand's last row; the buffer
    // extent makes those loads return zero without touching memory (they are masked anyway)
    const unsigned recA = AKC ? 0xffffffffu : (unsigned)((g.K - kbeg - 1) * g.lda + ((min(BM, g.M - m0) + 3) & ~3)) * 4u;
    const unsigned recB = BKC ? 0xffffffffu : (unsigned)((g.K - kbeg - 1) * g.ldb + ((min(BN, g.N - n0) + 3) & ~3)) * 4u;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Ab), 0, recA, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Bb), 0, recB, RSRC_FLAGS);
    const int kstepA = AKC ? 4 : g.lda * 4, kstepB = BKC ? 4 : g.ldb * 4;        // bytes per unit of k
    auto koff = [&](int t) { return t == nkt - 1 ? wlast : t * BK; };              // scalar

    Stager<AKC> sa;
    Stager<BKC> sb;
    sa.init(tid, g.lda, g.M - m0, 0);
    sb.init(tid, g.ldb, g.N - n0, IMG_BYTES);
    if (nkt == 1 && r4 < BK) { sa.clamp_short(tid, g.lda, AKC ? r4 : klen); sb.clamp_short(tid, g.ldb, BKC ? r4 : klen); }

    // fragment read addresses: lane (l31, half) reads out (wm|wn) * 64 + {0, 32} + l31, k-chunk 2g + half
    const int frA0 = (half * KC_SLOTS + slot_of(wm * 64 + l31)) * 16;
    const int frA1 = (half * KC_SLOTS + slot_of(wm * 64 + 32 + l31)) * 16;
    const int frB0 = IMG_BYTES + (half * KC_SLOTS + slot_of(wn * 64 + l31)) * 16;
    const int frB1 = IMG_BYTES + (half * KC_SLOTS + slot_of(wn * 64 + 32 + l31)) * 16;

    // accumulators start from the bias (EPI 0): the epilogue has no bias add left
    f32x16 acc[2][2];
    {
        float b0 = 0.f, b1 = 0.f;
        if (g.epi == 0 && g.bias) {
            const float* bias = g.bias + bz * g.sBias;
            const int c0 = n0 + wn * 64 + l31;
            if (c0 < g.N) b0 = bias[c0];
            if (c0 + 32 < g.N) b1 = bias[c0 + 32];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][0][r] = b0; acc[i][1][r] = b1; }
    }

    f32x4 fa[2][2], fb[2][2];                                    // two fragment sets: one feeding MFMAs, one in flight from LDS
    float rsum[2] = {0.f, 0.f};
    const bool do_rs = !AKC && g.rowsum != nullptr && tn == 0 && wn == 0;      // wave-uniform

    // fragment unit u of set SET for k-group G of the stage at byte offset ST: order fa0, fb0, fb1, fa1 = consumption order
    auto frag_unit = [&](int set, int u, int st, int G) {
        const int o = st + 2 * G * KC_SLOTS * 16;
        if (u == 0) fa[set][0] = lds_read(frA0 + o);
        else if (u == 1) fb[set][0] = lds_read(frB0 + o);
        else if (u == 2) fb[set][1] = lds_read(frB1 + o);
        else fa[set][1] = lds_read(frA1 + o);
    };

    // One k-tile = 64 MFMAs issued as 32 PAIRS (two accumulator chains alternate, so no MFMA waits on its predecessor);
    // after every pair exactly one small unit of side work (none of them VALU in the steady state for KC operands):
    //   pairs  0-3   fragment reads for pairs  8-15   (set 1, k-group 1)
    //   pairs  8-11  fragment reads for pairs 16-23   (set 0, k-group 2)
    //   pairs 16-19  fragment reads for pairs 24-31   (set 1, k-group 3)
    //   pairs 19-26  the 8 register->LDS stores of tile t+1 (its global loads were issued a tile ago)
    //   after pair 27  the ONE barrier of the tile
    //   pairs 24-31  the 8 buffer loads of tile t+2 (load k re-uses the registers store k just drained; A's four stores
    //                are done by pair 22, B's by pair 26)
    //   pairs 28-31  fragment reads for pairs 0-7 of tile t+1 (set 0, other LDS stage)
    // MODE 0 steady, 1 = stages the LAST tile (masked stores, no further loads), 2 = last tile (compute only).
    auto tile = [&](auto mode_tag, auto stage_tag, int t) {
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr int CUR = decltype(stage_tag)::value * STAGE_BYTES, OTH = STAGE_BYTES - CUR;
        const int soA = koff(t + 2) * kstepA, soB = koff(t + 2) * kstepB;
#pragma unroll
        for (int p = 0; p < 32; ++p) {
            const int grp = p >> 3, set = grp & 1, q = p & 7, i = q >> 2, c = q & 3;
            if constexpr (!AKC) {                  // dW pass: the A fragments are dY -- their k-sums are the bias gradient
                if (q == 0 && do_rs) {
                    rsum[0] += (fa[set][0].x + fa[set][0].y) + (fa[set][0].z + fa[set][0].w);
                    rsum[1] += (fa[set][1].x + fa[set][1].y) + (fa[set][1].z + fa[set][1].w);
                }
            }
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][i][c], fb[set][0][c], acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][i][c], fb[set][1][c], acc[i][1], 0, 0, 0);
            if (p < 4) frag_unit(1, p, CUR, 1);
            else if (p >= 8 && p < 12) frag_unit(0, p - 8, CUR, 2);
            else if (p >= 16 && p < 20) frag_unit(1, p - 16, CUR, 3);
            if constexpr (MODE != 2) {
                if (p >= 19 && p < 27) {
                    const int u = p - 19;
                    if (u < 4) sa.template store<MODE == 1>(OTH, u, lo, hi);
                    else sb.template store<MODE == 1>(OTH, u - 4, lo, hi);
                }
                if (p == 27) {
                    __builtin_amdgcn_sched_barrier(0);
                    __syncthreads();
                }
                if constexpr (MODE == 0) {
                    if (p >= 24) {
                        const int u = p - 24;
                        if (u < 4) sa.load(rsA, soA, u);
                        else sb.load(rsB, soB, u - 4);
                    }
                }
                if (p >= 28) frag_unit(0, p - 28, OTH, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;

    if (nkt > 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { sa.load(rsA, koff(0) * kstepA, u); sb.load(rsB, koff(0) * kstepB, u); }
        if (nkt == 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { sa.template store<true>(0, u, lo, hi); sb.template store<true>(0, u, lo, hi); }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) { sa.template store<false>(0, u, 0, BK); sb.template store<false>(0, u, 0, BK); }
        }
    }
    __syncthreads();
    if (nkt > 1) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { sa.load(rsA, koff(1) * kstepA, u); sb.load(rsB, koff(1) * kstepB, u); }
    }
    if (nkt > 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) frag_unit(0, u, 0, 0);
    }
    {
        int t = 0;
        for (; t + 3 < nkt; t += 2) { tile(I0{}, I0{}, t); tile(I0{}, I1{}, t + 1); }      // steady tiles t < nkt - 2, two per trip
        if (t + 2 < nkt) {                                                               // one more steady tile: parity flips
            tile(I0{}, I0{}, t);
            tile(I1{}, I1{}, t + 1);
            tile(I2{}, I0{}, t + 2);
        } else if (t + 2 == nkt) {
            tile(I1{}, I0{}, t);
            tile(I2{}, I1{}, t + 1);
        } else if (t + 1 == nkt) {
            tile(I2{}, I0{}, t);
        }
    }
    if constexpr (!AKC) {
        if (do_rs) {
            // this lane summed the k-chunks of its half; the other half's lane holds the rest of the same row
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
    if (g.dbg) { dbg_c1 = clock64(); dbg_w1 = wall_clock64(); }
    struct DbgStamp {
        const GemmArgs& g; long long c0, w0, c1, w1;
        __device__ ~DbgStamp() {
            if (g.dbg && threadIdx.x == 0) {
                long long* o = g.dbg + 8 * (blockIdx.y * gridDim.x + blockIdx.x);
                o[0] = c0; o[1] = w0; o[2] = c1; o[3] = w1; o[4] = clock64(); o[5] = wall_clock64();
                o[7] = ((long long)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
            }
        }
    } dbg_stamp{g, dbg_c0, dbg_w0, dbg_c1, dbg_w1};

    gemm_epilogue(g, acc, tid, m0, n0, bz, sp, wm, wn, half, l31);
}

// =====================================================================================================================
// bf16 MFMA variant (BASELINE.json configs[4]: "AMP discriminator + PPO ... bf16"; the reference's autocast site is
// phc/learning/amp_agent.py:671, common_agent.py:426,461).  Storage stays fp32 (fp32 master weights, fp32 activations in HBM);
// operands are rounded to bf16 ON THE WAY INTO LDS, products accumulate in fp32 on v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA
// rate), and with round_bf16 the outputs leave as bf16-representable values -- exactly what a bf16 autocast Linear computes.
// Same 128x128 tile, same LDS image geometry with 8 bf16 per 16-byte slot, so one k-tile is 64 deep and one ds_read_b128 is the
// whole operand of one MFMA.  At this MFMA rate the kernel is bound by the fp32 operand traffic (L2 / HBM), not by the matrix pipe:
// a plain double-buffered loop, all fragment reads of a stage issued before the tile's barrier (same race rule as above).
// =====================================================================================================================
constexpr int BK16 = 64;

__device__ __forceinline__ bf16x8 pack8(const float (&v)[8]) {
    bf16x8 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bf16x2 t = __builtin_convertvector((f32x2){v[2 * k], v[2 * k + 1]}, bf16x2);
        o[2 * k] = t[0]; o[2 * k + 1] = t[1];
    }
    return o;
}

template <bool KC>
struct Stager16 {
    f32x4 r[8];
    int voff[8];         // KC: 4 used (each slot = two adjacent 16-byte loads); MC: 8 k rows
    int lds[4];
    int kpos;            // first k position of this thread's slot(s) inside the 64-deep tile

    __device__ __forceinline__ void init(int tid, int ld, int ext_rel, int img_off) {
        if constexpr (KC) {
            const int kc = tid & 7, row0 = tid >> 3;
            kpos = kc * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + 32 * i;
                const int rr = row < ext_rel ? row : ext_rel - 1;
                voff[i] = (rr * ld + kc * 8) * 4;
                lds[i] = img_off + (kc * KC_SLOTS + slot_of(row)) * 16;
            }
        } else {
            const int kch = tid >> 5, L = tid & 31;
            kpos = kch * 8;
            const int col = 4 * L < ext_rel ? 4 * L : 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) voff[i] = ((kch * 8 + i) * ld + col) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) lds[j] = img_off + (kch * KC_SLOTS + slot_of(4 * L + j)) * 16;
        }
    }
    __device__ __forceinline__ void clamp_short(int ld, int readable) {
        if constexpr (KC) {
            // a 16-byte load covers 4 k positions: the first half is readable if kpos < readable, the second if kpos + 4 < readable
            if (kpos >= readable) {                                  // nothing of this slot is readable: both halves re-read position 0
#pragma unroll
                for (int i = 0; i < 4; ++i) voff[i] -= kpos * 4;
                kpos_hi_ok = false;
            } else {
                kpos_hi_ok = kpos + 4 < readable;                     // second half past the readable range: re-read the first half
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (kpos + i >= readable) voff[i] -= (kpos + i) * ld * 4;
        }
    }
    bool kpos_hi_ok = true;
    __device__ __forceinline__ void load_all(__amdgpu_buffer_rsrc_t rs, int soff) {
        if constexpr (KC) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                r[2 * i] = buf_load(rs, voff[i], soff);
                r[2 * i + 1] = buf_load(rs, voff[i] + (kpos_hi_ok ? 16 : 0), soff);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] = buf_load(rs, voff[i], soff);
        }
    }
    template <bool MASKED>
    __device__ __forceinline__ void store(int st, int u, int lo, int hi) {
        float v[8];
        if constexpr (KC) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = r[2 * u][e]; v[4 + e] = r[2 * u + 1][e]; }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = r[i][u];
        }
        if constexpr (MASKED) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (kpos + e < lo || kpos + e >= hi) v[e] = 0.f;
        }
        extern __shared__ __attribute__((aligned(16))) char smem_c[];
        *reinterpret_cast<bf16x8*>(smem_c + st + lds[u]) = pack8(v);
    }
};

template <bool AKC, bool BKC>
__global__ void __launch_bounds__(256) gemm_bf16_kernel(const GemmArgs g) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const WgMap wg = map_workgroup(g.tiles_m * g.tiles_n, g.batch, g.splitk);
    const int id = wg.id;
    const int tm = id / g.tiles_n, tn = id - tm * g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int bz = wg.bz, sp = wg.sp;
    const int kbeg = sp * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int klen = kend - kbeg;
    const int nkt = (klen + BK16 - 1) / BK16;
    const int r4 = (klen + 3) & ~3;
    const int wlast = r4 > BK16 ? r4 - BK16 : 0;
    const int lo = nkt > 0 ? (nkt - 1) * BK16 - wlast : 0;
    const int hi = klen - wlast;

    const float* Ab = g.A + bz * g.sA + (AKC ? (long long)m0 * g.lda + kbeg : (long long)kbeg * g.lda + m0);
    const float* Bb = g.B + bz * g.sB + (BKC ? (long long)n0 * g.ldb + kbeg : (long long)kbeg * g.ldb + n0);
    // [red][out] operands: the shifted window of the last k-tile may name up to three k rows past the operand's last row; the buffer
    // extent makes those loads return zero without touching memory (they are masked anyway)
    const unsigned recA = AKC ? 0xffffffffu : (unsigned)((g.K - kbeg - 1) * g.lda + ((min(BM, g.M - m0) + 3) & ~3)) * 4u;
    const unsigned recB = BKC ? 0xffffffffu : (unsigned)((g.K - kbeg - 1) * g.ldb + ((min(BN, g.N - n0) + 3) & ~3)) * 4u;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Ab), 0, recA, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Bb), 0, recB, RSRC_FLAGS);
    const int kstepA = AKC ? 4 : g.lda * 4, kstepB = BKC ? 4 : g.ldb * 4;
    auto koff = [&](int t) { return t == nkt - 1 ? wlast : t * BK16; };

    Stager16<AKC> sa;
    Stager16<BKC> sb;
    sa.init(tid, g.lda, g.M - m0, 0);
    sb.init(tid, g.ldb, g.N - n0, IMG_BYTES);
    if (nkt == 1 && r4 < BK16) { sa.clamp_short(g.lda, AKC ? r4 : klen); sb.clamp_short(g.ldb, BKC ? r4 : klen); }

    const int frA0 = (half * KC_SLOTS + slot_of(wm * 64 + l31)) * 16;
    const int frA1 = (half * KC_SLOTS + slot_of(wm * 64 + 32 + l31)) * 16;
    const int frB0 = IMG_BYTES + (half * KC_SLOTS + slot_of(wn * 64 + l31)) * 16;
    const int frB1 = IMG_BYTES + (half * KC_SLOTS + slot_of(wn * 64 + 32 + l31)) * 16;

    f32x16 acc[2][2];
    {
        float b0 = 0.f, b1 = 0.f;
        if (g.epi == 0 && g.bias) {
            const float* bias = g.bias + bz * g.sBias;
            const int c0 = n0 + wn * 64 + l31;
            if (c0 < g.N) b0 = bias[c0];
            if (c0 + 32 < g.N) b1 = bias[c0 + 32];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][0][r] = b0; acc[i][1][r] = b1; }
    }
    float rsum[2] = {0.f, 0.f};
    const bool do_rs = !AKC && g.rowsum != nullptr && tn == 0 && wn == 0;

    bf16x8 fa[4][2], fb[4][2];                                     // the four 16-k groups of one tile
    auto frags = [&](int G, int st) {
        extern __shared__ __attribute__((aligned(16))) char smem_c[];
        const int o = st + 2 * G * KC_SLOTS * 16;
        fa[G][0] = *reinterpret_cast<const bf16x8*>(smem_c + frA0 + o);
        fb[G][0] = *reinterpret_cast<const bf16x8*>(smem_c + frB0 + o);
        fb[G][1] = *reinterpret_cast<const bf16x8*>(smem_c + frB1 + o);
        fa[G][1] = *reinterpret_cast<const bf16x8*>(smem_c + frA1 + o);
    };
    auto mma = [&](int G) {
        if constexpr (!AKC) {
            if (do_rs) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { rsum[0] += (float)fa[G][0][e]; rsum[1] += (float)fa[G][1][e]; }
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[G][i], fb[G][0], acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[G][i], fb[G][1], acc[i][1], 0, 0, 0);
        }
    };

    if (nkt > 0) {
        sa.load_all(rsA, koff(0) * kstepA);
        sb.load_all(rsB, koff(0) * kstepB);
        if (nkt == 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { sa.template store<true>(0, u, lo, hi); sb.template store<true>(0, u, lo, hi); }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) { sa.template store<false>(0, u, 0, BK16); sb.template store<false>(0, u, 0, BK16); }
        }
    }
    __syncthreads();
    if (nkt > 1) { sa.load_all(rsA, koff(1) * kstepA); sb.load_all(rsB, koff(1) * kstepB); }
    if (nkt > 0) frags(0, 0);
    for (int t = 0; t < nkt; ++t) {
        const int cur = (t & 1) * STAGE_BYTES, oth = STAGE_BYTES - cur;
        const bool has_next = t + 1 < nkt, stage_last = t + 2 == nkt;
        frags(1, cur);
        mma(0);
        if (has_next) {
            if (stage_last) {
#pragma unroll
                for (int u = 0; u < 4; ++u) sa.template store<true>(oth, u, lo, hi);
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) sa.template store<false>(oth, u, 0, BK16);
            }
        }
        frags(2, cur);
        mma(1);
        if (has_next) {
            if (stage_last) {
#pragma unroll
                for (int u = 0; u < 4; ++u) sb.template store<true>(oth, u, lo, hi);
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) sb.template store<false>(oth, u, 0, BK16);
            }
        }
        frags(3, cur);                                           // every read of this stage is issued before the barrier
        if (has_next) __syncthreads();
        if (t + 2 < nkt) { sa.load_all(rsA, koff(t + 2) * kstepA); sb.load_all(rsB, koff(t + 2) * kstepB); }
        mma(2);
        mma(3);
        if (has_next) frags(0, oth);
    }
    if constexpr (!AKC) {
        if (do_rs) {
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
    __syncthreads();
    gemm_epilogue(g, acc, tid, m0, n0, bz, sp, wm, wn, half, l31);
}

// =====================================================================================================================
// fp32 GEMM on the bf16 matrix pipe ("x3": three-way operand split, six products).
//
// gfx950's fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at the VALU's rate, 1/16 of the bf16 MFMA.  An fp32 number is the sum of three
// bf16 numbers to within 2^-27 of itself (8 + 8 + 8 significand bits: a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2), round
// to nearest even, the remainders exact; |a2| <= 2^-9 |a|, |a3| <= 2^-18 |a|), every bf16 x bf16 product is exact in fp32, and the
// products that matter at fp32 precision are the six with plane indices i + j <= 2: the dropped ones are below 2^-26 of the product,
// a quarter of an fp32 ulp.  So C = sum_k a b is computed as six v_mfma_f32_32x32x16_bf16 per 16-deep k step, all into the same fp32
// accumulator: fp32-grade results (fewer accumulator roundings per k than the fp32 MFMA's one per 2 k) at up to 16 / 6 = 2.67x the
// fp32 MFMA's matrix-pipe ceiling.  Inputs, outputs and storage are fp32; this is an fp32 GEMM, not a reduced-precision one (tests:
// same fp64-referenced tolerances as the fp32 MFMA kernel; the exactness, linearity and tile-position-independence properties hold bit
// for bit).  Non-finite inputs give NaN (inf - inf in the split).
//
// Tile 128 x 128 x 16, 4 waves, each 2 x 2 MFMA tiles.  LDS image per operand and stage: 3 planes x [2 k-chunks of 8][132 slots][16 B]
// (slot = out ^ ((out >> 3) & 7); a ds_write_b128 is serviced in groups of 8 consecutive lanes over 32 banks, i.e. 4 rows x 2 k-chunks: the chunk stride
// 132 = 4 mod 8 puts the two chunks of a row in different halves of the 128-byte bank window), two stages.  Per thread and k-tile:
// 8 elements of A and 8 of B are split (about 44 VALU each, spread over the first MFMAs of the tile), 6 ds_write_b128, 12
// ds_read_b128 (next tile's fragments, second register set), 24 MFMAs.  Global loads: reduction-contiguous operands 2 x 16 B per
// thread (two lanes per row), [red][out] operands 8 dwords per thread (lane = out: no register transpose).  The buffer resources
// carry the operand's true extent, so loads the hardware range check catches (rows / outs past the operand, k rows past its end) return
// zero without touching memory and no address is clamped.  Correctness does not lean on the check: the k tail is zeroed by a compare in the
// tile that stages the last k-tile, and rows / outs beyond the extent only feed outputs that are never stored.
// =====================================================================================================================
constexpr int XK = 16;
#ifndef X3_CSTRIDE
#define X3_CSTRIDE 132
#endif
constexpr int X_CSTRIDE = X3_CSTRIDE;               // 16-byte slots per 8-k chunk block (see the store-pattern note above)
constexpr int X_PLANE = 2 * X_CSTRIDE * 16;          // 4,224 B
constexpr int X_IMG = 3 * X_PLANE;                   // 12,672 B per operand
constexpr int X_STAGE = 2 * X_IMG;                   // 25,344 B
#ifndef X3_BARRIER_GAP
#define X3_BARRIER_GAP 13
#endif
constexpr int X_LDS = BM * CP * 4;                   // 65,536 B: the epilogue transpose (>= 2 stages = 50,688 B) -> two workgroups per CU


template <bool KC>
struct StagerX {
    float v[2][8];       // two register sets (tile parity): loads run two tiles ahead of their split.  KC: two 16-byte loads; MC: 8 dwords
    int voff;            // per-lane byte offset (constant); the k advance and MC's row advance are scalar offsets
    int lds;
    int kpos;
    int ld4;             // MC: bytes per k row (wave-uniform)

    __device__ __forceinline__ void init(int tid, int ld, int img_off) {
        ld4 = ld * 4;
        if constexpr (KC) {
            const int row = tid >> 1, kc = tid & 1;
            kpos = kc * 8;
            voff = (row * ld + kc * 8) * 4;
            lds = img_off + (kc * X_CSTRIDE + slot_of(row)) * 16;
        } else {
            const int out = tid & 127, kch = tid >> 7;
            kpos = kch * 8;
            voff = (kch * 8 * ld + out) * 4;
            lds = img_off + (kch * X_CSTRIDE + slot_of(out)) * 16;
        }
    }
    template <int S>
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, int soff) {
        if constexpr (KC) {
            const f32x4 a = buf_load(rs, voff, soff), b = buf_load(rs, voff, soff + 16);
            v[S][0] = a.x; v[S][1] = a.y; v[S][2] = a.z; v[S][3] = a.w; v[S][4] = b.x; v[S][5] = b.y; v[S][6] = b.z; v[S][7] = b.w;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[S][i] = bitsf(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff + i * ld4, 0));
        }
    }
    template <int S, bool MASKED>
    __device__ __forceinline__ void mask(int hi) {
        if constexpr (MASKED) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (kpos + e >= hi) v[S][e] = 0.f;
        }
    }
    template <int S>
    __device__ __forceinline__ float sum8() const {
        return ((v[S][0] + v[S][1]) + (v[S][2] + v[S][3])) + ((v[S][4] + v[S][5]) + (v[S][6] + v[S][7]));
    }
    // split element pair k (elements 2k, 2k+1) into the three planes' packed dwords: round-to-nearest-even at every level
    // (v_cvt_pk_bf16_f32), remainders exact (the difference of a float and its 8-bit rounding is representable)
    u32x4 p0, p1, p2;
    template <int S>
    __device__ __forceinline__ void split_pair(int k) {
        const float a = v[S][2 * k], b = v[S][2 * k + 1];
        const unsigned q0 = pack_rn(a, b);
        p0[k] = q0;
        const float ra = a - bitsf(q0 << 16), rb = b - bitsf(q0 & 0xffff0000u);
        const unsigned q1 = pack_rn(ra, rb);
        p1[k] = q1;
        const float sa = ra - bitsf(q1 << 16), sb = rb - bitsf(q1 & 0xffff0000u);
        p2[k] = pack_rn(sa, sb);
    }
    __device__ __forceinline__ void write_plane(int st, int pl) {
        extern __shared__ __attribute__((aligned(16))) char smem_c[];
        *reinterpret_cast<u32x4*>(smem_c + st + lds + pl * X_PLANE) = pl == 0 ? p0 : pl == 1 ? p1 : p2;
    }
    __device__ __forceinline__ void write(int st) { write_plane(st, 0); write_plane(st, 1); write_plane(st, 2); }
};

// WM = 32-row MFMA tiles per wave: 2 = the 128-row tile, 1 = a 64-row tile (half the MFMAs per k-tile beside the same B staging) for
// skinny launches whose 128-row tiling would leave the chip at one workgroup per CU (the mu / value heads).
template <bool AKC, bool BKC, int WM>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) gemm_x3_kernel(const GemmArgs g) {
    constexpr int BMx = 64 * WM;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const WgMap wg = map_workgroup(g.tiles_m * g.tiles_n, g.batch, g.splitk);
    const int id = wg.id;
    const int tm = id / g.tiles_n, tn = id - tm * g.tiles_n;
    const int m0 = tm * BMx, n0 = tn * BN;
    const int bz = wg.bz, sp = wg.sp;
    const int kbeg = sp * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int klen = kend - kbeg;
    const int nkt = (klen + XK - 1) / XK;
    const int hi = klen - (nkt - 1) * XK;                       // valid k positions of the last tile (1 .. 16)

    long long dbg_c0 = 0, dbg_w0 = 0, dbg_c1 = 0, dbg_w1 = 0;
    if (g.dbg) { dbg_c0 = clock64(); dbg_w0 = wall_clock64(); }

    // buffer resources with the TRUE extent from this workgroup's origin: what the range check catches reads as zero (no memory access)
    const int extA = min(BMx, g.M - m0), extB = min(BN, g.N - n0);
    const int k4rem = ((g.K + 3) & ~3) - kbeg;                   // readable k positions of a reduction-contiguous row from kbeg
    const float* Ab = g.A + bz * g.sA + (AKC ? (long long)m0 * g.lda + kbeg : (long long)kbeg * g.lda + m0);
    const float* Bb = g.B + bz * g.sB + (BKC ? (long long)n0 * g.ldb + kbeg : (long long)kbeg * g.ldb + n0);
    const unsigned recA = (unsigned)(AKC ? ((extA - 1) * g.lda + k4rem) : ((klen - 1) * g.lda + extA)) * 4u;
    const unsigned recB = (unsigned)(BKC ? ((extB - 1) * g.ldb + k4rem) : ((klen - 1) * g.ldb + extB)) * 4u;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Ab), 0, klen > 0 ? recA : 0u, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Bb), 0, klen > 0 ? recB : 0u, RSRC_FLAGS);
    const int kstepA = (AKC ? 4 : g.lda * 4) * XK, kstepB = (BKC ? 4 : g.ldb * 4) * XK;      // bytes per k-tile

    StagerX<AKC> sa;
    StagerX<BKC> sb;
    sa.init(tid, g.lda, 0);
    sb.init(tid, g.ldb, X_IMG);

    // fragment read addresses: lane (l31, half) reads out (wm|wn) * 64 + {0, 32} + l31, k-chunk = half, plane p at + p * X_PLANE
    const int frA0 = (half * X_CSTRIDE + slot_of(wm * 32 * WM + l31)) * 16;
    const int frA1 = (half * X_CSTRIDE + slot_of(wm * 32 * WM + 32 + l31)) * 16;      // WM == 2 only
    const int frB0 = X_IMG + (half * X_CSTRIDE + slot_of(wn * 64 + l31)) * 16;
    const int frB1 = X_IMG + (half * X_CSTRIDE + slot_of(wn * 64 + 32 + l31)) * 16;

    f32x16 acc[WM][2];
    {
        float b0 = 0.f, b1 = 0.f;
        if (g.epi == 0 && g.bias) {
            const float* bias = g.bias + bz * g.sBias;
            const int c0 = n0 + wn * 64 + l31;
            if (c0 < g.N) b0 = bias[c0];
            if (c0 + 32 < g.N) b1 = bias[c0 + 32];
        }
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][0][r] = b0; acc[i][1][r] = b1; }
    }
    float rs_acc = 0.f;
    const bool do_rs = !AKC && g.rowsum != nullptr && tn == 0;      // workgroup-uniform: the stager of A sums its k rows (bias gradient)

    // Fragment registers: plane 0 in two sets (tile parity), planes 1 and 2 in ONE set that is refilled as soon as the tile's last
    // MFMA reading it has issued.  Term order (A plane, B plane): (2,0) (0,2) (1,1) (1,0) (0,1) (0,0)  =>  A2 is dead after MFMA 3,
    // B2 after 7, A1 after 15, B1 after 19.
    bf16x8 fa0[2][WM], fb0[2][2], fa1[WM], fb1[2], fa2[WM], fb2[2];
    // fragment read unit u (0..11) of the stage at byte offset st (plane 0 into set S), in the order the slots allow
    auto frag_unit = [&](auto set_tag, int u, int st) {
        constexpr int S = decltype(set_tag)::value;
        extern __shared__ __attribute__((aligned(16))) char smem_c[];
        auto rd = [&](int addr) { return *reinterpret_cast<const bf16x8*>(smem_c + st + addr); };
        switch (u) {
            case 0: fa2[0] = rd(frA0 + 2 * X_PLANE); break;
            case 1: if constexpr (WM == 2) fa2[1] = rd(frA1 + 2 * X_PLANE); break;
            case 2: fb0[S][0] = rd(frB0); break;
            case 3: fb0[S][1] = rd(frB1); break;
            case 4: fb2[0] = rd(frB0 + 2 * X_PLANE); break;
            case 5: fb2[1] = rd(frB1 + 2 * X_PLANE); break;
            case 6: fa0[S][0] = rd(frA0); break;
            case 7: if constexpr (WM == 2) fa0[S][1] = rd(frA1); break;
            case 8: fa1[0] = rd(frA0 + X_PLANE); break;
            case 9: if constexpr (WM == 2) fa1[1] = rd(frA1 + X_PLANE); break;
            case 10: fb1[0] = rd(frB0 + X_PLANE); break;
            default: fb1[1] = rd(frB1 + X_PLANE); break;
        }
    };

    // One k-tile = 24 MFMAs (6 plane pairs x 4 accumulator tiles; an accumulator is reused every 4th MFMA), one unit of side work
    // after each:
    //   slots 0-7    split of tile t+1: A pairs 0-3, B pairs 0-3 (its loads were issued TWO tiles ago: a tile is only ~770 MFMA
    //                cycles per wave, far less than the memory latency)
    //   slots 4-6, 8-10   A's / B's three ds_write_b128, one per slot
    //   slot 11      global loads of tile t+3 into the register set tile t+1 just left
    //   slot 13      the ONE barrier of the tile (its lgkmcnt wait falls three MFMAs after the last store)
    //   slots 14-17  next tile's fragment reads A2 B0' B2 A0' (two per slot), slot 20: A1 (dead after MFMA 15), slot 22: B1 (after 19)
    // MODE 0 steady, 1 = stages the LAST tile (k tail zeroed, no further loads), 2 = last tile (compute only).
    auto tile = [&](auto mode_tag, auto stage_tag, int t) {
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr int S = decltype(stage_tag)::value;
        constexpr int OTH = (1 - S) * X_STAGE;
        using SetO = std::integral_constant<int, 1 - S>;
        constexpr int O = 1 - S;                                  // register set / stage of tile t+1 (and t+3)
        if constexpr (MODE != 2) {
            sa.template mask<O, MODE == 1>(hi);
            sb.template mask<O, MODE == 1>(hi);
            if constexpr (!AKC) {
                if (do_rs) rs_acc += sa.template sum8<O>();
            }
        }
        // the side-work schedule is written in 24 SLOTS (slot s belongs to term s / 4); with WM == 2 every slot follows its own MFMA,
        // with WM == 1 the tile has 12 MFMAs and each is followed by two slots
        constexpr int SPM = 2 / WM;
#pragma unroll
        for (int q = 0; q < 12 * WM; ++q) {
            {
                const int term = q / (2 * WM), i = (q >> 1) % WM, j = q & 1;
                const bf16x8 a = term == 0 ? fa2[i] : (term == 2 || term == 3) ? fa1[i] : fa0[S][i];
                const bf16x8 b = term == 1 ? fb2[j] : (term == 2 || term == 4) ? fb1[j] : fb0[S][j];
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int p = q * SPM; p < (q + 1) * SPM; ++p)
            if constexpr (MODE != 2) {
                if (p < 4) sa.template split_pair<O>(p);
                else if (p < 8) sb.template split_pair<O>(p - 4);
                if (p >= 4 && p < 7) sa.write_plane(OTH, p - 4);          // one 16-byte store per gap: the store path takes ~13 cycles each
                if (p >= 8 && p < 11) sb.write_plane(OTH, p - 8);
                if constexpr (MODE == 0) {
                    if (p == 11 && t + 3 < nkt) {
                        sa.template load<O>(rsA, (t + 3) * kstepA); sb.template load<O>(rsB, (t + 3) * kstepB);
                    }
                }
                if (p == X3_BARRIER_GAP) {                                // a few MFMAs after the last store: its lgkmcnt wait is short
                    __builtin_amdgcn_sched_barrier(0);
                    __syncthreads();
                }
                {
                    // 12 fragment reads in the gaps after the barrier; A1 may be refilled after MFMA 15, B1 after MFMA 19
                    constexpr int R0 = X3_BARRIER_GAP + 1;
                    if (p == R0) { frag_unit(SetO{}, 0, OTH); frag_unit(SetO{}, 1, OTH); }
                    else if (p == R0 + 1) { frag_unit(SetO{}, 2, OTH); frag_unit(SetO{}, 3, OTH); }
                    else if (p == R0 + 2) { frag_unit(SetO{}, 4, OTH); frag_unit(SetO{}, 5, OTH); }
                    else if (p == R0 + 3) { frag_unit(SetO{}, 6, OTH); frag_unit(SetO{}, 7, OTH); }
                    if (p == 20) { frag_unit(SetO{}, 8, OTH); frag_unit(SetO{}, 9, OTH); }
                    if (p == 22) { frag_unit(SetO{}, 10, OTH); frag_unit(SetO{}, 11, OTH); }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;

    if (nkt > 0) {
        sa.template load<0>(rsA, 0); sb.template load<0>(rsB, 0);
        if (nkt > 1) { sa.template load<1>(rsA, kstepA); sb.template load<1>(rsB, kstepB); }
        if (nkt == 1) { sa.template mask<0, true>(hi); sb.template mask<0, true>(hi); }
        if constexpr (!AKC) {
            if (do_rs) rs_acc += sa.template sum8<0>();
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { sa.template split_pair<0>(k); sb.template split_pair<0>(k); }
        sa.write(0); sb.write(0);
    }
    __syncthreads();
    if (nkt > 2) { sa.template load<0>(rsA, 2 * kstepA); sb.template load<0>(rsB, 2 * kstepB); }
    if (nkt > 0) {
#pragma unroll
        for (int u = 0; u < 12; ++u) frag_unit(I0{}, u, 0);
    }
    {
        int t = 0;
        for (; t + 3 < nkt; t += 2) { tile(I0{}, I0{}, t); tile(I0{}, I1{}, t + 1); }
        if (t + 2 < nkt) {
            tile(I0{}, I0{}, t);
            tile(I1{}, I1{}, t + 1);
            tile(I2{}, I0{}, t + 2);
        } else if (t + 2 == nkt) {
            tile(I1{}, I0{}, t);
            tile(I2{}, I1{}, t + 1);
        } else if (t + 1 == nkt) {
            tile(I2{}, I0{}, t);
        }
    }
    __syncthreads();                                              // the epilogue (and the row-sum exchange) reuse the staging buffers
    if constexpr (!AKC) {
        if (do_rs) {
            extern __shared__ __attribute__((aligned(16))) float smem[];
            smem[tid] = rs_acc;                                   // thread (kch = tid >> 7, out = tid & 127) summed its 8 k rows of every tile
            __syncthreads();
            if (tid < BMx && m0 + tid < g.M) g.rowsum[bz * g.sRowsum + sp * g.sSplit + m0 + tid] = smem[tid] + smem[tid + 128];
            __syncthreads();
        }
    }
    if (g.dbg) { dbg_c1 = clock64(); dbg_w1 = wall_clock64(); }
    struct DbgStamp {
        const GemmArgs& g; long long c0, w0, c1, w1;
        __device__ ~DbgStamp() {
            if (g.dbg && threadIdx.x == 0) {
                long long* o = g.dbg + 8 * (blockIdx.y * gridDim.x + blockIdx.x);
                o[0] = c0; o[1] = w0; o[2] = c1; o[3] = w1; o[4] = clock64(); o[5] = wall_clock64();
                o[7] = ((long long)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
            }
        }
    } dbg_stamp{g, dbg_c0, dbg_w0, dbg_c1, dbg_w1};
    gemm_epilogue(g, acc, tid, m0, n0, bz, sp, wm, wn, half, l31);
}

// ---- deterministic reduction of split-K slabs (and of column-sum partials) ---------------------
__global__ void __launch_bounds__(256) reduce_slabs_kernel(const float* __restrict__ slabs, int nslab, long long slab_stride,
                                                          long long count, float* __restrict__ out, float scale) {
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < count; i += (long long)gridDim.x * blockDim.x * 4) {
        if (i + 3 < count) {
            // four independent chains (slabs k, k+1, k+2, k+3 of every group of four), combined in a fixed order: the loads of a group are in
            // flight together (a 64-row reduce of a few KB was one dependent L2 round trip per row: 16 us)
            float4 s = *reinterpret_cast<const float4*>(slabs + i);
            float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1, s3 = s1;
            int k = 1;
            for (; k + 15 < nslab; k += 16) {                  // [r6] sixteen slabs in flight, added in the same order (reduce_grads_kernel, b16_ops.hip)
                float4 v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const float4*>(slabs + (k + u) * slab_stride + i);
#pragma unroll
                for (int u = 0; u < 16; u += 4) {
                    s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w;
                    s1.x += v[u + 1].x; s1.y += v[u + 1].y; s1.z += v[u + 1].z; s1.w += v[u + 1].w;
                    s2.x += v[u + 2].x; s2.y += v[u + 2].y; s2.z += v[u + 2].z; s2.w += v[u + 2].w;
                    s3.x += v[u + 3].x; s3.y += v[u + 3].y; s3.z += v[u + 3].z; s3.w += v[u + 3].w;
                }
            }
            for (; k + 3 < nslab; k += 4) {
                const float4 v0 = *reinterpret_cast<const float4*>(slabs + k * slab_stride + i);
                const float4 v1 = *reinterpret_cast<const float4*>(slabs + (k + 1) * slab_stride + i);
                const float4 v2 = *reinterpret_cast<const float4*>(slabs + (k + 2) * slab_stride + i);
                const float4 v3 = *reinterpret_cast<const float4*>(slabs + (k + 3) * slab_stride + i);
                s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
                s1.x += v1.x; s1.y += v1.y; s1.z += v1.z; s1.w += v1.w;
                s2.x += v2.x; s2.y += v2.y; s2.z += v2.z; s2.w += v2.w;
                s3.x += v3.x; s3.y += v3.y; s3.z += v3.z; s3.w += v3.w;
            }
            const int rem = nslab - k;                          // the last one to three slabs onto chain 0 in order, loads issued together
            if (rem == 3) {
                const float4 t0 = *reinterpret_cast<const float4*>(slabs + k * slab_stride + i);
                const float4 t1 = *reinterpret_cast<const float4*>(slabs + (k + 1) * slab_stride + i);
                const float4 t2 = *reinterpret_cast<const float4*>(slabs + (k + 2) * slab_stride + i);
                s.x += t0.x; s.y += t0.y; s.z += t0.z; s.w += t0.w;
                s.x += t1.x; s.y += t1.y; s.z += t1.z; s.w += t1.w;
                s.x += t2.x; s.y += t2.y; s.z += t2.z; s.w += t2.w;
            } else if (rem == 2) {
                const float4 t0 = *reinterpret_cast<const float4*>(slabs + k * slab_stride + i);
                const float4 t1 = *reinterpret_cast<const float4*>(slabs + (k + 1) * slab_stride + i);
                s.x += t0.x; s.y += t0.y; s.z += t0.z; s.w += t0.w;
                s.x += t1.x; s.y += t1.y; s.z += t1.z; s.w += t1.w;
            } else if (rem == 1) {
                const float4 t0 = *reinterpret_cast<const float4*>(slabs + k * slab_stride + i);
                s.x += t0.x; s.y += t0.y; s.z += t0.z; s.w += t0.w;
            }
            s.x = (s.x + s1.x) + (s2.x + s3.x); s.y = (s.y + s1.y) + (s2.y + s3.y); s.z = (s.z + s1.z) + (s2.z + s3.z); s.w = (s.w + s1.w) + (s2.w + s3.w);
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

namespace {
// Diagnostics state: THREAD-LOCAL (round-3 verdict, hygiene): a tool thread that arms the clock stamps or an occupancy knob changes the
// launches it issues itself, never those of another host thread driving its own stream through the library.
thread_local long long* g_dbg = nullptr;       // tools/gemm_bench --clocks
thread_local int g_last_tile = 0;               // tile rows of the calling thread's last pulse_gemm_f32 launch (pulse_gemm_last_tile: bench.py's per-kernel roofline)
thread_local int g_opt[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // [1] extra LDS bytes per workgroup, [2] no 64-row tile, [3] bf16-storage tile choice (see pulse_hip.h)
}

namespace pulse { int gemm_option(int key) { return key >= 0 && key < 16 ? g_opt[key] : 0; } long long* gemm_debug_buffer() { return g_dbg; } }    // read by gemm_x3p.hip (common.h)

namespace {
// Which tiling serves an x3 launch.  Cost model in units of (one 128 x 128 output tile) x (k per split), per CU: the narrow kernel keeps two
// workgroups per CU (a round of 512 costs 2; a lone workgroup per CU 1.5), the wide kernel one workgroup of four tiles' area
// per CU and round at 1.1-1.2 x the narrow kernel's rate on long reductions (about 3.3 units per wide round, more on short ones; calibrated on
// profiles/r05_gemm_x3_wide_ab.txt and on every launch of a cfg2 / cfg3 epoch, profiles/r05_gemm_shapes_cfg{2,3}.txt).  Option 4 (pulse_gemm_set_option) / PULSE_X3_WIDE: 0 automatic, 1 never,
// 2 whenever the output has more than 128 rows and columns (tests).
constexpr int WIDE_TILE = 256;
bool skinny_env_off() { static const bool off = [] { const char* e = getenv("PULSE_X3_SKINNY"); return e && e[0] == '0'; }(); return off; }   // A/B switch, read once
bool g_skinny_unavailable = false;    // the device refused the skinny-N kernel's 144 KB of LDS once
bool g_wide_unavailable = false;      // the device refused the wide tile's LDS request once: never asked again
// 0 = the launcher's cost model, 1 = never the 256 x 256 tile, 2 = whenever the output has more than 128 rows and columns
int x3_mode() {
    static const int env = [] { const char* e = getenv("PULSE_X3_WIDE"); return e ? atoi(e) : -1; }();
    int mode = g_opt[4];
    if (mode == 0 && env >= 0) mode = env == 0 ? 1 : env == 1 ? 0 : env;          // PULSE_X3_WIDE=0 off, 1 automatic, 2 always
    return g_wide_unavailable ? 1 : mode;
}
bool x3_wide_tile(const GemmArgs& g, int lda, int ldb, bool akc, bool bkc) {
    const int mode = x3_mode();
    if (mode == 1 || g.M <= 128 || g.N <= 128) return false;
    // per-workgroup buffer offsets are 32-bit: 256 rows of a reduction-contiguous operand, kchunk rows of a [red][out] operand
    if ((long long)lda * (akc ? 257 : g.kchunk + 1) >= (1LL << 28) || (long long)ldb * (bkc ? 257 : g.kchunk + 1) >= (1LL << 28) ||
        (long long)g.ldc * 257 >= (1LL << 28) || (long long)g.ldaux * 257 >= (1LL << 28))
        return false;
    if (mode == 2) return true;
    const long long z = (long long)g.batch * g.splitk;
    const long long nt = (long long)((g.M + 127) / 128) * ((g.N + 127) / 128) * z;
    const long long wt = (long long)((g.M + 255) / 256) * ((g.N + 255) / 256) * z;
    const long long rem = nt % 512;
    const double cost_narrow = 2.0 * (double)(nt / 512) + (rem == 0 ? 0.0 : rem <= 256 ? 1.5 : 2.0);
    // a round of wide workgroups against a round of 512 narrow ones (= 2 units), from the per-round times of both tilings over the reduction length
    // each workgroup walks (us: narrow 0.0924 k + 5, wide 0.1526 k + f): the wide tile's prologue and epilogue are exposed (one workgroup per CU),
    // f = 8 for plain / ReLU / mask / multiply epilogues, 30 for the SiLU forms that write two outputs and evaluate an exp and a division per
    // element, 45 for the SiLU-derivative epilogue (fits of profiles/r05_gemm_shapes_cfg{2,3}.txt)
    const double kk = (double)g.kchunk < (double)g.K ? (double)g.kchunk : (double)g.K;
    const double fw = g.epi == 2 ? 45.0 : (g.epi == 0 && g.act >= 2) ? 30.0 : 8.0;
    const double wide_round = 2.0 * (0.1526 * kk + fw) / (0.0924 * kk + 5.0);
    const double cost_wide = wide_round * (double)((wt + 255) / 256);
    return cost_wide < cost_narrow;
}
}  // namespace

extern "C" {

int pulse_sizeof_gemm_desc(void) { return (int)sizeof(pulse_gemm_desc); }

int pulse_gemm_set_debug_buffer(long long* device_buffer) { g_dbg = device_buffer; return PULSE_OK; }

int pulse_gemm_last_tile(void) { return g_last_tile; }
int pulse_gemm_x3_mode(void) { return x3_mode(); }

int pulse_gemm_set_option(int key, int value) {
    PULSE_REQUIRE(key >= 0 && key < 16, "pulse_gemm_set_option: bad key");
    g_opt[key] = value;
    return PULSE_OK;
}

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
    PULSE_REQUIRE(d->epilogue >= 0 && d->epilogue <= 3 && d->activation >= 0 && d->activation <= 3, "pulse_gemm_f32: bad epilogue / activation");
    PULSE_REQUIRE(d->activation != PULSE_ACT_SILU_D || (d->C2 != nullptr && d->ldc2 >= d->N), "pulse_gemm_f32: ACT_SILU_D stores the derivative in C2");
    PULSE_REQUIRE(d->epilogue == 0 || d->activation == 0, "pulse_gemm_f32: a gradient epilogue takes no activation");
    PULSE_REQUIRE(d->epilogue == 0 || d->aux != nullptr || (d->epilogue == PULSE_EPI_RELU_GRAD && d->relu_mask != nullptr),
                  "pulse_gemm_f32: gradient epilogue needs aux (or, for relu-grad, relu_mask)");
    const bool mask_on = d->relu_mask != nullptr && ((d->epilogue == PULSE_EPI_RELU_GRAD && d->aux == nullptr) ||
                                                     (d->epilogue == PULSE_EPI_BIAS_ACT && d->activation == PULSE_ACT_RELU));
    PULSE_REQUIRE(!mask_on || (d->ld_mask >= (d->N + 3) / 4 && d->split_k == 1), "pulse_gemm_f32: relu_mask needs ld_mask >= roundup4(N) / 4 and no split-K");
    PULSE_REQUIRE(d->rowsum == nullptr || (!akc && !bkc), "pulse_gemm_f32: rowsum needs the (OUT, OUT) layouts (dW pass)");
    PULSE_REQUIRE(d->split_k == 1 || (d->epilogue == 0 && d->activation == 0 && d->bias == nullptr),
                  "pulse_gemm_f32: split-K slabs carry no epilogue");

    GemmArgs g;
    g.A = d->A; g.B = d->B; g.C = d->C; g.C2 = d->C2; g.bias = d->bias; g.aux = d->aux;
    g.M = d->M; g.N = d->N; g.K = d->K;
    g.lda = d->lda; g.ldb = d->ldb; g.ldc = d->ldc; g.ldc2 = d->ldc2; g.ldaux = d->ldaux;
    g.sA = d->stride_a; g.sB = d->stride_b; g.sC = d->stride_c; g.sC2 = d->stride_c2; g.sBias = d->stride_bias; g.sAux = d->stride_aux;
    g.batch = d->batch; g.splitk = d->split_k;
    const int bk = d->compute_type == PULSE_GEMM_COMPUTE_BF16 ? BK16 : d->compute_type == PULSE_GEMM_COMPUTE_F32X3 ? XK : BK;
    int kchunk = (d->K + d->split_k - 1) / d->split_k;
    kchunk = ((kchunk + bk - 1) / bk) * bk;
    g.kchunk = kchunk > 0 ? kchunk : bk;
    g.sSplit = d->split_stride;
    g.act = d->activation; g.epi = d->epilogue;
    g.rowsum = d->rowsum; g.sRowsum = d->stride_rowsum;
    g.mask = mask_on ? d->relu_mask : nullptr; g.ldmask = d->ld_mask; g.sMask = d->stride_mask;
    g.tiles_m = (d->M + BM - 1) / BM; g.tiles_n = (d->N + BN - 1) / BN;
    // x3 only: a 64-row tile for skinny outputs (one column tile: the mu / value heads) whose 128-row tiling leaves the chip at one
    // workgroup per CU or less.  Measured: heads at M = 16384 36.2 -> 32.8 us, at M = 4096 28.5 -> 21.6 us; full-width outputs at the same
    // workgroup count get SLOWER with the half tile (twice the B staging per MFMA: rollout layer 2 52.8 -> 58.3 us), so they keep 128 rows.
    const bool half_tile = d->compute_type == PULSE_GEMM_COMPUTE_F32X3 && d->M >= 256 && g.tiles_n == 1 &&
                           (long long)g.tiles_m * d->batch * d->split_k < 384 && g_opt[2] == 0;
    if (half_tile) g.tiles_m = (d->M + 63) / 64;
    g.dbg = g_dbg;
    PULSE_REQUIRE(d->compute_type == PULSE_GEMM_COMPUTE_F32 || d->compute_type == PULSE_GEMM_COMPUTE_BF16 ||
                  d->compute_type == PULSE_GEMM_COMPUTE_F32X3, "pulse_gemm_f32: bad compute_type");
    const bool bf = d->compute_type == PULSE_GEMM_COMPUTE_BF16, x3 = d->compute_type == PULSE_GEMM_COMPUTE_F32X3;
    g.round_bf16 = bf && d->round_output_bf16 ? 1 : 0;
    // per-workgroup buffer offsets are 32-bit: tile-relative (128 rows) for reduction-contiguous operands, split-relative
    // (kchunk rows) for [red][out] operands
    PULSE_REQUIRE((long long)d->lda * (akc ? 129 : g.kchunk + 1) < (1LL << 28) && (long long)d->ldb * (bkc ? 129 : g.kchunk + 1) < (1LL << 28) &&
                  (long long)d->ldc * 129 < (1LL << 28) && (long long)d->ldaux * 129 < (1LL << 28),
                  "pulse_gemm_f32: pitch too large for 32-bit tile-relative offsets");
    auto al16 = [](const void* p, long long ld, long long st) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && (ld % 4) == 0 && (st % 4) == 0; };
    g.vec_epi = al16(d->C, d->ldc, d->stride_c) && (d->split_stride % 4) == 0 && (!d->aux || al16(d->aux, d->ldaux, d->stride_aux)) &&
                (!d->C2 || al16(d->C2, d->ldc2, d->stride_c2)) && (!d->bias || al16(d->bias, 4, d->stride_bias));
    PULSE_REQUIRE(!mask_on || g.vec_epi, "pulse_gemm_f32: relu_mask needs 16-byte aligned C / pitches (a lane owns four columns of a mask word)");
    const size_t lds = (size_t)(x3 ? X_LDS : LDS_BYTES) + (size_t)g_opt[1];   // 65,536 / 66,048 B -> two workgroups per CU
    const dim3 grid((unsigned)(g.tiles_m * g.tiles_n), (unsigned)(d->batch * d->split_k));
    // The 64.5 KiB dynamic-LDS opt-in is a per-function attribute: set it ONCE per instantiation (calling
    // hipFuncSetAttribute on every launch serialises the host against the stream).
    static size_t attr_done[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    hipError_t e = hipSuccess;
#define LAUNCH(IDX, AK, BK_)                                                                                       \
    if (attr_done[IDX] != lds) {                                                                                      \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_kernel<AK, BK_>),                          \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                            \
        if (e != hipSuccess) return fail(PULSE_ERR_LAUNCH, "pulse_gemm_f32: LDS attribute: %s", hipGetErrorString(e)); \
        attr_done[IDX] = lds;                                                                                     \
    }                                                                                                             \
    hipLaunchKernelGGL((gemm_f32_kernel<AK, BK_>), grid, dim3(256), lds, as_stream(s), g)
#define LAUNCH16(IDX, AK, BK_)                                                                                     \
    if (attr_done[IDX] != lds) {                                                                                  \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<AK, BK_>),                         \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                            \
        if (e != hipSuccess) return fail(PULSE_ERR_LAUNCH, "pulse_gemm_f32: LDS attribute: %s", hipGetErrorString(e)); \
        attr_done[IDX] = lds;                                                                                     \
    }                                                                                                             \
    hipLaunchKernelGGL((gemm_bf16_kernel<AK, BK_>), grid, dim3(256), lds, as_stream(s), g)
#define LAUNCHX(IDX, AK, BK_, WM_)                                                                                 \
    if (attr_done[IDX] != lds) {                                                                                  \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x3_kernel<AK, BK_, WM_>),                      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                            \
        if (e != hipSuccess) return fail(PULSE_ERR_LAUNCH, "pulse_gemm_f32: LDS attribute: %s", hipGetErrorString(e)); \
        attr_done[IDX] = lds;                                                                                     \
    }                                                                                                             \
    hipLaunchKernelGGL((gemm_x3_kernel<AK, BK_, WM_>), grid, dim3(256), lds, as_stream(s), g)
    g_last_tile = half_tile ? 64 : 128;
    // skinny outputs over a long M (the mu / value heads, the latent-width layers): gemm_x3s.hip -- a workgroup owns 128 rows x all N <= 96
    // columns, A goes global -> registers -> fragments, B is split once per workgroup and 128-deep k phase.  Bit-identical to the other tilings.
    // Taken when the launch is ONE round of 128-row workgroups that fills most of the chip (192 .. 256 of them; measured, tools/bench_gemm_x3_skinny.py:
    // 16384 x 69 x 512 x 2 nets 34.2 -> 29.1 us, but two rounds (32768 rows) 47.7 -> 57.9 and half a round no gain); gemm option 6 = 1: never.
    if (x3 && akc && d->N <= 96 && d->epilogue == PULSE_EPI_BIAS_ACT && d->activation <= PULSE_ACT_RELU && d->C2 == nullptr && d->split_k == 1 &&
        d->rowsum == nullptr && !mask_on && g_dbg == nullptr && g_opt[6] == 0 && !g_skinny_unavailable && !skinny_env_off() &&
        (long long)((d->M + 127) / 128) * d->batch >= 192 && (long long)((d->M + 127) / 128) * d->batch <= 256 && (long long)d->lda * 129 < (1LL << 28) &&
        (bkc ? (long long)d->ldb * 97 : (long long)d->ldb * (d->K + 1)) < (1LL << 28)) {
        const int rc = launch_gemm_x3s(g, bkc, as_stream(s));
        if (rc != kWideTileUnavailable) { g_last_tile = 96; return rc; }
        g_skinny_unavailable = true;
    }
    if (x3 && !half_tile && x3_wide_tile(g, d->lda, d->ldb, akc, bkc)) {
        // 256 x 256 tile (gemm_x3w.hip): half the split / staging work per MFMA; taken when its one-workgroup-per-CU rounds cost less than the
        // 128 x 128 tiling's (two workgroups per CU) -- see x3_wide_tile.  A device that does not grant its 135 KB of LDS keeps the narrow tile
        // (same bits either way).
        // A narrow column tail that costs the wide tiling a whole extra round of workgroups (N = 3096 = 12 x 256 + 24: 13 column tiles, 832
        // workgroups = 4 rounds at M = 16384, where 12 x 64 = 768 is exactly 3) goes to the 128 x 128 tiling as a launch of its own: the two
        // tilings are bit-identical, so the split is invisible in the results.  (Epilogue-carrying launches only: a split-K / row-sum launch
        // writes slabs whose tiling the planner already sized.)
        const int ntail = d->N % WIDE_TILE;
        if (ntail > 0 && ntail <= 64 && d->N > WIDE_TILE && d->split_k == 1 && d->rowsum == nullptr && g_opt[5] == 0) {
            const long long tm = (d->M + WIDE_TILE - 1) / WIDE_TILE, z = d->batch;
            const long long r_all = (tm * ((d->N + WIDE_TILE - 1) / WIDE_TILE) * z + 255) / 256, r_main = (tm * (d->N / WIDE_TILE) * z + 255) / 256;
            if (r_main < r_all) {
                const int n0 = d->N - ntail;
                pulse_gemm_desc m = *d, t = *d;
                m.N = n0;
                t.N = ntail;
                t.B = bkc ? d->B + (long long)n0 * d->ldb : d->B + n0;
                t.C = d->C + n0;
                if (d->C2) t.C2 = d->C2 + n0;
                if (d->bias) t.bias = d->bias + n0;
                if (d->aux) t.aux = d->aux + n0;
                if (d->relu_mask) t.relu_mask = d->relu_mask + n0 / 4;
                const int rc_main = pulse_gemm_f32(&m, s);
                if (rc_main != PULSE_OK) return rc_main;
                const int rc_tail = pulse_gemm_f32(&t, s);
                g_last_tile = 256;                                   // (diagnostics: the launch's time is the wide kernel's)
                return rc_tail;
            }
        }
        const int rc = launch_gemm_x3w(g, akc, bkc, as_stream(s));
        if (rc != kWideTileUnavailable) { g_last_tile = 256; return rc; }
        g_wide_unavailable = true;
    }
    if (x3 && half_tile) {
        if (akc && bkc) { LAUNCHX(9, true, true, 1); }
        else if (akc && !bkc) { LAUNCHX(10, true, false, 1); }
        else { LAUNCHX(11, false, false, 1); }
    } else if (x3) {
        if (akc && bkc) { LAUNCHX(6, true, true, 2); }
        else if (akc && !bkc) { LAUNCHX(7, true, false, 2); }
        else { LAUNCHX(8, false, false, 2); }
    } else if (bf) {
        if (akc && bkc) { LAUNCH16(3, true, true); }
        else if (akc && !bkc) { LAUNCH16(4, true, false); }
        else { LAUNCH16(5, false, false); }
    } else if (akc && bkc) { LAUNCH(0, true, true); }
    else if (akc && !bkc) { LAUNCH(1, true, false); }
    else { LAUNCH(2, false, false); }
#undef LAUNCH
#undef LAUNCH16
#undef LAUNCHX
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
