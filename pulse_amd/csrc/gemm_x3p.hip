// fp32-grade GEMM on the bf16 matrix pipe over operands kept PRE-SPLIT in HBM ("x3p": x3, planar).
//
// Same arithmetic as gemm_x3_kernel (gemm_f32.hip): every fp32 operand is the exact sum of three bf16 numbers (a1 = bf16(a),
// a2 = bf16(a - a1), a3 = bf16(a - a1 - a2), round to nearest even), and C = sum_k a b is the six bf16 MFMAs per 16-deep k step whose
// plane indices satisfy i + j <= 2, accumulated in fp32.  It replaces the same reference call sites (every nn.Linear forward / backward
// of phc/learning/network_builder.py:105-124,245-261, amp_network_builder.py:127-148,206-211, amp_network_z_builder.py:469-580).
//
// What is different (round-2 verdict, weak #2: the x3 kernel re-split every operand element in every tile that consumed it --
// 3.7 VALU and a quarter of an LDS store per MFMA -- and sat at 0.42 of its pipe):
//   * the three planes live in HBM.  Whoever PRODUCES a matrix writes them once: the optimiser step for the weights (and their
//     transposes), the producing GEMM's epilogue for the activations / gradients, pulse_split_planes for everything else;
//   * the main loop has no VALU staging work and no ds_write at all: tiles arrive by LDS-DMA (buffer_load_dwordx4 ... lds), fragments
//     are one ds_read_b128 each (reduction-contiguous operands) or two ds_read_b64_tr_b16 each ([red][out] operands: the dW pass reads
//     activations and gradients in their natural row-major planes through the transposing LDS read -- no transposed copies in HBM);
//   * tile 256 x 128 x 32 (8 waves = 4 x 2, each 64 x 64), so an operand byte moved into LDS feeds 1.33x the MFMAs of the 128 x 128
//     tile, and four lanes fetch the 64 contiguous bytes of a row's k-tile (one quarter of the cache-line requests per byte).
//
// LDS image of a reduction-contiguous operand tile, per plane: [row][4 slots of 16 B] = [row][32 k], slot s of row r holding
// k-chunk s ^ ((r >> 2) & 3).  The DMA writes 64 consecutive slots per wave instruction (lane L: row L >> 2, slot L & 3), which fixes
// the image to be lane-linear; the XOR on the SOURCE chunk makes the fragment reads (32 consecutive rows, one chunk) hit every bank
// once.  [red][out] operand tile, per plane: [32 k rows][out pieces of 16 B], piece p of row m at slot p ^ ((m & 3) << 2): the four
// rows x four pieces a half-wave's transposing read touches are 16 different bank groups.
// Two stages of (3 A planes + 3 B planes) = 144 KB (256-row tile), one workgroup per CU, two waves per SIMD; one barrier per k-tile.
//
// Single-plane mode (NPL = 1, "b16"): the operands ARE bf16 matrices (bf16 autocast training, BASELINE.json configs[4]; the reference's
// autocast sites are phc/learning/amp_agent.py:671 and common_agent.py:426,461) -- activations and gradients live in HBM as bf16, half
// the bytes of the fp32-storage bf16 kernel in gemm_f32.hip, which is bound by exactly that traffic.  Same pipeline: the three plane
// slots of a stage hold three CONSECUTIVE 32-deep k-tiles of the one plane, a k step is the three diagonal products (12 MFMAs) instead
// of the six cross terms, and k-tiles past the reduction's end are skipped (their slots hold stale bytes nobody multiplies).
#include <cstdlib>
#include <type_traits>
#include "common.h"

namespace pulse {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;

constexpr int PK = 32;                 // k per tile
constexpr int PBN = 128;               // tile columns
constexpr int ROWB = PK * 2;           // bytes of one plane row of a reduction-contiguous tile
constexpr unsigned P_RSRC = 0x00020000u;

struct XpArgs {
    const unsigned short* A; const unsigned short* B;
    long long pa, pb;                  // plane strides (elements)
    int lda, ldb;                      // pitches (elements)
    float* C; float* C2; unsigned short* Cp; const float* bias; const float* aux; const unsigned short* aux16;
    long long pc;                      // plane stride of Cp (elements)
    int ldc, ldc2, ldcp, ldaux;
    int M, N, K;
    long long sA, sB, sC, sC2, sCp, sBias, sAux;   // batch strides (elements of the respective arrays)
    int batch, splitk, kchunk;
    long long sSplit;
    int act, epi;
    int tiles_m, tiles_n;
    float* rowsum; long long sRowsum;
    float* colsum; long long sColsum; int ldcs;      // optional: per-row-tile column sums of the OUTPUT (colsum[bz * sColsum + tm * ldcs + n])
    long long* dbg;                                  // optional per-workgroup wall-clock stamps (tools/gemm_b16_phases.py; pulse_gemm_set_debug_buffer)
    unsigned char* mask8; int ldm8; long long sM8;   // ReLU bit mask, one byte per (row, 8 columns): written by EPI 0 + relu, read by EPI 1 when there is no aux
    int general_rows;                                // gemm option 9 (tests): every epilogue row through the general form
};

template <int WMW>
struct XpGeom {
    static constexpr int BM = 64 * WMW, NW = 2 * WMW, NT = 64 * NW;
    static constexpr int A_PLANE = BM * ROWB, B_PLANE = PBN * ROWB;
    static constexpr int A_IMG = 3 * A_PLANE, B_IMG = 3 * B_PLANE, STAGE = A_IMG + B_IMG;
    static constexpr int CPF = PBN + 4;                          // epilogue transpose pitch (floats)
    static constexpr int EPI_BYTES = BM * CPF * 4;
    static constexpr int TOUCH_LDS = 256 * NW;                  // landing strip of the L2 touch loads (single-plane mode): 256 B per wave
    static constexpr int LDS = (2 * STAGE > EPI_BYTES ? 2 * STAGE : EPI_BYTES) + TOUCH_LDS;
    static constexpr int TOUCH_OFF = LDS - TOUCH_LDS;
};

__device__ __forceinline__ unsigned xp_pack_rn(float lo, float hi) { return split_pack_rn(lo, hi); }
__device__ __forceinline__ float xp_bitsf(unsigned v) { return split_bitsf(v); }
// the three planes' packed dwords of an element pair (same rounding as StagerX::split_pair in gemm_f32.hip; common.h)
__device__ __forceinline__ void xp_split_pair(float a, float b, unsigned& q0, unsigned& q1, unsigned& q2) { split_pair3(a, b, q0, q1, q2); }

__device__ __forceinline__ bf16x8 xp_lds128(int addr) {
    extern __shared__ __attribute__((aligned(16))) char xp_smem[];
    return *reinterpret_cast<const bf16x8*>(xp_smem + addr);
}
// [red][out] fragment: 8 consecutive k of one out = two transposing 8-byte reads (k rows 0-3 and 4-7 of the lane's k-chunk)
__device__ __forceinline__ bf16x8 xp_lds_tr(int addr_lo, int addr_hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char xp_smem[];
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(xp_smem + addr_lo));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(xp_smem + addr_hi));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return __builtin_bit_cast(bf16x8, v);
#else
    // host pass of the single-source compile: the gfx950-only builtin does not exist there, and a kernel template whose instantiation
    // reaches it is silently not emitted (its launch stub goes missing at link time)
    return bf16x8{};
#endif
}

// ---- epilogue shared by the planar kernels: accumulators -> LDS (fp32, pitch CPF) -> rows of 8 consecutive columns per lane -> fp32 C and / or
// the output's own planes (bf16 matrix in single-plane mode), optional per-row-tile column sums.  The caller has drained its DMA and passed a
// barrier: the staging buffers are free.
template <int WMW, int NPL>
__device__ __forceinline__ void xp_epilogue(const XpArgs& g, f32x16 (&acc)[2][2], int tid, int wm, int wn, int half, int l31, int m0, int n0, int bz,
                                            int sp, int tm) {
    using G = XpGeom<WMW>;
    constexpr int BM = G::BM;
    extern __shared__ __attribute__((aligned(16))) char xp_smem[];
    float* sC = reinterpret_cast<float*>(xp_smem);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                sC[(wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * G::CPF + wn * 64 + j * 32 + l31] = acc[i][j][r];
    __syncthreads();
    if (g.dbg && tid == 0) g.dbg[8 * (blockIdx.y * gridDim.x + blockIdx.x) + 6] = wall_clock64();     // (diagnostics: transpose image written)
    {
        float* C = g.C ? g.C + bz * g.sC + sp * g.sSplit : nullptr;
        float* C2 = g.C2 ? g.C2 + bz * g.sC2 : nullptr;
        unsigned short* Cp = g.Cp ? g.Cp + bz * g.sCp : nullptr;
        const float* aux = g.aux ? g.aux + bz * g.sAux : nullptr;
        const unsigned short* aux16 = g.aux16 ? g.aux16 + bz * g.sAux : nullptr;
        unsigned char* mask8 = g.mask8 ? g.mask8 + bz * g.sM8 : nullptr;
        const bool use_mask = g.epi == 1 && aux == nullptr && aux16 == nullptr;
        const int c8 = (tid & 15) * 8;
        const int col = n0 + c8;
        constexpr int RPI = G::NT / 16;                             // rows per iteration
        float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // column sums of what this thread stores (the bias gradient of the producer layer)
        if (col < g.N) {
            const bool full = col + 7 < g.N;
            // one row of 8 columns: image -> rounding / activation / mask -> C, Cp, column sums
            auto do_row = [&](int rl, const f32x4 v0, const f32x4 v1) {
                const int row = m0 + rl;
                float o[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                if constexpr (NPL == 1) {
                    // a bf16 autocast Linear hands bf16 to the next op: the product leaves rounded, the activation / mask acts on that
                    // (split-K slabs are partial sums and stay fp32)
                    if (g.splitk == 1) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) o[k] = (float)(__bf16)o[k];
                    }
                }
                if (g.epi == 0) {
                    if (g.act == 1) {
                        if (mask8) {                                // the sign bits of this thread's eight outputs: one byte nobody else touches
                            unsigned bits = 0;
#pragma unroll
                            for (int k = 0; k < 8; ++k) bits |= (col + k < g.N && o[k] > 0.f ? 1u : 0u) << k;
                            mask8[(long long)row * g.ldm8 + (col >> 3)] = (unsigned char)bits;
                        }
#pragma unroll
                        for (int k = 0; k < 8; ++k) o[k] = fmaxf(o[k], 0.f);
                    } else if (g.act == 2) {
                        if (C2) {
                            float* p2 = C2 + (long long)row * g.ldc2 + col;
                            if (full) { *reinterpret_cast<f32x4*>(p2) = (f32x4){o[0], o[1], o[2], o[3]}; *reinterpret_cast<f32x4*>(p2 + 4) = (f32x4){o[4], o[5], o[6], o[7]}; }
                            else for (int k = 0; k < 8 && col + k < g.N; ++k) p2[k] = o[k];
                        }
#pragma unroll
                        for (int k = 0; k < 8; ++k) o[k] = o[k] / (1.f + __expf(-o[k]));
                    }
                } else if (use_mask) {                              // relu-grad from the forward's sign bits (one byte instead of 16 / 32 of activations)
                    const unsigned bits = mask8[(long long)row * g.ldm8 + (col >> 3)];
#pragma unroll
                    for (int k = 0; k < 8; ++k) o[k] = ((bits >> k) & 1u) ? o[k] : 0.f;
                } else {
                    float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    if (aux16) {                                    // bf16-stored activations (rows hold roundup8(N) columns)
                        const u32x4 t = *reinterpret_cast<const u32x4*>(aux16 + (long long)row * g.ldaux + col);
#pragma unroll
                        for (int k = 0; k < 4; ++k) { a8[2 * k] = xp_bitsf(t[k] << 16); a8[2 * k + 1] = xp_bitsf(t[k] & 0xffff0000u); }
                    } else {
                        const float* pa = aux + (long long)row * g.ldaux + col;
                        if (full) {
                            const f32x4 t0 = *reinterpret_cast<const f32x4*>(pa), t1 = *reinterpret_cast<const f32x4*>(pa + 4);
                            a8[0] = t0.x; a8[1] = t0.y; a8[2] = t0.z; a8[3] = t0.w; a8[4] = t1.x; a8[5] = t1.y; a8[6] = t1.z; a8[7] = t1.w;
                        } else for (int k = 0; k < 8 && col + k < g.N; ++k) a8[k] = pa[k];
                    }
                    if (g.epi == 1) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) o[k] = a8[k] > 0.f ? o[k] : 0.f;
                    } else {
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const float sg = 1.f / (1.f + __expf(-a8[k]));
                            o[k] *= sg * (1.f + a8[k] * (1.f - sg));
                        }
                    }
                }
                if (g.colsum) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) cs[k] += o[k];
                }
                if (C) {
                    float* pc = C + (long long)row * g.ldc + col;
                    if (full) {
                        *reinterpret_cast<f32x4*>(pc) = (f32x4){o[0], o[1], o[2], o[3]};
                        *reinterpret_cast<f32x4*>(pc + 4) = (f32x4){o[4], o[5], o[6], o[7]};
                    } else for (int k = 0; k < 8 && col + k < g.N; ++k) pc[k] = o[k];
                }
                if (Cp) {
                    // the output's own planes: columns past N inside this 8-group are written as zeros (they are k padding of the consumer)
                    u32x4 q0, q1, q2;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float a = col + 2 * k < g.N ? o[2 * k] : 0.f, b = col + 2 * k + 1 < g.N ? o[2 * k + 1] : 0.f;
                        unsigned x0, x1 = 0u, x2 = 0u;
                        if constexpr (NPL == 3) xp_split_pair(a, b, x0, x1, x2);
                        else x0 = xp_pack_rn(a, b);
                        q0[k] = x0; q1[k] = x1; q2[k] = x2;
                    }
                    unsigned short* pp = Cp + (long long)row * g.ldcp + col;
                    *reinterpret_cast<u32x4*>(pp) = q0;
                    if constexpr (NPL == 3) {
                        *reinterpret_cast<u32x4*>(pp + g.pc) = q1;
                        *reinterpret_cast<u32x4*>(pp + 2 * g.pc) = q2;
                    }
                }
            };
            // [r6] The two hot epilogues of the bf16-storage path -- ReLU forward (sign byte + bf16 row) and ReLU gradient from the sign byte -- on
            // whole 8-column groups with the bf16 matrix as the only output.  ReLU and the mask select either keep a value or replace it by +0, so they
            // commute with the rounding: the row is rounded ONCE, by the pack that stores it, and the sign byte / the column sums are read off the
            // packed words.  (The general row rounds, converts back, acts, tests eight column bounds and packs again: ~95 VALU per row -- at 16 rows per
            // thread and two waves per SIMD that is the 8 us the phases tool shows for a 256 x 256 tile: the epilogue was VALU-bound, not store-bound.)
            // Same bits as the general row: tests/test_bf16_gpu.py::test_b16_fast_epilogue_rows_equal_the_general_row.
            const bool fast = NPL == 1 && g.splitk == 1 && Cp != nullptr && C == nullptr && C2 == nullptr && full && g.general_rows == 0 &&
                              ((g.epi == 0 && g.act <= 1) || use_mask);
            auto fast_row = [&](int rl, const f32x4 v0, const f32x4 v1) {
                const int row = m0 + rl;
                float o[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                if (g.epi == 0) {
                    if (g.act == 1) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) o[k] = fmaxf(o[k], 0.f);
                    }
                } else {
                    const unsigned bits = mask8[(long long)row * g.ldm8 + (col >> 3)];
#pragma unroll
                    for (int k = 0; k < 8; ++k) o[k] = ((bits >> k) & 1u) ? o[k] : 0.f;
                }
                u32x4 q;
#pragma unroll
                for (int k = 0; k < 4; ++k) q[k] = xp_pack_rn(o[2 * k], o[2 * k + 1]);
                if (g.epi == 0 && g.act == 1 && mask8) {                 // rounded value > 0  <=>  its bf16 magnitude bits are not all zero (after ReLU nothing is negative)
                    unsigned bits = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) bits |= ((q[k] & 0x7fffu) ? 1u : 0u) << (2 * k) | ((q[k] & 0x7fff0000u) ? 1u : 0u) << (2 * k + 1);
                    mask8[(long long)row * g.ldm8 + (col >> 3)] = (unsigned char)bits;
                }
                if (g.colsum) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) { cs[2 * k] += xp_bitsf(q[k] << 16); cs[2 * k + 1] += xp_bitsf(q[k] & 0xffff0000u); }
                }
                *reinterpret_cast<u32x4*>(Cp + (long long)row * g.ldcp + col) = q;
            };
            constexpr int ITER = BM / RPI;
            if (m0 + BM <= g.M) {
                // full tile in M: every image read of the thread's ITER rows is issued before the first row is processed (a rolled loop was one
                // LDS round trip + one store issue per row, end to end: 4.2 us per 256 x 128 half, profiles/r04_gemm_b16_phases.txt)
                f32x4 va[ITER], vb[ITER];
#pragma unroll
                for (int it = 0; it < ITER; ++it) {
                    const int rl = (tid >> 4) + it * RPI;
                    va[it] = *reinterpret_cast<const f32x4*>(sC + rl * G::CPF + c8);
                    vb[it] = *reinterpret_cast<const f32x4*>(sC + rl * G::CPF + c8 + 4);
                }
                if (fast) {
#pragma unroll
                    for (int it = 0; it < ITER; ++it) fast_row((tid >> 4) + it * RPI, va[it], vb[it]);
                } else {
#pragma unroll
                    for (int it = 0; it < ITER; ++it) do_row((tid >> 4) + it * RPI, va[it], vb[it]);
                }
            } else {
#pragma unroll 2
                for (int rl = tid >> 4; rl < BM; rl += RPI) {
                    if (m0 + rl >= g.M) break;
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(sC + rl * G::CPF + c8), v1 = *reinterpret_cast<const f32x4*>(sC + rl * G::CPF + c8 + 4);
                    if (fast) fast_row(rl, v0, v1);
                    else do_row(rl, v0, v1);
                }
            }
        }
        if (g.dbg && tid == 0) g.dbg[8 * (blockIdx.y * gridDim.x + blockIdx.x) + 7] = wall_clock64();     // (diagnostics: this thread's stores issued)
        if (g.colsum) {
            // Column sums of the tile as stored (rounded, masked): what pulse_colsum_partial_b16 would compute from the written matrix, taken
            // here while the values are in registers -- the bias gradient of the layer whose dZ this launch produces costs no pass over dZ.
            // The RPI thread rows of a column group are added in row order (fixed tree: deterministic).
            __syncthreads();                                        // every read of the transpose image is done
            float* red = sC;                                        // [RPI][128]
#pragma unroll
            for (int k = 0; k < 8; ++k) red[(tid >> 4) * PBN + c8 + k] = cs[k];
            __syncthreads();
            if (tid < PBN && n0 + tid < g.N) {
                float t = 0.f;
#pragma unroll 8
                for (int r = 0; r < RPI; ++r) t += red[r * PBN + tid];
                g.colsum[bz * g.sColsum + (long long)tm * g.ldcs + n0 + tid] = t;
            }
        }
    }
}

// AKC / BKC: operand stored [out][k] (reduction-contiguous); otherwise [k][out].
template <bool AKC, bool BKC, int WMW, int NPL>
__global__ void __launch_bounds__(128 * WMW) gemm_x3p_kernel(const XpArgs g) {
    static_assert(NPL == 1 || NPL == 3, "three planes (fp32-grade) or one (bf16 operands)");
    using G = XpGeom<WMW>;
    constexpr int BM = G::BM, NW = G::NW;
    extern __shared__ __attribute__((aligned(16))) char xp_smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;

    // XCD-aware remap (common.h map_workgroup; block b runs on XCD b % 8): every XCD owns a contiguous band of output tiles, or -- split-K
    // launches, i.e. the weight gradients -- a K RANGE of both operands, so an operand element crosses the fabric once instead of once per XCD
    const WgMap wgm = map_workgroup(g.tiles_m * g.tiles_n, g.batch, g.splitk);
    const int id = wgm.id;
    const int tm = id / g.tiles_n, tn = id - tm * g.tiles_n;
    const int m0 = tm * BM, n0 = tn * PBN;
    const int bz = wgm.bz, sp = wgm.sp;
    const int kbeg = sp * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int klen = kend - kbeg;
    const int nkt32 = (klen + PK - 1) / PK;                      // 32-deep k-tiles of this split
    const int kpad = nkt32 * PK;                                 // the planes are zero-padded to a multiple of 32 in k
    const int nkt = NPL == 3 ? nkt32 : (nkt32 + 2) / 3;          // pipeline stages: one k-tile of three planes, or three k-tiles of one

    // ---- buffer resources: one per plane and operand, based at this workgroup's tile origin, with the true extent (rows / outs past
    // the operand read as zero and write zeros into LDS; they only feed outputs that are never stored)
    const int extA = min(BM, g.M - m0), extB = min(PBN, g.N - n0);
    __amdgpu_buffer_rsrc_t rsA[3], rsB[3];                          // (a template-dependent extent here makes the host pass drop the kernel's stub)
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
        const unsigned short* a = g.A + bz * g.sA + p * g.pa + (AKC ? (long long)m0 * g.lda + kbeg : (long long)kbeg * g.lda + m0);
        const unsigned short* b = g.B + bz * g.sB + p * g.pb + (BKC ? (long long)n0 * g.ldb + kbeg : (long long)kbeg * g.ldb + n0);
        const unsigned ra = (unsigned)(AKC ? ((extA - 1) * g.lda + kpad) : ((klen - 1) * g.lda + ((extA + 7) & ~7))) * 2u;
        const unsigned rb = (unsigned)(BKC ? ((extB - 1) * g.ldb + kpad) : ((klen - 1) * g.ldb + ((extB + 7) & ~7))) * 2u;
        rsA[p] = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a), 0, klen > 0 ? ra : 0u, P_RSRC);
        rsB[p] = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(b), 0, klen > 0 ? rb : 0u, P_RSRC);
    }
    // per-lane DMA source offsets (bytes, constant); the k-tile / row-block advance is the scalar offset.  One wave instruction fills
    // 1024 consecutive LDS bytes (lane L: bytes 16 L ..), "row block" rb of a sub-tile:
    //   KC: 16 rows x 4 chunks: lane L -> row 16 rb + (L >> 2), slot L & 3 holding chunk (L & 3) ^ ((L >> 4) & 3)
    //   MC: 4 k rows x 16 pieces of one 128-out block: block rb >> 3, k rows 4 (rb & 7) + (L >> 4), slot L & 15 holding piece
    //       (L & 15) ^ ((L >> 4) << 2)
    const int voA = AKC ? ((lane >> 2) * g.lda + (((lane & 3) ^ ((lane >> 4) & 3)) << 3)) * 2
                        : ((lane >> 4) * g.lda + (((lane & 15) ^ ((lane >> 4) << 2)) << 3)) * 2;
    const int voB = BKC ? ((lane >> 2) * g.ldb + (((lane & 3) ^ ((lane >> 4) & 3)) << 3)) * 2
                        : ((lane >> 4) * g.ldb + (((lane & 15) ^ ((lane >> 4) << 2)) << 3)) * 2;
    const int ktA = AKC ? PK * 2 : PK * g.lda * 2, ktB = BKC ? PK * 2 : PK * g.ldb * 2;   // bytes per 32-deep k-tile
    // DMA of stage t into the stage buffer at byte offset stage_off: (3 BM / 16 + 3 * 128 / 16) wave instructions, dealt round-robin to
    // the waves: unit j of this wave is instruction i = wave + j NW (sub-tile and operand compile-time, row block wave + const)
    constexpr int DMA_PER_WAVE = (3 * BM / 16 + 3 * PBN / 16) / NW;
    auto issue_unit = [&](int stage_off, int t, int j) {
        constexpr int PA = BM / 16, PB = PBN / 16, NA = 3 * PA;
        static_assert(PA % NW == 0 && (PB % NW == 0), "row blocks per plane must be a multiple of the wave count");
        const int i0 = j * NW;
        if (i0 < NA) {
            const int sub = i0 / PA;
            const int rb = wave + (i0 % PA);
            const int kt = NPL == 3 ? t : 3 * t + sub;
            const int so = AKC ? kt * ktA + rb * 16 * g.lda * 2 : kt * ktA + (rb & 7) * 4 * g.lda * 2 + (rb >> 3) * 256;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA[NPL == 3 ? sub : 0], (lds_void_t*)(xp_smem + stage_off + sub * G::A_PLANE + rb * 1024), 16, voA, so, 0, 0);
        } else {
            const int ii0 = i0 - NA;
            const int sub = ii0 / PB;
            const int rb = wave + (ii0 % PB);
            const int kt = NPL == 3 ? t : 3 * t + sub;
            const int so = BKC ? kt * ktB + rb * 16 * g.ldb * 2 : kt * ktB + (rb & 7) * 4 * g.ldb * 2 + (rb >> 3) * 256;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB[NPL == 3 ? sub : 0], (lds_void_t*)(xp_smem + stage_off + G::A_IMG + sub * G::B_PLANE + rb * 1024), 16, voB, so, 0, 0);
        }
    };
    auto issue_tile = [&](int stage_off, int t) {
#pragma unroll
        for (int j = 0; j < DMA_PER_WAVE; ++j) issue_unit(stage_off, t, j);
    };
    // ---- L2 touch prefetch (single-plane mode).  Counters (profiles/r04_gemm_b16_pmc.txt): the bf16 kernel is not limited by the matrix pipe
    // or by power but by the latency of its stage DMA -- pipe 0.17-0.29 busy at 2.2-2.5 GHz, waves waiting 0.42-0.83 of their cycles: with two
    // stages in LDS only ONE stage of DMA is ever in flight, and a stage that misses L2 costs ~2 us against 0.6 us of MFMAs.  gfx950 has no
    // prefetch instruction, so TOUCH_AHEAD stages ahead every 128-byte line of the stage is touched by a 4-byte LDS-DMA load into a scratch
    // strip (no VGPR destination, no register hazard): the line is in L2 when the real DMA asks for it.  One wave instruction touches 64
    // lines; the touches are dealt round-robin to the waves and ALWAYS issued (past the reduction's end the buffer range check drops them),
    // so every wave has the same number of vector-memory operations in flight and the barrier waits can count them (vmcnt is in order).
    constexpr int TOUCH_AHEAD = 4;
    constexpr int LPR_A = BM * 2 / 128, LPR_B = PBN * 2 / 128;                 // lines per k row of a [red][out] tile
    // Only [red][out] operands are touched (the weight-gradient form: 354 -> 461 TFLOP/s in situ).  Measured on the forward form the touches
    // cost more than they return (491 -> 397 TFLOP/s: its stage is 16-row x 64-byte pieces, twelve more 64-line instructions per stage on
    // the same texture path the DMA uses), so reduction-contiguous operands are left alone.
    constexpr int TCH_A = AKC ? 0 : 96 * LPR_A / 64, TCH_B = BKC ? 0 : 96 * LPR_B / 64;   // wave instructions per stage
    constexpr int TPW = (NPL == 1 && TCH_A + TCH_B > 0) ? (TCH_A + TCH_B + NW - 1) / NW : 0;   // per wave (padded: the extra ones re-touch)
    auto touch_stage = [&](int t) {
        if constexpr (TPW > 0) {
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                const int u = (wave + j * NW) % (TCH_A + TCH_B);               // wave-uniform
                int vo, so;
                if (u < TCH_A) {
                    if constexpr (AKC) { vo = ((32 * u + (lane >> 1)) * g.lda) * 2 + (lane & 1) * 128; so = 3 * t * ktA; }
                    else { const int q = 64 * u + lane; vo = (q / LPR_A) * g.lda * 2 + (q % LPR_A) * 128; so = 3 * t * ktA; }
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA[0], (lds_void_t*)(xp_smem + G::TOUCH_OFF + wave * 256), 4, vo, so, 0, 0);
                } else {
                    const int ub = u - TCH_A;
                    if constexpr (BKC) { vo = ((32 * ub + (lane >> 1)) * g.ldb) * 2 + (lane & 1) * 128; so = 3 * t * ktB; }
                    else { const int q = 64 * ub + lane; vo = (q / LPR_B) * g.ldb * 2 + (q % LPR_B) * 128; so = 3 * t * ktB; }
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB[0], (lds_void_t*)(xp_smem + G::TOUCH_OFF + wave * 256), 4, vo, so, 0, 0);
                }
            }
        }
    };

    // ---- fragment read addresses (bytes, per lane; the sub-tile and the stage are immediates / added constants)
    //   KC: lane (l31, half) of k-step ks reads row (tile row + l31), chunk 2 ks + half -> slot (2 ks + half) ^ ((l31 >> 2) & 3)
    //   MC: 16-lane group gq = lane >> 4 covers outs 16 (gq & 1) .. + 15 of the 32-wide MFMA tile and k rows 8 (gq >> 1) .. + 7 of the
    //       k-step; lane 4 j + q of the group addresses k row j (second read: j + 4), 8-byte piece q of those 16 outs, and receives the
    //       four k values of out (lane & 15) (ds_read_b64_tr_b16; tools/tr_probe.cpp)
    int frA[2][2], frB[2][2];                                       // [mfma tile][k-step]
    {
        const int sw = (l31 >> 2) & 3;
        const int gq = lane >> 4, jj = (lane >> 2) & 3, qq = lane & 3;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if constexpr (AKC) {
                    frA[i][ks] = ((wm * 64 + i * 32 + l31) * 4 + ((2 * ks + half) ^ sw)) * 16;
                } else {
                    const int o = wm * 64 + i * 32 + 16 * (gq & 1);               // first out of this group inside the tile
                    const int piece = ((o & 127) >> 3) + (qq >> 1);
                    frA[i][ks] = (o >> 7) * 8192 + (ks * 16 + 8 * (gq >> 1) + jj) * 256 + ((piece ^ (jj << 2)) << 4) + (qq & 1) * 8;
                }
                if constexpr (BKC) {
                    frB[i][ks] = G::A_IMG + ((wn * 64 + i * 32 + l31) * 4 + ((2 * ks + half) ^ sw)) * 16;
                } else {
                    const int o = wn * 64 + i * 32 + 16 * (gq & 1);
                    const int piece = (o >> 3) + (qq >> 1);
                    frB[i][ks] = G::A_IMG + (ks * 16 + 8 * (gq >> 1) + jj) * 256 + ((piece ^ (jj << 2)) << 4) + (qq & 1) * 8;
                }
            }
    }

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

    // fragments: [set][sub-tile][mfma tile]
    bf16x8 fa[2][3][2], fb[2][3][2];
    auto rdA = [&](int addr) { if constexpr (AKC) return xp_lds128(addr); else return xp_lds_tr(addr, addr + 1024); };
    auto rdB = [&](int addr) { if constexpr (BKC) return xp_lds128(addr); else return xp_lds_tr(addr, addr + 1024); };
    // read unit u (0..11) of k-step ks of the stage at byte offset st into set S, in consumption order:
    //   three planes: term sequence (2,0) (0,2) (1,1) (1,0) (0,1) (0,0) -> A2 A2' B0 B0' | B2 B2' A0 A0' | A1 A1' B1 B1'
    //   one plane:    sub-tile 0, 1, 2 -> A A' B B' each
    auto frag_unit = [&](auto set_tag, int u, int st, int ks) {
        constexpr int S = decltype(set_tag)::value;
        const int grp = u >> 2, w = u & 3, i = w & 1;
        bool isA; int pl;
        if constexpr (NPL == 3) {
            isA = grp == 1 ? (w >= 2) : (w < 2);
            pl = isA ? (grp == 0 ? 2 : grp == 1 ? 0 : 1) : (grp == 0 ? 0 : grp == 1 ? 2 : 1);
        } else {
            isA = w < 2; pl = grp;
        }
        if (isA) fa[S][pl][i] = rdA(st + pl * G::A_PLANE + frA[i][ks]);
        else fb[S][pl][i] = rdB(st + pl * G::B_PLANE + frB[i][ks]);
    };
    // MFMAs of one k-step on set S (24 cross terms, or 12 diagonal ones of which those of k-tiles past the end are skipped: ``live`` =
    // k-tiles of this stage that exist, wave-uniform); after MFMA q the side work slot(q) runs
    constexpr int NQ = NPL == 3 ? 24 : 12;
    auto kstep = [&](auto set_tag, int live, auto&& slot) {
        constexpr int S = decltype(set_tag)::value;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int term = q >> 2, i = (q >> 1) & 1, j = q & 1;
            if constexpr (NPL == 3) {
                const int pa_ = term == 0 ? 2 : (term == 2 || term == 3) ? 1 : 0;
                const int pb_ = term == 1 ? 2 : (term == 2 || term == 4) ? 1 : 0;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[S][pa_][i], fb[S][pb_][j], acc[i][j], 0, 0, 0);
            } else {
                if (term < live) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[S][term][i], fb[S][term][j], acc[i][j], 0, 0, 0);
            }
            slot(q);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;

    // ---- prologue: stages 0 and 1 on their way, stage 0 landed, first fragments read
#pragma unroll
    for (int a = 2; a < TOUCH_AHEAD - 1; ++a) touch_stage(a);
    if (nkt > 0) issue_tile(0, 0);
    if (nkt > 1) issue_tile(G::STAGE, 1);
    touch_stage(TOUCH_AHEAD - 1);                                   // youngest: the only thing allowed in flight behind a stage's DMA at a barrier
    __builtin_amdgcn_sched_barrier(0);
    if (nkt > 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(DMA_PER_WAVE + TPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(TPW) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (nkt > 0) {
#pragma unroll
        for (int u = 0; u < 12; ++u) frag_unit(I0{}, u, 0, 0);
    }

    // Behind k-step 1's first MFMAs the first fragments of stage t + 1 are read (unconditionally: past the last stage they fetch stale LDS
    // bytes nobody uses); behind its MFMAs too (three planes: the later ones) the DMA of stage t + 2 is issued, one instruction per MFMA
    // gap, into the buffer this stage has just released (wave-uniform condition: the last two stages issue none).
    auto tile = [&](auto stage_tag, int t) {
        constexpr int CUR = decltype(stage_tag)::value * G::STAGE, OTH = G::STAGE - CUR;
        const bool more2 = t + 2 < nkt;
        const int live = NPL == 3 ? 3 : min(3, nkt32 - 3 * t);
        // k-step 0 on set 0; the 12 fragment reads of k-step 1 behind its first MFMAs
        kstep(I0{}, live, [&](int q) { if (q < 12) frag_unit(I1{}, q, CUR, 1); });
        // every wave has read what it needs of this stage (its reads are issued; lgkmcnt(0) completes them); stage t + 1 has landed
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(TPW) : "memory");   // stage t + 1 has landed; the touches behind it may still fly
        __builtin_amdgcn_sched_barrier(0);
        kstep(I1{}, live, [&](int q) {
            if (q < 12) frag_unit(I0{}, q, OTH, 0);
            constexpr int D0 = NPL == 3 ? 12 : 0;
            if (q >= D0 && q - D0 < DMA_PER_WAVE) { if (more2) issue_unit(CUR, t + 2, q - D0); }
            if (q == NQ - 1) touch_stage(t + TOUCH_AHEAD);
        });
    };
    for (int t = 0; t < nkt; t += 2) {
        tile(I0{}, t);
        if (t + 1 < nkt) tile(I1{}, t + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                // the epilogue reuses the staging buffers

    xp_epilogue<WMW, NPL>(g, acc, tid, wm, wn, half, l31, m0, n0, bz, sp, tm);
}

// =====================================================================================================================
// bf16-storage GEMM, three-stage ring ("b16r"): the single-plane kernel above is not limited by the matrix pipe or by power but by the
// latency of its stage DMA (profiles/r04_gemm_b16_pmc.txt: pipe 0.17-0.29 busy at 2.2-2.5 GHz, waves waiting 0.42-0.83 of their cycles):
// with two 72 KB stages only one stage of DMA is ever in flight, issued one stage (1536 SIMD cycles = 0.65 us) before it is needed.
// Same tile (256 x 128, 8 waves of 64 x 64), same fragment / MFMA / epilogue code, but
//   * a stage is 64 k (two 32-deep k-tiles, 48 KB) and THREE stages ring through the same 144 KB: two stages of DMA are in flight while the
//     third is multiplied, and a stage is issued two stages (2048 SIMD cycles) before its first fragment read;
//   * a reduction-contiguous operand is fetched in whole 128-byte lines: one wave instruction = 8 rows x 128 B (both k-tiles of a row),
//     LDS image [row][8 chunks of 16 B], chunk c of row r at slot c ^ ((r >> 1) & 7) -- conflict-free for the 16-lane service groups of
//     ds_read_b128 (see the note at the DMA offsets).  (The two-stage kernel fetches 16 rows x 64 B: every line is requested twice, by different instructions.)
//   * [red][out] operands keep the transposing-read image of the kernel above, two sub-tiles per stage.
// One barrier per stage, placed before the stage's last k-step: behind it the first fragments of the next stage are read and the DMA of
// stage t + 3 is issued into the buffer stage t has just released.
struct B16rGeom {
    static constexpr int BM = 256, NW = 8, NT = 512, SUBS = 2, NST = 3;
    static constexpr int A_IMG = BM * 64 * SUBS, B_IMG = PBN * 64 * SUBS, STAGE = A_IMG + B_IMG;       // 32 + 16 KB
    static constexpr int A_SUB = BM * 64, B_SUB = PBN * 64;                                         // one 32-deep sub-tile ([red][out] image)
    static constexpr int DMA_PER_WAVE = (STAGE / 1024) / NW;                                        // 6
};
static_assert(3 * B16rGeom::STAGE + XpGeom<4>::TOUCH_LDS <= XpGeom<4>::LDS, "the ring and the touch strip fit the launch's LDS");

template <bool AKC, bool BKC>
__global__ void __launch_bounds__(512) gemm_b16r_kernel(const XpArgs g) {
    using R = B16rGeom;
    extern __shared__ __attribute__((aligned(16))) char xp_smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;
    const WgMap wgm = map_workgroup(g.tiles_m * g.tiles_n, g.batch, g.splitk);
    const int id = wgm.id;
    const int tm = id / g.tiles_n, tn = id - tm * g.tiles_n;
    const int m0 = tm * R::BM, n0 = tn * PBN;
    const int bz = wgm.bz, sp = wgm.sp;
    const int kbeg = sp * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int klen = kend - kbeg;
    const int nkt32 = (klen + PK - 1) / PK;
    const int kpad = nkt32 * PK;
    const int nst = (nkt32 + 1) / 2;                                 // 64-deep stages

    const int extA = min(R::BM, g.M - m0), extB = min(PBN, g.N - n0);
    const unsigned short* a = g.A + bz * g.sA + (AKC ? (long long)m0 * g.lda + kbeg : (long long)kbeg * g.lda + m0);
    const unsigned short* b = g.B + bz * g.sB + (BKC ? (long long)n0 * g.ldb + kbeg : (long long)kbeg * g.ldb + n0);
    const unsigned ra = (unsigned)(AKC ? ((extA - 1) * g.lda + kpad) : ((klen - 1) * g.lda + ((extA + 7) & ~7))) * 2u;
    const unsigned rb_ = (unsigned)(BKC ? ((extB - 1) * g.ldb + kpad) : ((klen - 1) * g.ldb + ((extB + 7) & ~7))) * 2u;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a), 0, klen > 0 ? ra : 0u, P_RSRC);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(b), 0, klen > 0 ? rb_ : 0u, P_RSRC);
    // per-lane DMA source offsets (bytes).  KC: lane L -> row L >> 3 of the 8-row block, LDS slot L & 7 holding chunk (L & 7) ^ (L >> 3).
    // MC: as in the two-stage kernel (4 k rows x 16 pieces).
    // KC swizzle: chunk c of row r sits at slot c ^ ((r >> 1) & 7).  ds_read_b128 is serviced in four 16-lane groups ({0-3, 12-15, 20-27},
    // {4-11, 16-19, 28-31} and the same + 32: MI355X_MICROARCH.md, LDS table) over a 256-byte bank row = two 128-byte tile rows: the eight even
    // and the eight odd rows of every group then carry eight different values of (r >> 1) & 7 -- conflict-free.  (c ^ (r & 7), the first
    // version, put rows 12 and 20 of a group on the same slot: SQ_LDS_BANK_CONFLICT was half of SQ_LDS_IDX_ACTIVE.)  Row 8 rb + (L >> 3) of
    // instruction rb: (r >> 1) & 7 = (4 rb + (L >> 4)) & 7, and rb has the wave's parity.
    const int kcx = (4 * (wave & 1) + (lane >> 4)) & 7;
    const int voA = AKC ? ((lane >> 3) * g.lda + (((lane & 7) ^ kcx) << 3)) * 2
                        : ((lane >> 4) * g.lda + (((lane & 15) ^ ((lane >> 4) << 2)) << 3)) * 2;
    const int voB = BKC ? ((lane >> 3) * g.ldb + (((lane & 7) ^ kcx) << 3)) * 2
                        : ((lane >> 4) * g.ldb + (((lane & 15) ^ ((lane >> 4) << 2)) << 3)) * 2;
    const int stA = AKC ? 64 * 2 : 64 * g.lda * 2, stB = BKC ? 64 * 2 : 64 * g.ldb * 2;      // bytes per 64-deep stage
    // unit j of this wave = instruction i = wave + 8 j of the stage's 48 (32 of A, 16 of B)
    auto issue_unit = [&](int stage_off, int t, int j) {
        const int i = wave + j * R::NW;
        if (j * R::NW < 32) {                                        // (j < 4: A; compile-time after unrolling)
            int so, dst;
            if constexpr (AKC) { so = t * stA + i * 8 * g.lda * 2; dst = i * 1024; }
            else {
                const int sub = i >> 4, rb = i & 15;                  // sub-tile, row block (8 k groups x 2 out halves of 128)
                so = t * stA + sub * 32 * g.lda * 2 + (rb & 7) * 4 * g.lda * 2 + (rb >> 3) * 256;
                dst = sub * R::A_SUB + rb * 1024;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_t*)(xp_smem + stage_off + dst), 16, voA, so, 0, 0);
        } else {
            const int ib = i - 32;
            int so, dst;
            if constexpr (BKC) { so = t * stB + ib * 8 * g.ldb * 2; dst = ib * 1024; }
            else {
                const int sub = ib >> 3, rb = ib & 7;                 // 8 k groups, one out block of 128
                so = t * stB + sub * 32 * g.ldb * 2 + rb * 4 * g.ldb * 2;
                dst = sub * R::B_SUB + rb * 1024;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_t*)(xp_smem + stage_off + R::A_IMG + dst), 16, voB, so, 0, 0);
        }
    };
    auto issue_stage = [&](int stage_off, int t) {
#pragma unroll
        for (int j = 0; j < R::DMA_PER_WAVE; ++j) issue_unit(stage_off, t, j);
    };
    // L2 touch prefetch of [red][out] operands (see the two-stage kernel): every 128-byte line of stage t is touched two stages before its
    // DMA is issued.  Always issued (past the reduction's end the range check drops them): the barrier waits count them.
    constexpr int LPR_A = R::BM * 2 / 128, LPR_B = PBN * 2 / 128;
    // (only the weight-gradient form, both operands [red][out]: measured, the touches cost the mixed form 4-8 %)
    constexpr int TCH_A = (AKC || BKC) ? 0 : 64 * LPR_A / 64, TCH_B = (AKC || BKC) ? 0 : 64 * LPR_B / 64;
    constexpr int TPW = (TCH_A + TCH_B > 0) ? (TCH_A + TCH_B + R::NW - 1) / R::NW : 0;
    auto touch_stage = [&](int t) {
        if constexpr (TPW > 0) {
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                const int u = (wave + j * R::NW) % (TCH_A + TCH_B);
                if (u < TCH_A) {
                    const int q = 64 * u + lane;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_t*)(xp_smem + XpGeom<4>::TOUCH_OFF + wave * 256), 4,
                                                             (q / LPR_A) * g.lda * 2 + (q % LPR_A) * 128, t * stA, 0, 0);
                } else {
                    const int q = 64 * (u - TCH_A) + lane;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_t*)(xp_smem + XpGeom<4>::TOUCH_OFF + wave * 256), 4,
                                                             (q / LPR_B) * g.ldb * 2 + (q % LPR_B) * 128, t * stB, 0, 0);
                }
            }
        }
    };

    // fragment read addresses of k-step q4 = 2 kt + ks (kt: 32-deep k-tile of the stage, ks: its 16-deep half), MFMA tile i
    int frA[2][4], frB[2][4];
    {
        const int gq = lane >> 4, jj = (lane >> 2) & 3, qq = lane & 3;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int kt = q4 >> 1, ks = q4 & 1;
                if constexpr (AKC) {
                    const int row = wm * 64 + i * 32 + l31;
                    frA[i][q4] = row * 128 + (((4 * kt + 2 * ks + half) ^ ((row >> 1) & 7)) << 4);
                } else {
                    const int o = wm * 64 + i * 32 + 16 * (gq & 1);
                    const int piece = ((o & 127) >> 3) + (qq >> 1);
                    frA[i][q4] = kt * R::A_SUB + (o >> 7) * 8192 + (ks * 16 + 8 * (gq >> 1) + jj) * 256 + ((piece ^ (jj << 2)) << 4) + (qq & 1) * 8;
                }
                if constexpr (BKC) {
                    const int row = wn * 64 + i * 32 + l31;
                    frB[i][q4] = R::A_IMG + row * 128 + (((4 * kt + 2 * ks + half) ^ ((row >> 1) & 7)) << 4);
                } else {
                    const int o = wn * 64 + i * 32 + 16 * (gq & 1);
                    const int piece = (o >> 3) + (qq >> 1);
                    frB[i][q4] = R::A_IMG + kt * R::B_SUB + (ks * 16 + 8 * (gq >> 1) + jj) * 256 + ((piece ^ (jj << 2)) << 4) + (qq & 1) * 8;
                }
            }
    }

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

    bf16x8 fa[2][2], fb[2][2];                                       // [set][mfma tile]
    auto rdA = [&](int addr) { if constexpr (AKC) return xp_lds128(addr); else return xp_lds_tr(addr, addr + 1024); };
    auto rdB = [&](int addr) { if constexpr (BKC) return xp_lds128(addr); else return xp_lds_tr(addr, addr + 1024); };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    auto frag_unit = [&](auto set_tag, int u, int st, int q4) {       // u = 0..3: A0 A1 B0 B1
        constexpr int S = decltype(set_tag)::value;
        if (u < 2) fa[S][u] = rdA(st + frA[u][q4]);
        else fb[S][u - 2] = rdB(st + frB[u - 2][q4]);
    };
    // the four MFMAs of a k-step on set S, slot(q) after each
    auto kstep = [&](auto set_tag, bool live, auto&& slot) {
        constexpr int S = decltype(set_tag)::value;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = q >> 1, j = q & 1;
            if (live) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[S][i], fb[S][j], acc[i][j], 0, 0, 0);
            slot(q);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- prologue: stages 0, 1, 2 on their way; stage 0 landed; its first fragments read
    if (nst > 0) issue_stage(0, 0);
    if (nst > 1) issue_stage(R::STAGE, 1);
    if (nst > 2) issue_stage(2 * R::STAGE, 2);
    touch_stage(3);
    touch_stage(4);
    __builtin_amdgcn_sched_barrier(0);
    if (nst > 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * R::DMA_PER_WAVE + 2 * TPW) : "memory");
    else if (nst > 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(R::DMA_PER_WAVE + 2 * TPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * TPW) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (nst > 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) frag_unit(I0{}, u, 0, 0);
    }

    int cur = 0, nxt = R::STAGE;                                     // byte offsets of the stage being multiplied and of the next one
    for (int t = 0; t < nst; ++t) {
        const bool live1 = 2 * t + 1 < nkt32;                        // the stage's second k-tile exists (wave-uniform)
        kstep(I0{}, true, [&](int q) { frag_unit(I1{}, q, cur, 1); });
        kstep(I1{}, true, [&](int q) { frag_unit(I0{}, q, cur, 2); });
        kstep(I0{}, live1, [&](int q) { frag_unit(I1{}, q, cur, 3); });
        // every fragment read of this stage is issued (lgkmcnt(0) completes them); stage t + 1 has landed (stage t + 2, if it was issued, may
        // still be in flight: it is the youngest DMA of this wave)
        __builtin_amdgcn_sched_barrier(0);
        if (t + 2 < nst) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(R::DMA_PER_WAVE + 2 * TPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * TPW) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        const bool more3 = t + 3 < nst;
        kstep(I1{}, live1, [&](int q) {
            frag_unit(I0{}, q, nxt, 0);                               // (past the last stage: stale bytes nobody multiplies)
            if (more3) { issue_unit(cur, t + 3, q); if (q < R::DMA_PER_WAVE - 4) issue_unit(cur, t + 3, q + 4); }
            if (q == 3) touch_stage(t + 5);
        });
        cur = nxt;
        nxt = nxt + R::STAGE == R::NST * R::STAGE ? 0 : nxt + R::STAGE;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                // the epilogue reuses the staging buffers
    xp_epilogue<4, 1>(g, acc, tid, wm, wn, half, l31, m0, n0, bz, sp, tm);
}

// =====================================================================================================================
// bf16-storage GEMM, 256 x 256 tile ("b16w"): the ring kernel above stages 384 bytes per MFMA (48 KB per 128 MFMAs of a stage) and is bound
// by the L2 -> LDS path, not by the matrix pipe (profiles/r04_gemm_b16_pmc.txt: pipe 0.24-0.32 busy, no bank conflicts; tools/mfma_feed_probe:
// LDS-DMA sustains 9.6 TB/s from L2 and 6.2-6.7 TB/s from the Infinity Cache / HBM, i.e. 820 / 550 TFLOP/s at 384 B per MFMA).  This kernel
// is the same code on a 256 x 256 output tile: 8 waves of 64 x 128 (eight accumulator tiles, 128 VGPRs), a stage of 64 k = 64 KB for 256
// MFMAs = 256 bytes per MFMA, six fragment reads per eight MFMAs instead of four per four.  Two stages ring through 128 KB: a stage holds twice
// the MFMA work of the ring kernel's, so "issued one stage ahead" is the same 2048 SIMD cycles of lead.  The tile is two 256 x 128 tiles side
// by side (column half p: columns 128 p + 64 wn + 32 j), so images, fragment addresses and the epilogue are the ring kernel's, used twice.
// Used when it does not cost the launch a round of workgroups (xp_wide_tiles).
struct B16wGeom {
    static constexpr int BM = 256, BN = 256, NW = 8, NT = 512, SUBS = 2, NST = 2;
    static constexpr int A_IMG = BM * 64 * SUBS, B_IMG = BN * 64 * SUBS, STAGE = A_IMG + B_IMG;        // 32 + 32 KB
    static constexpr int A_SUB = BM * 64, B_SUB = BN * 64;
    static constexpr int DMA_PER_WAVE = (STAGE / 1024) / NW;                                        // 8
};
static_assert(2 * B16wGeom::STAGE + XpGeom<4>::TOUCH_LDS <= XpGeom<4>::LDS, "two stages and the touch strip fit the launch's LDS");

template <bool AKC, bool BKC>
__global__ void __launch_bounds__(512) gemm_b16w_kernel(const XpArgs g) {
    using R = B16wGeom;
    extern __shared__ __attribute__((aligned(16))) char xp_smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;
    const WgMap wgm = map_workgroup(g.tiles_m * g.tiles_n, g.batch, g.splitk);
    const int id = wgm.id;
    const int tm = id / g.tiles_n, tn = id - tm * g.tiles_n;
    const int m0 = tm * R::BM, n0 = tn * R::BN;
    const int bz = wgm.bz, sp = wgm.sp;
    const int kbeg = sp * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int klen = kend - kbeg;
    const int nkt32 = (klen + PK - 1) / PK;
    const int kpad = nkt32 * PK;
    const int nst = (nkt32 + 1) / 2;                                 // 64-deep stages

    const int extA = min(R::BM, g.M - m0), extB = min(R::BN, g.N - n0);
    const unsigned short* a = g.A + bz * g.sA + (AKC ? (long long)m0 * g.lda + kbeg : (long long)kbeg * g.lda + m0);
    const unsigned short* b = g.B + bz * g.sB + (BKC ? (long long)n0 * g.ldb + kbeg : (long long)kbeg * g.ldb + n0);
    const unsigned ra = (unsigned)(AKC ? ((extA - 1) * g.lda + kpad) : ((klen - 1) * g.lda + ((extA + 7) & ~7))) * 2u;
    const unsigned rb_ = (unsigned)(BKC ? ((extB - 1) * g.ldb + kpad) : ((klen - 1) * g.ldb + ((extB + 7) & ~7))) * 2u;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a), 0, klen > 0 ? ra : 0u, P_RSRC);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(b), 0, klen > 0 ? rb_ : 0u, P_RSRC);
    // DMA lane offsets and LDS images: exactly the ring kernel's (KC: 8 rows x 128 B per instruction, chunk c of row r at slot
    // c ^ ((r >> 1) & 7); MC: 4 k rows x 16 pieces of a 128-out block); both operands are 256 rows / columns = 32 instructions per stage each
    const int kcx = (4 * (wave & 1) + (lane >> 4)) & 7;
    const int voA = AKC ? ((lane >> 3) * g.lda + (((lane & 7) ^ kcx) << 3)) * 2
                        : ((lane >> 4) * g.lda + (((lane & 15) ^ ((lane >> 4) << 2)) << 3)) * 2;
    const int voB = BKC ? ((lane >> 3) * g.ldb + (((lane & 7) ^ kcx) << 3)) * 2
                        : ((lane >> 4) * g.ldb + (((lane & 15) ^ ((lane >> 4) << 2)) << 3)) * 2;
    const int stA = AKC ? 64 * 2 : 64 * g.lda * 2, stB = BKC ? 64 * 2 : 64 * g.ldb * 2;      // bytes per 64-deep stage
    // unit j of this wave = instruction i = wave + 8 j of the stage's 64 (32 of A, 32 of B)
    auto issue_unit = [&](int stage_off, int t, int j) {
        const int i = wave + (j & 3) * R::NW;                        // row block inside the operand
        const int sub = i >> 4, rb = i & 15;                         // [red][out]: sub-tile, (8 k groups x 2 out blocks of 128)
        if (j < 4) {
            int so, dst;
            if constexpr (AKC) { so = t * stA + i * 8 * g.lda * 2; dst = i * 1024; }
            else { so = t * stA + sub * 32 * g.lda * 2 + (rb & 7) * 4 * g.lda * 2 + (rb >> 3) * 256; dst = sub * R::A_SUB + rb * 1024; }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_t*)(xp_smem + stage_off + dst), 16, voA, so, 0, 0);
        } else {
            int so, dst;
            if constexpr (BKC) { so = t * stB + i * 8 * g.ldb * 2; dst = i * 1024; }
            else { so = t * stB + sub * 32 * g.ldb * 2 + (rb & 7) * 4 * g.ldb * 2 + (rb >> 3) * 256; dst = sub * R::B_SUB + rb * 1024; }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_t*)(xp_smem + stage_off + R::A_IMG + dst), 16, voB, so, 0, 0);
        }
    };
    // L2 touch prefetch of the weight-gradient form's operands, one stage ahead of the stage's DMA: 4 lines per k row and operand, 64 k rows
    constexpr int LPR = R::BM * 2 / 128;
    constexpr int TCH = (AKC || BKC) ? 0 : 64 * LPR / 64;            // wave instructions per operand and stage
    constexpr int TPW = TCH > 0 ? (2 * TCH + R::NW - 1) / R::NW : 0;
    auto touch_stage = [&](int t) {
        if constexpr (TPW > 0) {
            const int u = wave % (2 * TCH);
            const int q = 64 * (u % TCH) + lane;
            if (u < TCH)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_t*)(xp_smem + XpGeom<4>::TOUCH_OFF + wave * 256), 4,
                                                         (q / LPR) * g.lda * 2 + (q % LPR) * 128, t * stA, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_t*)(xp_smem + XpGeom<4>::TOUCH_OFF + wave * 256), 4,
                                                         (q / LPR) * g.ldb * 2 + (q % LPR) * 128, t * stB, 0, 0);
        }
    };

    // fragment read addresses of k-step q4 = 2 kt + ks: A tile i (rows 64 wm + 32 i), B tile u = 2 p + j (columns 128 p + 64 wn + 32 j)
    int frA[2][4], frB[4][4];
    {
        const int gq = lane >> 4, jj = (lane >> 2) & 3, qq = lane & 3;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int kt = q4 >> 1, ks = q4 & 1;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if constexpr (AKC) {
                    const int row = wm * 64 + i * 32 + l31;
                    frA[i][q4] = row * 128 + (((4 * kt + 2 * ks + half) ^ ((row >> 1) & 7)) << 4);
                } else {
                    const int o = wm * 64 + i * 32 + 16 * (gq & 1);
                    const int piece = ((o & 127) >> 3) + (qq >> 1);
                    frA[i][q4] = kt * R::A_SUB + (o >> 7) * 8192 + (ks * 16 + 8 * (gq >> 1) + jj) * 256 + ((piece ^ (jj << 2)) << 4) + (qq & 1) * 8;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c0 = (u >> 1) * 128 + wn * 64 + (u & 1) * 32;
                if constexpr (BKC) {
                    const int row = c0 + l31;
                    frB[u][q4] = R::A_IMG + row * 128 + (((4 * kt + 2 * ks + half) ^ ((row >> 1) & 7)) << 4);
                } else {
                    const int o = c0 + 16 * (gq & 1);
                    const int piece = ((o & 127) >> 3) + (qq >> 1);
                    frB[u][q4] = R::A_IMG + kt * R::B_SUB + (o >> 7) * 8192 + (ks * 16 + 8 * (gq >> 1) + jj) * 256 + ((piece ^ (jj << 2)) << 4) + (qq & 1) * 8;
                }
            }
        }
    }

    f32x16 acc[2][2][2];                                             // [column half p][i][j]
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        float b0 = 0.f, b1 = 0.f;
        if (g.epi == 0 && g.bias) {
            const float* bias = g.bias + bz * g.sBias;
            const int c0 = n0 + p * 128 + wn * 64 + l31;
            if (c0 < g.N) b0 = bias[c0];
            if (c0 + 32 < g.N) b1 = bias[c0 + 32];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[p][i][0][r] = b0; acc[p][i][1][r] = b1; }
    }

    bf16x8 fa[2][2], fb[2][4];                                       // [set][tile]
    auto rdA = [&](int addr) { if constexpr (AKC) return xp_lds128(addr); else return xp_lds_tr(addr, addr + 1024); };
    auto rdB = [&](int addr) { if constexpr (BKC) return xp_lds128(addr); else return xp_lds_tr(addr, addr + 1024); };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    auto frag_unit = [&](auto set_tag, int u, int st, int q4) {       // u = 0..5: A0 A1 B0 B1 B2 B3
        constexpr int S = decltype(set_tag)::value;
        if (u < 2) fa[S][u] = rdA(st + frA[u][q4]);
        else if (u < 6) fb[S][u - 2] = rdB(st + frB[u - 2][q4]);
    };
    // the eight MFMAs of a k-step on set S (A tile outermost: each A fragment feeds four consecutive MFMAs), slot(q) after each
    auto kstep = [&](auto set_tag, bool live, auto&& slot) {
        constexpr int S = decltype(set_tag)::value;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int i = q >> 2, p = (q >> 1) & 1, j = q & 1;
            if (live) acc[p][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[S][i], fb[S][2 * p + j], acc[p][i][j], 0, 0, 0);
            slot(q);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto issue_stage = [&](int stage_off, int t) {
#pragma unroll
        for (int j = 0; j < R::DMA_PER_WAVE; ++j) issue_unit(stage_off, t, j);
    };

    long long dbg_w[3] = {0, 0, 0};
    if (g.dbg) dbg_w[0] = wall_clock64();
    // ---- prologue: stages 0 and 1 on their way, the lines of stage 2 touched; stage 0 landed; its first fragments read.
    // vm queue order from here on: [DMA(t + 1) x 8, touch(t + 2)] at the barrier of stage t: vmcnt(TPW) = "DMA(t + 1) has landed".
    if (nst > 0) issue_stage(0, 0);
    if (nst > 1) issue_stage(R::STAGE, 1);
    touch_stage(2);
    __builtin_amdgcn_sched_barrier(0);
    if (nst > 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(R::DMA_PER_WAVE + TPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(TPW) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (nst > 0) {
#pragma unroll
        for (int u = 0; u < 6; ++u) frag_unit(I0{}, u, 0, 0);
    }

    if (g.dbg) dbg_w[1] = wall_clock64();
    int cur = 0, nxt = R::STAGE;
    for (int t = 0; t < nst; ++t) {
        const bool live1 = 2 * t + 1 < nkt32;                        // the stage's second k-tile exists (wave-uniform)
        kstep(I0{}, true, [&](int q) { frag_unit(I1{}, q, cur, 1); });
        kstep(I1{}, true, [&](int q) { frag_unit(I0{}, q, cur, 2); });
        kstep(I0{}, live1, [&](int q) { frag_unit(I1{}, q, cur, 3); });
        // every fragment read of this stage is issued (lgkmcnt(0) completes them): its buffer is free behind the barrier; stage t + 1 has landed
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(TPW) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        const bool more2 = t + 2 < nst;
        kstep(I1{}, live1, [&](int q) {
            frag_unit(I0{}, q, nxt, 0);                               // (past the last stage: stale bytes nobody multiplies)
            if (more2) issue_unit(cur, t + 2, q);
            if (q == 7) touch_stage(t + 3);
        });
        const int tmp = cur; cur = nxt; nxt = tmp;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                // the epilogue reuses the staging buffers
    if (g.dbg) dbg_w[2] = wall_clock64();
    xp_epilogue<4, 1>(g, acc[0], tid, wm, wn, half, l31, m0, n0, bz, sp, tm);
    if (n0 + PBN < g.N) {                                           // (workgroup-uniform)
        __syncthreads();
        xp_epilogue<4, 1>(g, acc[1], tid, wm, wn, half, l31, m0, n0 + PBN, bz, sp, tm);
    }
    if (g.dbg && tid == 0) {                                        // start | first stage landed | main loop done | epilogue's stores issued | all of them acknowledged
        long long* o = g.dbg + 8 * (blockIdx.y * gridDim.x + blockIdx.x);
        o[0] = dbg_w[0]; o[1] = dbg_w[1]; o[2] = dbg_w[2]; o[3] = wall_clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        o[4] = wall_clock64();
        o[5] = (long long)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);      // XCC_ID
    }
}

// ---- fp32 matrix -> three bf16 planes (optionally transposed); pad columns [cols, ld_out) of every written row are zero-filled ----
// out plane p, element (r, c) at out[p * plane_stride + r * ld_out + c].  transpose: out(r, c) = in(c, r) (rows_out = cols_in).
__global__ void __launch_bounds__(256) split_planes_kernel(const float* __restrict__ in, long long ld_in, int rows_out, int cols_out,
                                                          unsigned short* __restrict__ out, long long plane_stride, int ld_out, int transpose,
                                                          const long long* __restrict__ row_idx, int vec_in) {
    const int pieces = ld_out >> 3;                                 // 16-byte pieces per output row
    const long long total = (long long)rows_out * pieces;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / pieces), c0 = (int)(i - (long long)r * pieces) * 8;
        float v[8];
        if (!transpose) {
            const float* src = in + (row_idx ? row_idx[r] : (long long)r) * ld_in + c0;
            if (vec_in && c0 + 7 < cols_out) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = c0 + k < cols_out ? src[k] : 0.f;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = c0 + k < cols_out ? in[(long long)(c0 + k) * ld_in + r] : 0.f;
        }
        u32x4 q0, q1, q2;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned x0, x1, x2;
            xp_split_pair(v[2 * k], v[2 * k + 1], x0, x1, x2);
            q0[k] = x0; q1[k] = x1; q2[k] = x2;
        }
        unsigned short* o = out + (long long)r * ld_out + c0;
        *reinterpret_cast<u32x4*>(o) = q0;
        if (plane_stride) {                                         // plane_stride 0: a plain bf16 matrix (plane 0 only)
            *reinterpret_cast<u32x4*>(o + plane_stride) = q1;
            *reinterpret_cast<u32x4*>(o + 2 * plane_stride) = q2;
        }
    }
}

}  // namespace pulse

using namespace pulse;

extern "C" {

int pulse_sizeof_gemm_x3p_desc(void) { return (int)sizeof(pulse_gemm_x3p_desc); }

int pulse_split_planes(const float* in, int64_t ld_in, int32_t rows_out, int32_t cols_out, void* out, int64_t plane_stride, int32_t ld_out,
                       int32_t transpose, const int64_t* row_idx, pulse_stream_t s) {
    PULSE_REQUIRE(rows_out >= 0 && cols_out >= 0, "pulse_split_planes: negative size");
    if (rows_out == 0) return PULSE_OK;
    PULSE_REQUIRE(in && out, "pulse_split_planes: null pointer");
    PULSE_REQUIRE(ld_out % 8 == 0 && ld_out >= cols_out && (plane_stride == 0 || plane_stride >= (int64_t)rows_out * ld_out) && plane_stride % 8 == 0,
                  "pulse_split_planes: ld_out must be a multiple of 8 covering cols_out, plane_stride a multiple of 8 covering the plane (or 0: one plane)");
    PULSE_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0, "pulse_split_planes: out must be 16-byte aligned");
    const int vec_in = !transpose && (ld_in % 4) == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0;      // 16-byte loads when the rows allow them
    PULSE_REQUIRE(!(transpose && row_idx), "pulse_split_planes: row_idx with transpose is not supported");
    const long long total = (long long)rows_out * (ld_out / 8);
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(s), in, (long long)ld_in, rows_out, cols_out,
                       reinterpret_cast<unsigned short*>(out), (long long)plane_stride, ld_out, transpose, reinterpret_cast<const long long*>(row_idx), vec_in);
    return check_launch("pulse_split_planes");
}

// PULSE_B16_RING=0: keep the two-stage kernel for the 256-row bf16-storage launches (A/B switch, read once)
static const bool g_b16_ring = [] { const char* v = getenv("PULSE_B16_RING"); return !(v && v[0] == '0'); }();

static bool xp_big_tiles(int M, int N, int batch, int split_k) {
    // 256-row tiles when they still give every CU a workgroup; otherwise 128-row tiles (4 waves)
    const long long t256 = (long long)((M + 255) / 256) * ((N + PBN - 1) / PBN) * batch * split_k;
    return t256 >= 256 || M > 128 * 64;
}

// 256 x 256 tiles (gemm_b16w_kernel) when they do not cost the launch a round of workgroups: a wide workgroup does the work of two narrow
// ones, so it wins whenever 2 x rounds(wide) <= rounds(narrow) on the 256 CUs.  gemm option 3: 1 = never, 2 = whenever the tile has a second half.
static const bool g_b16_wide = [] { const char* v = getenv("PULSE_B16_WIDE"); return !(v && v[0] == '0'); }();     // A/B switch, read once
static bool xp_wide_tiles(int M, int N, int batch, int split_k) {
    const int opt = gemm_option(3);
    if (opt == 1 || N <= PBN || (!g_b16_wide && opt != 2)) return false;
    if (opt == 2) return true;
    const long long tm = (M + 255) / 256, bs = (long long)batch * split_k;
    const long long tn = tm * ((N + PBN - 1) / PBN) * bs, tw = tm * ((N + 255) / 256) * bs;
    return 2 * ((tw + 255) / 256) <= (tn + 255) / 256;
}

int pulse_gemm_x3p_row_tiles(int32_t M, int32_t N, int32_t batch) {
    if (M <= 0 || N <= 0 || batch <= 0) return 0;
    return xp_big_tiles(M, N, batch, 1) ? (M + 255) / 256 : (M + 127) / 128;
}

int pulse_gemm_x3p(const pulse_gemm_x3p_desc* d, pulse_stream_t s) {
    PULSE_REQUIRE(d != nullptr, "pulse_gemm_x3p: null descriptor");
    PULSE_REQUIRE(d->M >= 0 && d->N >= 0 && d->K >= 0, "pulse_gemm_x3p: negative size");
    if (d->M == 0 || d->N == 0 || d->batch == 0) return PULSE_OK;
    PULSE_REQUIRE(d->A && d->B && (d->C || d->Cp), "pulse_gemm_x3p: null operand / no output");
    PULSE_REQUIRE(d->batch >= 1 && d->split_k >= 1, "pulse_gemm_x3p: batch / split_k must be >= 1");
    PULSE_REQUIRE(d->planes == 0 || d->planes == 1 || d->planes == 3, "pulse_gemm_x3p: planes must be 3 (fp32-grade; 0 means 3) or 1 (bf16 operands)");
    const int npl = d->planes == 1 ? 1 : 3;
    const bool akc = d->a_layout == PULSE_GEMM_RED_CONTIG, bkc = d->b_layout == PULSE_GEMM_RED_CONTIG;
    PULSE_REQUIRE(akc == bkc || (akc && !bkc), "pulse_gemm_x3p: layout combination (A out-contiguous, B reduction-contiguous) unsupported");
    PULSE_REQUIRE((d->lda % 8) == 0 && (d->ldb % 8) == 0 && (d->a_plane_stride % 8) == 0 && (d->b_plane_stride % 8) == 0 &&
                  (d->stride_a % 8) == 0 && (d->stride_b % 8) == 0, "pulse_gemm_x3p: operand pitches / strides must be multiples of 8 elements");
    PULSE_REQUIRE((reinterpret_cast<uintptr_t>(d->A) & 15) == 0 && (reinterpret_cast<uintptr_t>(d->B) & 15) == 0, "pulse_gemm_x3p: A / B must be 16-byte aligned");
    const int kpad = (d->K + PK - 1) / PK * PK;
    // reduction-contiguous rows must hold the zero-padded k extent; [k][out] operands must hold roundup8(extent) columns
    PULSE_REQUIRE(akc ? d->lda >= kpad : d->lda >= ((d->M + 7) & ~7), "pulse_gemm_x3p: lda too small (k is padded to a multiple of 32 with zeros)");
    PULSE_REQUIRE(bkc ? d->ldb >= kpad : d->ldb >= ((d->N + 7) & ~7), "pulse_gemm_x3p: ldb too small (k is padded to a multiple of 32 with zeros)");
    PULSE_REQUIRE(d->split_k == 1 || (akc == false), "pulse_gemm_x3p: split-K is for the [red][out] x [red][out] (weight-gradient) form");
    PULSE_REQUIRE(!d->C || d->ldc >= d->N, "pulse_gemm_x3p: ldc too small");
    PULSE_REQUIRE(!d->C || ((reinterpret_cast<uintptr_t>(d->C) & 15) == 0 && (d->ldc % 4) == 0 && (d->stride_c % 4) == 0 && (d->split_stride % 4) == 0),
                  "pulse_gemm_x3p: C rows must be 16-byte aligned");
    PULSE_REQUIRE(!d->Cp || ((reinterpret_cast<uintptr_t>(d->Cp) & 15) == 0 && (d->ldcp % 8) == 0 && d->ldcp >= ((d->N + 7) & ~7) && (d->c_plane_stride % 8) == 0 &&
                             (d->stride_cp % 8) == 0), "pulse_gemm_x3p: Cp rows must be 16-byte aligned and hold roundup8(N) columns");
    PULSE_REQUIRE(!d->Cp || d->split_k == 1, "pulse_gemm_x3p: split-K slabs carry no planes");
    PULSE_REQUIRE(d->epilogue >= 0 && d->epilogue <= 2 && d->activation >= 0 && d->activation <= 2, "pulse_gemm_x3p: bad epilogue / activation");
    PULSE_REQUIRE(d->epilogue == 0 || d->aux != nullptr || (d->epilogue == PULSE_EPI_RELU_GRAD && d->relu_mask8 != nullptr),
                  "pulse_gemm_x3p: gradient epilogue needs aux (or, for relu-grad, relu_mask8)");
    const bool mask8_on = d->relu_mask8 != nullptr && ((d->epilogue == PULSE_EPI_RELU_GRAD && d->aux == nullptr) ||
                                                       (d->epilogue == PULSE_EPI_BIAS_ACT && d->activation == PULSE_ACT_RELU));
    PULSE_REQUIRE(!mask8_on || (d->ld_mask8 >= (d->N + 7) / 8 && d->split_k == 1), "pulse_gemm_x3p: relu_mask8 needs ld_mask8 >= roundup8(N) / 8 and no split-K");
    if (d->aux_is_bf16) {
        PULSE_REQUIRE(!d->aux || ((reinterpret_cast<uintptr_t>(d->aux) & 15) == 0 && (d->ldaux % 8) == 0 && (d->stride_aux % 8) == 0 && d->ldaux >= ((d->N + 7) & ~7)),
                      "pulse_gemm_x3p: bf16 aux rows must be 16-byte aligned and hold roundup8(N) columns");
    } else {
        PULSE_REQUIRE(!d->aux || ((reinterpret_cast<uintptr_t>(d->aux) & 15) == 0 && (d->ldaux % 4) == 0 && (d->stride_aux % 4) == 0), "pulse_gemm_x3p: aux rows must be 16-byte aligned");
    }
    PULSE_REQUIRE(!d->C2 || ((reinterpret_cast<uintptr_t>(d->C2) & 15) == 0 && (d->ldc2 % 4) == 0 && (d->stride_c2 % 4) == 0), "pulse_gemm_x3p: C2 rows must be 16-byte aligned");
    PULSE_REQUIRE(d->split_k == 1 || (d->epilogue == 0 && d->activation == 0 && d->bias == nullptr), "pulse_gemm_x3p: split-K slabs carry no epilogue");
    PULSE_REQUIRE(d->rowsum == nullptr, "pulse_gemm_x3p: rowsum is not implemented in this build");

    XpArgs g;
    g.A = reinterpret_cast<const unsigned short*>(d->A); g.B = reinterpret_cast<const unsigned short*>(d->B);
    g.pa = d->a_plane_stride; g.pb = d->b_plane_stride; g.lda = d->lda; g.ldb = d->ldb;
    g.C = d->C; g.C2 = d->C2; g.Cp = reinterpret_cast<unsigned short*>(d->Cp); g.bias = d->bias;
    g.aux = d->aux_is_bf16 ? nullptr : d->aux;
    g.aux16 = d->aux_is_bf16 ? reinterpret_cast<const unsigned short*>(d->aux) : nullptr;
    g.pc = d->c_plane_stride; g.ldc = d->ldc; g.ldc2 = d->ldc2; g.ldcp = d->ldcp; g.ldaux = d->ldaux;
    g.M = d->M; g.N = d->N; g.K = d->K;
    g.sA = d->stride_a; g.sB = d->stride_b; g.sC = d->stride_c; g.sC2 = d->stride_c2; g.sCp = d->stride_cp; g.sBias = d->stride_bias; g.sAux = d->stride_aux;
    g.batch = d->batch; g.splitk = d->split_k;
    const bool big_ = xp_big_tiles(d->M, d->N, d->batch, d->split_k);
    const bool ring = npl == 1 && big_ && g_b16_ring;                // bf16 storage, 256-row tiles: the three-stage ring kernel
    const bool wide = ring && xp_wide_tiles(d->M, d->N, d->batch, d->split_k);     // ... or its 256 x 256 form
    const int kq = ring ? 2 * PK : npl == 1 ? 3 * PK : PK;           // split-K chunks are whole pipeline stages
    int kchunk = (d->K + d->split_k - 1) / d->split_k;
    kchunk = ((kchunk + kq - 1) / kq) * kq;
    g.kchunk = kchunk > 0 ? kchunk : kq;
    g.sSplit = d->split_stride;
    g.act = d->activation; g.epi = d->epilogue;
    g.rowsum = d->rowsum; g.sRowsum = d->stride_rowsum;
    g.colsum = d->out_colsum; g.sColsum = d->stride_out_colsum; g.ldcs = d->ld_out_colsum;
    g.mask8 = mask8_on ? d->relu_mask8 : nullptr; g.ldm8 = d->ld_mask8; g.sM8 = d->stride_mask8;
    g.general_rows = gemm_option(9);
    g.dbg = gemm_debug_buffer();
    PULSE_REQUIRE(!d->out_colsum || (d->split_k == 1 && d->ld_out_colsum >= d->N), "pulse_gemm_x3p: out_colsum needs split_k == 1 and a pitch covering N");
    const bool big = xp_big_tiles(d->M, d->N, d->batch, d->split_k);
    g.tiles_m = big ? (d->M + 255) / 256 : (d->M + 127) / 128;
    g.tiles_n = wide ? (d->N + B16wGeom::BN - 1) / B16wGeom::BN : (d->N + PBN - 1) / PBN;
    PULSE_REQUIRE((long long)d->lda * 300 < (1LL << 29) && (long long)d->ldb * 300 < (1LL << 29), "pulse_gemm_x3p: pitch too large for 32-bit tile-relative offsets");
    // [red][out] operands advance lda elements per k row: the whole k extent of a split must stay inside the 32-bit scalar offset
    PULSE_REQUIRE(akc || (long long)g.kchunk * d->lda * 2 < (1LL << 31), "pulse_gemm_x3p: split the reduction further (k extent x pitch exceeds 2 GiB)");
    PULSE_REQUIRE(bkc || (long long)g.kchunk * d->ldb * 2 < (1LL << 31), "pulse_gemm_x3p: split the reduction further (k extent x pitch exceeds 2 GiB)");
    const dim3 grid((unsigned)(g.tiles_m * g.tiles_n), (unsigned)(d->batch * d->split_k));
    hipError_t e = hipSuccess;
#define PULSE_XP_LAUNCH(AK, BK_, W, NP, SLOT)                                                                                                   \
    do {                                                                                                                                        \
        constexpr int lds = XpGeom<W>::LDS;                                                                                                     \
        static bool done = false;                                                                                                               \
        if (!done) {                                                                                                                            \
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x3p_kernel<AK, BK_, W, NP>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
            if (e != hipSuccess) return fail(PULSE_ERR_LAUNCH, "pulse_gemm_x3p: LDS attribute: %s", hipGetErrorString(e));                     \
            done = true;                                                                                                                        \
        }                                                                                                                                       \
        hipLaunchKernelGGL((gemm_x3p_kernel<AK, BK_, W, NP>), grid, dim3(128 * W), lds, as_stream(s), g);                                       \
    } while (0)
    if (npl == 3 && akc && bkc) {
        if (big) PULSE_XP_LAUNCH(true, true, 4, 3, 0); else PULSE_XP_LAUNCH(true, true, 2, 3, 1);
    } else if (npl == 3 && akc) {
        if (big) PULSE_XP_LAUNCH(true, false, 4, 3, 8); else PULSE_XP_LAUNCH(true, false, 2, 3, 9);
    } else if (npl == 3) {
        if (big) PULSE_XP_LAUNCH(false, false, 4, 3, 10); else PULSE_XP_LAUNCH(false, false, 2, 3, 11);
    } else if (wide) {
#define PULSE_B16W_LAUNCH(AK, BK_)                                                                                                              \
    do {                                                                                                                                        \
        constexpr int lds = XpGeom<4>::LDS;                                                                                                     \
        static bool done = false;                                                                                                               \
        if (!done) {                                                                                                                            \
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_b16w_kernel<AK, BK_>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
            if (e != hipSuccess) return fail(PULSE_ERR_LAUNCH, "pulse_gemm_x3p: LDS attribute: %s", hipGetErrorString(e));                     \
            done = true;                                                                                                                        \
        }                                                                                                                                       \
        hipLaunchKernelGGL((gemm_b16w_kernel<AK, BK_>), grid, dim3(512), lds, as_stream(s), g);                                                 \
    } while (0)
        if (akc && bkc) PULSE_B16W_LAUNCH(true, true); else if (akc) PULSE_B16W_LAUNCH(true, false); else PULSE_B16W_LAUNCH(false, false);
#undef PULSE_B16W_LAUNCH
    } else if (ring) {
#define PULSE_B16R_LAUNCH(AK, BK_)                                                                                                              \
    do {                                                                                                                                        \
        constexpr int lds = XpGeom<4>::LDS;                                                                                                     \
        static bool done = false;                                                                                                               \
        if (!done) {                                                                                                                            \
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_b16r_kernel<AK, BK_>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
            if (e != hipSuccess) return fail(PULSE_ERR_LAUNCH, "pulse_gemm_x3p: LDS attribute: %s", hipGetErrorString(e));                     \
            done = true;                                                                                                                        \
        }                                                                                                                                       \
        hipLaunchKernelGGL((gemm_b16r_kernel<AK, BK_>), grid, dim3(512), lds, as_stream(s), g);                                                 \
    } while (0)
        if (akc && bkc) PULSE_B16R_LAUNCH(true, true); else if (akc) PULSE_B16R_LAUNCH(true, false); else PULSE_B16R_LAUNCH(false, false);
#undef PULSE_B16R_LAUNCH
    } else if (akc && bkc) {
        if (big) PULSE_XP_LAUNCH(true, true, 4, 1, 2); else PULSE_XP_LAUNCH(true, true, 2, 1, 3);
    } else if (akc) {
        if (big) PULSE_XP_LAUNCH(true, false, 4, 1, 4); else PULSE_XP_LAUNCH(true, false, 2, 1, 5);
    } else {
        if (big) PULSE_XP_LAUNCH(false, false, 4, 1, 6); else PULSE_XP_LAUNCH(false, false, 2, 1, 7);
    }
#undef PULSE_XP_LAUNCH
    return check_launch("pulse_gemm_x3p");
}
}
