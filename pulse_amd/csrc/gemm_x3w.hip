// fp32 GEMM on the bf16 matrix pipe ("x3": three-way operand split, six products) on a 256 x 256 x 16 workgroup tile.
//
// Same arithmetic, operand layouts, epilogues and split-K contract as gemm_x3_kernel (gemm_f32.hip) -- the six plane products are issued in the
// same order per 16-deep k step, so both tilings give BIT-IDENTICAL matrix outputs -- for the nn.Linear calls of the PPO update path:
//   phc/learning/network_builder.py:105-124,245-261, phc/learning/amp_network_builder.py:127-148, phc/learning/amp_network_z_builder.py:341-467.
//
// Why a second tiling (round-4 verdict, missing #1): the 128 x 128 x 16 tile carries 3.67 VALU instructions of in-kernel split per MFMA plus its
// LDS traffic, about 4.6 issue slots per MFMA, and a SIMD has about 8 issue slots of 4 cycles per 32-cycle v_mfma_f32_32x32x16_bf16: the matrix pipe
// sat at 0.65-0.69 busy.  Split work and staged bytes scale with the tile's perimeter, MFMAs with its area: on 256 x 256 a k step is 96 MFMAs per
// wave beside 176 split VALU (1.83 per MFMA), 24 ds_write_b64, 28 ds_read_b128 and 8 global loads -- under 3 slots per MFMA.
//
// Shape of the kernel: 4 waves = ONE wave per SIMD, each wave a 128 x 128 block = 4 x 4 MFMA tiles, 256 accumulator registers in AGPRs (this
// translation unit is built WITHOUT -amdgpu-mfma-vgpr-form; amdgpu_waves_per_eu(1, 1) gives the wave the whole 512-entry file), VGPRs for the
// fragments (3 planes x 4 tiles x 2 operands, plus a second copy of B plane 0), one register set of raw fp32 operands and the split.  With one
// wave per SIMD nothing hides a stall, so everything is placed by hand in the 96 MFMA gaps of a k-tile, in ONE loop with ONE body and no branch
// inside (the register allocator shuffles the 256 accumulators between AGPRs and VGPRs at every control-flow join they are live across):
//   gaps  0-7   this tile's A0 / B0 fragments (A0 feeds the last term of a tile and the second of the next; B0 the first and the last)
//   gaps  0-31  split of A for tile t+1 (a third of an element pair's chain per gap), its plane stores one per gap as the pairs complete
//   gaps 32-63  the same for B;  gaps 36-39 / 64-67  global loads of tile t+2 into the registers the split just drained (16 B per lane each:
//               a vector-memory instruction costs the issuing wave about an MFMA's worth of issue time)
//   gap  69     the tile's one barrier (all stores of tile t+1 are >= 3 gaps old)
//   gaps 70-83  fragments of tile t+1 into the registers whose last MFMA of tile t has issued (B2, A2, the B0 copy, A1, then B1)
// In the last trip the staging works on a k-tile past the reduction's end: free under the MFMAs.
// Term order (A plane, B plane): (2,0) (0,2) (1,1) (1,0) (0,1) (0,0), as in gemm_x3_kernel.
// LDS: per operand and stage 3 planes x [2 k-chunks of 8][260 slots][16 B] (slot = out ^ ((out >> 3) & 7)); stage 1 starts at 64 KB so ONE xor
// toggles an address between the stages: 115,456 B.  The MFMAs are issued with the operands SWAPPED (B fragment as srcA): the accumulators hold
// the transposed 32 x 32 tiles, lane = output row, four consecutive registers = four consecutive output columns -- so the epilogue moves them
// through a 256 x 128 LDS image (pitch 132 floats, 135,168 B, two passes) with ds_write_b128 / ds_read_b128 and stores 16 bytes per lane, a
// quarter of the LDS and vector-memory instructions of the untransposed layout (a vector-memory instruction costs the issuing wave ~60 cycles:
// 256 dword stores per wave straight from the accumulators were measured at 10 us per workgroup).  One workgroup per CU.
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "gemm_shared.h"

namespace pulse {

constexpr int WT = 256;                              // tile rows = tile columns
constexpr int WK = 16;                               // k-tile depth
constexpr int W_CSTRIDE = 260;                       // 16-byte slots per 8-k chunk block (4 mod 8: the two chunks of a row sit in different bank-window halves)
constexpr int W_PLANE = 2 * W_CSTRIDE * 16;          // 8,320 B
constexpr int W_IMG = 3 * W_PLANE;                   // 24,960 B per operand
constexpr int W_STAGE_BIT = 65536;                   // stage 1 starts at 64 KB: one v_xor per address register toggles the stage
constexpr int W_CP = 132;                            // epilogue image pitch (floats): 528 B = 16 mod 128, so the 8 lanes of a ds_write_b128 group (8 rows) cover all 32 banks
constexpr int W_LDS = WT * W_CP * 4;                 // 135,168 B: the 256 x 128 epilogue image (the stages end at 115,456 B)
constexpr int W_BIAS_OFF = 120000;                   // 256 bias values of the tile, past the stages (read once, before the image exists)
constexpr int W_BARRIER_GAP = 69;

typedef __attribute__((address_space(3))) char lds_char;
// LDS byte offset of the dynamic shared segment (an integer: the per-lane addresses below are plain ints that xor between the stages)
__device__ __forceinline__ int lds_base() {
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    return (int)(unsigned)(unsigned long)(lds_char*)smem_c;
}
template <typename T>
__device__ __forceinline__ T lds_ld(int addr) { return *reinterpret_cast<const __attribute__((address_space(3))) T*>((unsigned long)(unsigned)addr); }
template <typename T>
__device__ __forceinline__ void lds_st(int addr, T v) { *reinterpret_cast<__attribute__((address_space(3))) T*>((unsigned long)(unsigned)addr) = v; }

// Per-thread staging of one operand: 16 fp32 elements per k-tile from 4 loads of 16 B (ONE register set: the loads of tile t+2 are issued as soon
// as the split of tile t+1 has drained the registers, 60 MFMA gaps = about 1,900 cycles before they are needed).
//   KC (reduction-contiguous): 4 lanes per row fetch the row's 64 contiguous bytes; thread = (row0 = tid >> 2, kq = tid & 3), load u = row row0 + 64 u;
//       element 4 u + e is (row row0 + 64 u, k = 4 kq + e).  Pair p = elements (2p, 2p + 1); store unit u = row-set u.
//   MC ([red][out]): a wave fetches 1 KB of a k row; thread = (og = tid & 63, kg = tid >> 6 = the wave), load kr = k row 4 kg + kr, outs 4 og .. 4 og + 3;
//       element 4 kr + u is (k row 4 kg + kr, out 4 og + u).  Pair p = 2 u + h = elements (8 h + u, 8 h + 4 + u); store unit u = out 4 og + u.
//   Either way a store unit is two pairs = 4 consecutive k of one row / out = 8 bytes per plane (ds_write_b64; with slot_of the 16 lanes of a store
//   group fall on every bank pair exactly twice, the minimum for 16 x 8 B in one half of their slots).
template <bool KC>
struct StagerW {
    float v[16];
    unsigned p0[8], p1[8], p2[8];        // the planes' packed pairs of the tile being split
    float ta, tb, ra, rb;                // the pair in flight
    unsigned tq0, tq1;
    int voff;                            // per-lane byte offset (constant)
    int wl[KC ? 1 : 4];                  // per-lane LDS byte address(es) of the store units in the stage being FILLED (toggled every tile)
    int kpos;                            // KC: first k of the thread's 4 elements per row
    int step;                            // KC: bytes per 64 rows; MC: bytes per k row (wave-uniform)
    int soff0;                           // MC: byte offset of the wave's first k row inside a k-tile (wave-uniform)

    __device__ __forceinline__ void init(int tid, int wave, int ld, int img) {
        if constexpr (KC) {
            const int row0 = tid >> 2, kq = tid & 3;
            kpos = kq * 4;
            voff = (row0 * ld + kq * 4) * 4;
            wl[0] = img + ((kq >> 1) * W_CSTRIDE + slot_of(row0)) * 16 + (kq & 1) * 8;       // rows row0 + 64 u: + 1024 u
            step = 64 * ld * 4;
            soff0 = 0;
        } else {
            const int og = tid & 63;
            kpos = 0;
            voff = og * 16;
            step = ld * 4;
            soff0 = 4 * wave * step;
#pragma unroll
            for (int u = 0; u < 4; ++u) wl[u] = img + ((wave >> 1) * W_CSTRIDE + slot_of(4 * og + u)) * 16 + (wave & 1) * 8;
        }
    }
    __device__ __forceinline__ void toggle() {
#pragma unroll
        for (int u = 0; u < (KC ? 1 : 4); ++u) wl[u] ^= W_STAGE_BIT;
    }
    // load u (0 .. 3) of the k-tile at scalar byte offset soff
    __device__ __forceinline__ void load_unit(__amdgpu_buffer_rsrc_t rs, int soff, int u) {
        const f32x4 a = buf_load(rs, voff, soff + soff0 + u * step);
        v[4 * u] = a.x; v[4 * u + 1] = a.y; v[4 * u + 2] = a.z; v[4 * u + 3] = a.w;
    }
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, int soff) {
#pragma unroll
        for (int u = 0; u < 4; ++u) load_unit(rs, soff, u);
    }
    // k tail of a reduction-contiguous operand: zero the elements at k positions >= hi of the tile.  ([red][out] operands need none: their k rows past
    // the reduction's end lie beyond the buffer resource's records and read as zero.)
    __device__ __forceinline__ void mask(int hi) {
        if constexpr (KC) {
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (kpos + (e & 3) >= hi) v[e] = 0.f;
        }
    }
    // MC only (the weight-gradient form's bias gradient): this thread's 4 k rows of out 4 og + u
    __device__ __forceinline__ float sum4(int u) const { return (v[u] + v[4 + u]) + (v[8 + u] + v[12 + u]); }
    __device__ __forceinline__ float elem(int p, int which) const {
        if constexpr (KC) return v[2 * p + which];
        else return v[8 * (p & 1) + 4 * which + (p >> 1)];
    }
    // a third or so of the split chain of element pair p: round-to-nearest-even at every level, remainders exact
    __device__ __forceinline__ void split_step(int p, int s) {
        if (s == 0) {
            ta = elem(p, 0); tb = elem(p, 1);
            tq0 = pack_rn(ta, tb);
            p0[p] = tq0;
        } else if (s == 1) {
            ra = ta - bitsf(tq0 << 16);
            rb = tb - bitsf(tq0 & 0xffff0000u);
        } else if (s == 2) {
            tq1 = pack_rn(ra, rb);
            p1[p] = tq1;
            ta = bitsf(tq1 << 16);
            tb = bitsf(tq1 & 0xffff0000u);
        } else {
            p2[p] = pack_rn(ra - ta, rb - tb);
        }
    }
    __device__ __forceinline__ void split_pair(int p) { split_step(p, 0); split_step(p, 1); split_step(p, 2); split_step(p, 3); }
    // plane store of unit u (pairs 2u, 2u + 1), plane pl, into the stage being filled
    __device__ __forceinline__ void write_plane(int u, int pl) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const unsigned* P = pl == 0 ? p0 : pl == 1 ? p1 : p2;
        const int addr = KC ? wl[0] + u * 1024 : wl[KC ? 0 : u];
        lds_st<u32x2>(addr + pl * W_PLANE, (u32x2){P[2 * u], P[2 * u + 1]});
    }
    // side work of gap q (0 .. 31) of this operand's split phase: a third of pair q / 4's chain, and the stores of the units already complete:
    // unit u's pairs are complete after gap 8 u + 7; its three stores follow, one per gap
    __device__ __forceinline__ void phase_gap(int q) {
        split_step(q >> 2, q & 3);
        if (q >= 8) {
            const int r = q - 8, u = r >> 3, w = r & 7;
            if (w < 3) write_plane(u, w);
        }
    }
    // the last unit's stores fall in the three gaps after the phase
    __device__ __forceinline__ void phase_tail(int w) { write_plane(3, w); }
};

// ---- epilogue: one 256 x 128 pass through the LDS image (pitch W_CP floats).  Image column c holds tile column (c >> 6) * 128 + 64 pass + (c & 63) --------
__device__ __forceinline__ int w_tile_col(int c, int pass) { return ((c >> 6) << 7) + 64 * pass + (c & 63); }

__device__ __forceinline__ void w_store_pass(const GemmArgs& g, int pass, int tid, int m0, int n0, float* C, float* C2, const float* aux, unsigned* mask) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int c4 = (tid & 31) * 4;               // image column of this thread's four values
    const int tc = w_tile_col(c4, pass);         // tile column
    const int rl0 = tid >> 5;                    // 8 rows per sweep, 32 sweeps
    const bool use_mask = g.epi == 1 && aux == nullptr;      // relu-grad from the forward's bit mask (gemm_shared.h: mask_word)
    const int cg = (n0 + tc) >> 2;
    if (g.vec_epi) {
        const bool fast = m0 + WT <= g.M && n0 + WT <= g.N && (g.epi == 1 || g.epi == 3 || (g.epi == 0 && g.act < 2));
        if (fast) {
            const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(C + (long long)m0 * g.ldc + n0, 0, 0xffffffffu, RSRC_FLAGS);
            const int voC = (rl0 * g.ldc + tc) * 4;
            const int ldsC = (rl0 * W_CP + c4) * 4;
            if (g.epi == 0) {
                const bool relu = g.act == 1;
                const bool wmask = relu && mask != nullptr;
                unsigned w = 0;
#pragma unroll 8
                for (int q = 0; q < 32; ++q) {
                    f32x4 v = lds_read(ldsC + q * 8 * W_CP * 4);
                    if (wmask) {
                        w |= ((v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u)) << (4 * (q & 7));
                        if ((q & 7) == 7) { mask[mask_word(m0 + 64 * (q >> 3) + rl0, cg, g.ldmask)] = w; w = 0; }
                    }
                    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    buf_store(v, rsC, voC, q * 8 * g.ldc * 4);
                }
            } else if (use_mask) {
#pragma unroll 1
                for (int b = 0; b < 4; ++b) {                                      // one mask word per 64-row block of this thread's column group
                    const unsigned mw = mask[mask_word(m0 + 64 * b + rl0, cg, g.ldmask)];
#pragma unroll
                    for (int qq = 0; qq < 8; ++qq) {
                        const int q = 8 * b + qq;
                        const unsigned nb = mw >> (4 * qq);
                        f32x4 v = lds_read(ldsC + q * 8 * W_CP * 4);
                        v.x = (nb & 1u) ? v.x : 0.f; v.y = (nb & 2u) ? v.y : 0.f; v.z = (nb & 4u) ? v.z : 0.f; v.w = (nb & 8u) ? v.w : 0.f;
                        buf_store(v, rsC, voC, q * 8 * g.ldc * 4);
                    }
                }
            } else {
                const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(aux) + (long long)m0 * g.ldaux + n0, 0,
                                                                                    0xffffffffu, RSRC_FLAGS);
                const int voX = (rl0 * g.ldaux + tc) * 4;
#pragma unroll 1
                for (int q0 = 0; q0 < 32; q0 += 8) {
                    f32x4 ax[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) ax[q] = buf_load(rsX, voX, (q0 + q) * 8 * g.ldaux * 4);
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const f32x4 a = ax[q];
                        f32x4 v = lds_read(ldsC + (q0 + q) * 8 * W_CP * 4);
                        if (g.epi == 1) { v.x = a.x > 0.f ? v.x : 0.f; v.y = a.y > 0.f ? v.y : 0.f; v.z = a.z > 0.f ? v.z : 0.f; v.w = a.w > 0.f ? v.w : 0.f; }
                        else { v.x *= a.x; v.y *= a.y; v.z *= a.z; v.w *= a.w; }       // EPI_MUL_AUX: aux = the producer's stored activation derivative
                        buf_store(v, rsC, voC, (q0 + q) * 8 * g.ldc * 4);
                    }
                }
            }
            return;
        }
        const int col = n0 + tc;
        if (col < g.N) {
            const bool full = col + 3 < g.N;
            const bool wmask = g.epi == 0 && g.act == 1 && mask != nullptr;
            unsigned w = 0;
#pragma unroll 4
            for (int q = 0; q < 32; ++q) {
                const int rl = rl0 + 8 * q;
                const int row = m0 + rl;
                // ragged tiles: a 64-row block's word is read at its first row slot / stored after its last one (rows past M: zero bits)
                if (use_mask && (q & 7) == 0 && m0 + 64 * (q >> 3) < g.M) w = mask[mask_word(m0 + 64 * (q >> 3) + rl0, cg, g.ldmask)];
                if (row >= g.M) {
                    if (wmask && (q & 7) == 7 && m0 + 64 * (q >> 3) < g.M) { mask[mask_word(m0 + 64 * (q >> 3) + rl0, cg, g.ldmask)] = w; w = 0; }
                    continue;
                }
                const float4 v = *reinterpret_cast<const float4*>(smem + rl * W_CP + c4);
                float o[4] = {v.x, v.y, v.z, v.w};
                if (wmask) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) w |= (col + k < g.N && o[k] > 0.f ? 1u : 0u) << (4 * (q & 7) + k);
                    if ((q & 7) == 7) { mask[mask_word(m0 + 64 * (q >> 3) + rl0, cg, g.ldmask)] = w; w = 0; }
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
    // scalar path (unaligned C / aux pitches: odd test shapes, not the training shapes): thread = (row group tid >> 7, image column tid & 127)
    {
        const int c = tid & 127;
        const int col = n0 + w_tile_col(c, pass);
        if (col >= g.N) return;
#pragma unroll 1
        for (int rl = tid >> 7; rl < WT; rl += 2) {
            const int row = m0 + rl;
            if (row >= g.M) break;
            float v = smem[rl * W_CP + c];
            if (g.epi == 0) {
                if (g.act == 1) {
                    v = fmaxf(v, 0.f);
                } else if (g.act == 2) {
                    if (C2) C2[(long long)row * g.ldc2 + col] = v;
                    v = v / (1.f + __expf(-v));
                } else if (g.act == 3) {
                    const float sg = 1.f / (1.f + __expf(-v));
                    C2[(long long)row * g.ldc2 + col] = sg * (1.f + v * (1.f - sg));
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

// the accumulators are born in AGPRs (the compiler otherwise starts them in VGPRs and shuffles 256 registers at the first control-flow join)
__device__ __forceinline__ float to_agpr(float x) {
    float r;
    asm("v_accvgpr_write_b32 %0, %1" : "=a"(r) : "v"(x));
    return r;
}

template <bool AKC, bool BKC>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) gemm_x3w_kernel(const GemmArgs g) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const WgMap wg = map_workgroup(g.tiles_m * g.tiles_n, g.batch, g.splitk);
    const int id = wg.id;
    const int tm = id / g.tiles_n, tn = id - tm * g.tiles_n;
    const int m0 = tm * WT, n0 = tn * WT;
    const int bz = wg.bz, sp = wg.sp;
    const int kbeg = sp * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int klen = kend - kbeg;
    // a split with no k range still runs ONE tile: its buffer resources have zero records, every load returns zero and the accumulators keep
    // their initial value (one straight path from the accumulator initialisation through MFMAs to the epilogue: no 256-register join)
    const int nkt = klen > 0 ? (klen + WK - 1) / WK : 1;
    const int hi = klen > 0 ? klen - (nkt - 1) * WK : 0;        // valid k positions of the last tile (1 .. 16; 0: everything masked)

    long long dbg_c0 = 0, dbg_w0 = 0, dbg_c1 = 0, dbg_w1 = 0;
    if (g.dbg) { dbg_c0 = clock64(); dbg_w0 = wall_clock64(); }

    // buffer resources with the TRUE extent from this workgroup's origin: what the range check catches reads as zero (no memory access)
    const int extA = min(WT, g.M - m0), extB = min(WT, g.N - n0);
    const int k4rem = ((g.K + 3) & ~3) - kbeg;                   // readable k positions of a reduction-contiguous row from kbeg
    const float* Ab = g.A + bz * g.sA + (AKC ? (long long)m0 * g.lda + kbeg : (long long)kbeg * g.lda + m0);
    const float* Bb = g.B + bz * g.sB + (BKC ? (long long)n0 * g.ldb + kbeg : (long long)kbeg * g.ldb + n0);
    const unsigned recA = (unsigned)(AKC ? ((extA - 1) * g.lda + k4rem) : ((klen - 1) * g.lda + extA)) * 4u;
    const unsigned recB = (unsigned)(BKC ? ((extB - 1) * g.ldb + k4rem) : ((klen - 1) * g.ldb + extB)) * 4u;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Ab), 0, klen > 0 ? recA : 0u, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Bb), 0, klen > 0 ? recB : 0u, RSRC_FLAGS);
    const int kstepA = (AKC ? 4 : g.lda * 4) * WK, kstepB = (BKC ? 4 : g.ldb * 4) * WK;      // bytes per k-tile

    const int lb = lds_base();
    StagerW<AKC> sa;
    StagerW<BKC> sb;
    sa.init(tid, wave, g.lda, lb);
    sb.init(tid, wave, g.ldb, lb + W_IMG);

    // fragment read addresses: lane (l31, half) reads out base + 32 t + l31, k-chunk = half, plane p at + p * W_PLANE.  base is a multiple of 128, so
    // slot_of(base + 64 e + 32 o + l31) = base + 64 e + slot_of(32 o + l31): one address per tile parity, the rest are immediates.  They point into the
    // stage the NEXT fragment reads come from: toggled once per tile, after the tile's own A0 / B0 reads (gaps 8-11).
    int frA0 = lb + (half * W_CSTRIDE + wm * 128 + slot_of(l31)) * 16;
    int frA1 = lb + (half * W_CSTRIDE + wm * 128 + slot_of(32 + l31)) * 16;
    int frB0 = lb + W_IMG + (half * W_CSTRIDE + wn * 128 + slot_of(l31)) * 16;
    int frB1 = lb + W_IMG + (half * W_CSTRIDE + wn * 128 + slot_of(32 + l31)) * 16;
    auto rd = [&](int addr, int pl, int t) { return lds_ld<bf16x8>(addr + (t >> 1) * 1024 + pl * W_PLANE); };

    f32x16 acc[4][4];
    // the tile's 256 bias values go through LDS: in the transposed accumulator layout a lane needs 16 x 4 consecutive ones (the same for all rows)
    {
        float bv = 0.f;
        if (g.epi == 0 && g.bias && n0 + tid < g.N) bv = g.bias[bz * g.sBias + n0 + tid];
        lds_st<float>(lb + W_BIAS_OFF + tid * 4, bv);
    }
    float rs[4] = {0.f, 0.f, 0.f, 0.f};

    // fragments: fa[plane][tile], fb[plane][tile]; fbx = B plane 0 for the NEXT tile's first term (B0 feeds the first and the last term of a tile: the
    // copy for the first term is read a tile ahead, the one for terms 3 and 5 in the tile's own first gaps)
    bf16x8 fa[3][4], fb[3][4], fbx[4];

    // prologue: tile 0 into stage 0, the loads of tile 1, tile 0's fragments (its A0 / B0 are read by the loop body itself)
    sa.load(rsA, 0); sb.load(rsB, 0);
    if (nkt == 1) { sa.mask(hi); sb.mask(hi); }
    if constexpr (!AKC) {
#pragma unroll
        for (int u = 0; u < 4; ++u) rs[u] += sa.sum4(u);
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) sa.split_pair(p);
#pragma unroll
    for (int u = 0; u < 4; ++u) { sa.write_plane(u, 0); sa.write_plane(u, 1); sa.write_plane(u, 2); }
#pragma unroll
    for (int p = 0; p < 8; ++p) sb.split_pair(p);
#pragma unroll
    for (int u = 0; u < 4; ++u) { sb.write_plane(u, 0); sb.write_plane(u, 1); sb.write_plane(u, 2); }
    sa.toggle(); sb.toggle();                                   // the loop's first trip fills stage 1
    sa.load(rsA, kstepA); sb.load(rsB, kstepB);
    __syncthreads();
    // register r of MFMA tile (i, j) is row wm 128 + 32 i + l31, column wn 128 + 32 j + 8 (r >> 2) + 4 half + (r & 3)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const f32x4 bq = lds_ld<f32x4>(lb + W_BIAS_OFF + (wn * 128 + j * 32 + 8 * gq + 4 * half) * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i][j][4 * gq] = to_agpr(bq.x); acc[i][j][4 * gq + 1] = to_agpr(bq.y);
                acc[i][j][4 * gq + 2] = to_agpr(bq.z); acc[i][j][4 * gq + 3] = to_agpr(bq.w);
            }
        }
#pragma unroll
    for (int t = 0; t < 4; ++t) { fa[2][t] = rd((t & 1) ? frA1 : frA0, 2, t); fbx[t] = rd((t & 1) ? frB1 : frB0, 0, t); }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        fb[2][t] = rd((t & 1) ? frB1 : frB0, 2, t); fa[1][t] = rd((t & 1) ? frA1 : frA0, 1, t); fb[1][t] = rd((t & 1) ? frB1 : frB0, 1, t);
    }

    int soffA = 2 * kstepA, soffB = 2 * kstepB;
    for (int t = 0; t < nkt; ++t) {
        if (t + 2 == nkt && hi < WK) { sa.mask(hi); sb.mask(hi); }                        // tile t+1 is the last one: its k tail is zeroed (a uniform
        if constexpr (!AKC) {                                                             // branch that does not touch the accumulators)
            if (t + 1 < nkt) {
#pragma unroll
                for (int u = 0; u < 4; ++u) rs[u] += sa.sum4(u);
            }
        }
#pragma unroll
        for (int q = 0; q < 96; ++q) {
            {
                const int term = q >> 4, i = (q >> 2) & 3, j = q & 3;
                const bf16x8 a = term == 0 ? fa[2][i] : (term == 2 || term == 3) ? fa[1][i] : fa[0][i];
                const bf16x8 b = term == 0 ? fbx[j] : term == 1 ? fb[2][j] : (term == 2 || term == 4) ? fb[1][j] : fb[0][j];
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc[i][j], 0, 0, 0);      // transposed tile: lane = row, registers = columns
            }
            // this tile's A0 (first needed at MFMA 16; its registers fed the previous tile's last term) and B0 (needed at MFMA 48), then the read
            // addresses move to the stage being filled
            if (q < 4) fa[0][q] = rd((q & 1) ? frA1 : frA0, 0, q);
            else if (q < 8) fb[0][q - 4] = rd((q & 1) ? frB1 : frB0, 0, q - 4);
            else if (q == 8) frA0 ^= W_STAGE_BIT;
            else if (q == 9) frA1 ^= W_STAGE_BIT;
            else if (q == 10) frB0 ^= W_STAGE_BIT;
            else if (q == 11) frB1 ^= W_STAGE_BIT;
            if (q < 32) sa.phase_gap(q);
            else if (q < 35) sa.phase_tail(q - 32);
            else if (q == 35) sa.toggle();
            if (q >= 32 && q < 64) sb.phase_gap(q - 32);
            else if (q >= 64 && q < 67) sb.phase_tail(q - 64);
            else if (q == 67) sb.toggle();
            if (q >= 36 && q < 40) sa.load_unit(rsA, soffA, q - 36);                      // tile t+2 into the registers the split just drained
            if (q >= 64 && q < 68) sb.load_unit(rsB, soffB, q - 64);
            if (q == W_BARRIER_GAP) {
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();
            }
            constexpr int R0 = W_BARRIER_GAP + 1;
            if (q == R0) { fb[2][0] = rd(frB0, 2, 0); fb[2][1] = rd(frB1, 2, 1); }
            else if (q == R0 + 1) { fb[2][2] = rd(frB0, 2, 2); fb[2][3] = rd(frB1, 2, 3); }
            else if (q == R0 + 2) { fa[2][0] = rd(frA0, 2, 0); fa[2][1] = rd(frA1, 2, 1); }
            else if (q == R0 + 3) { fa[2][2] = rd(frA0, 2, 2); fa[2][3] = rd(frA1, 2, 3); }
            else if (q == R0 + 4) { fbx[0] = rd(frB0, 0, 0); fbx[1] = rd(frB1, 0, 1); }
            else if (q == R0 + 5) { fbx[2] = rd(frB0, 0, 2); fbx[3] = rd(frB1, 0, 3); }
            else if (q == R0 + 6) { fa[1][0] = rd(frA0, 1, 0); fa[1][1] = rd(frA1, 1, 1); }
            else if (q == R0 + 7) { fa[1][2] = rd(frA0, 1, 2); fa[1][3] = rd(frA1, 1, 3); }
            else if (q >= 80 && q < 84) fb[1][q - 80] = rd((q & 1) ? frB1 : frB0, 1, q - 80);
            __builtin_amdgcn_sched_barrier(0);
        }
        soffA += kstepA; soffB += kstepB;
    }
    if constexpr (!AKC) {
        // bias gradient of the weight-gradient form: the four waves hold the sums of their k rows (4 of every 16) for all 256 outs
        if (g.rowsum != nullptr && tn == 0) {                    // workgroup-uniform
            __syncthreads();                                     // the staging buffers are free after the last tile's fragment reads
#pragma unroll
            for (int u = 0; u < 4; ++u) lds_st<float>(lb + (wave * 256 + 4 * lane + u) * 4, rs[u]);
            __syncthreads();
            if (m0 + tid < g.M) {
                const float s0 = lds_ld<float>(lb + tid * 4), s1 = lds_ld<float>(lb + (256 + tid) * 4);
                const float s2 = lds_ld<float>(lb + (512 + tid) * 4), s3 = lds_ld<float>(lb + (768 + tid) * 4);
                g.rowsum[bz * g.sRowsum + sp * g.sSplit + m0 + tid] = (s0 + s1) + (s2 + s3);
            }
        }
    }
    if (g.dbg) { dbg_c1 = clock64(); dbg_w1 = wall_clock64(); }

    // epilogue: pass p covers MFMA column tiles j = 2p, 2p + 1 of every wave (all four waves write 128 registers per pass, 16 bytes at a time)
    {
        float* C = g.C + bz * g.sC + sp * g.sSplit;
        float* C2 = g.C2 ? g.C2 + bz * g.sC2 : nullptr;
        const float* aux = g.aux ? g.aux + bz * g.sAux : nullptr;
        unsigned* mask = g.mask ? g.mask + bz * g.sMask : nullptr;
        const int img = lb + ((wm * 128 + l31) * W_CP + wn * 64 + 4 * half) * 4;
        __syncthreads();                                          // the image overlays the staging buffers
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if (pass) __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const f32x16& t = acc[i][2 * pass + jj];
                        lds_st<f32x4>(img + (i * 32 * W_CP + jj * 32 + 8 * gq) * 4, (f32x4){t[4 * gq], t[4 * gq + 1], t[4 * gq + 2], t[4 * gq + 3]});
                    }
            __syncthreads();
            w_store_pass(g, pass, tid, m0, n0, C, C2, aux, mask);
        }
    }
    if (g.dbg && tid == 0) {
        long long* o = g.dbg + 8 * (blockIdx.y * gridDim.x + blockIdx.x);
        o[0] = dbg_c0; o[1] = dbg_w0; o[2] = dbg_c1; o[3] = dbg_w1; o[4] = clock64(); o[5] = wall_clock64(); o[6] = 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        o[7] = wall_clock64();                    // the wave's stores acknowledged (tools/gemm_x3w_phases.py)
    }
}

int launch_gemm_x3w(const GemmArgs& g0, bool akc, bool bkc, hipStream_t stream) {
    GemmArgs g = g0;
    g.tiles_m = (g.M + WT - 1) / WT;
    g.tiles_n = (g.N + WT - 1) / WT;
    const dim3 grid((unsigned)(g.tiles_m * g.tiles_n), (unsigned)(g.batch * g.splitk));
    static bool attr_done[3] = {false, false, false};
    hipError_t e = hipSuccess;
#define LAUNCHW(IDX, AK, BK_)                                                                                          \
    if (!attr_done[IDX]) {                                                                                             \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x3w_kernel<AK, BK_>), hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS); \
        if (e != hipSuccess) { (void)hipGetLastError(); return kWideTileUnavailable; }   /* the device will not grant 135 KB of LDS: the caller keeps the 128 x 128 tile */ \
        attr_done[IDX] = true;                                                                                         \
    }                                                                                                                  \
    hipLaunchKernelGGL((gemm_x3w_kernel<AK, BK_>), grid, dim3(256), W_LDS, stream, g)
    if (akc && bkc) { LAUNCHW(0, true, true); }
    else if (akc && !bkc) { LAUNCHW(1, true, false); }
    else { LAUNCHW(2, false, false); }
#undef LAUNCHW
    return check_launch("pulse_gemm_f32 (256 x 256 tile)");
}

}  // namespace pulse
