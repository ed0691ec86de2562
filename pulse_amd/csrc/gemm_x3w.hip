// fp32 GEMM on the bf16 matrix pipe ("x3": three-way operand split, six products) on a 256 x 256 x 16 workgroup tile.
//
// Same arithmetic, operand layouts, epilogues and split-K contract as gemm_x3_kernel (gemm_f32.hip) -- the six plane products are issued in the
// same order per 16-deep k step, so both tilings give BIT-IDENTICAL outputs -- for the nn.Linear calls of the PPO update path:
//   phc/learning/network_builder.py:105-124,245-261, phc/learning/amp_network_builder.py:127-148, phc/learning/amp_network_z_builder.py:341-467.
//
// Why a second tiling (round-4 verdict, missing #1): the 128 x 128 x 16 tile carries 3.67 VALU instructions of in-kernel split per MFMA plus its
// LDS traffic, about 4.6 issue slots per MFMA, and a SIMD has about 8 issue slots of 4 cycles per 32-cycle v_mfma_f32_32x32x16_bf16: the matrix pipe
// sat at 0.65-0.69 busy.  Split work and staged bytes scale with the tile's perimeter, MFMAs with its area: on 256 x 256 a k step is 96 MFMAs per
// wave beside 176 split VALU (1.83 per MFMA), 12 ds_write, 24 ds_read_b128 and 8 (or 32) global loads -- under 3 slots per MFMA.
//
// Shape of the kernel: 4 waves = ONE wave per SIMD, each wave a 128 x 128 block = 4 x 4 MFMA tiles, 256 accumulator registers in AGPRs (this
// translation unit is built WITHOUT -amdgpu-mfma-vgpr-form; amdgpu_waves_per_eu(1, 1) gives the wave the whole 512-entry file), 256 VGPRs for
// fragments (3 planes x 4 tiles x 2 operands = 96, B plane 0 double-buffered), two register sets of raw fp32 operands (loads run two k-tiles
// ahead) and the split.  With one wave per SIMD nothing hides a stall, so everything is placed by hand in the 96 MFMA gaps of a k-tile:
//   gaps  0-31  split of A for tile t+1 (one third of an element pair's chain per gap), its plane stores one per gap as the pairs complete
//   gaps 32-63  the same for B;  gaps 36-39 / 64-67  global loads of tile t+3 into the register set the split just drained
//   gap  69     the tile's one barrier (all stores of tile t+1 are >= 3 gaps old)
//   gaps 70-83  fragments of tile t+1 into the registers whose last MFMA of tile t has issued (B2, A2, A1, the other B0 set, then B1)
//   gaps  0-3   (of tile t+1) its A0 fragments -- A0 feeds the last term of a tile and the second of the next
// Term order (A plane, B plane): (2,0) (0,2) (1,1) (1,0) (0,1) (0,0), as in gemm_x3_kernel.
// LDS: per operand and stage 3 planes x [2 k-chunks of 8][260 slots][16 B] (slot = out ^ ((out >> 3) & 7)), two stages = 99,840 B; the epilogue
// goes through a 256 x 128 fp32 image (131,072 B) in two passes.  One workgroup per CU.
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
constexpr int W_STAGE = 2 * W_IMG;                   // 49,920 B
constexpr int W_STAGE_BIT = 65536;                   // stage 1 starts at 64 KB: one v_xor per address register toggles the stage
constexpr int W_CP = 128;                            // epilogue image pitch (floats)
constexpr int W_LDS = WT * W_CP * 4;                 // 131,072 B (>= stage 1's end at 65,536 + 49,920)
constexpr int W_BARRIER_GAP = 69;

// Per-thread staging of one operand: 16 fp32 elements per k-tile (ONE register set: the loads of tile t+2 are issued as soon as the split of
// tile t+1 has drained the registers, 60 MFMA gaps = about 1,900 cycles before they are needed).
//   KC (reduction-contiguous): 4 lanes per row fetch the row's 64 contiguous bytes; thread = (row0 = tid >> 2, kq = tid & 3), 4 loads of 16 B for rows
//       row0 + 64 u; element 4 u + e is (row row0 + 64 u, k = 4 kq + e).  A row-set's two element pairs become 8 bytes per plane (ds_write_b64).
//   MC ([red][out]): lane = out (tid), 16 dword loads, element kr is (k row kr, out).  A chunk of 8 k rows becomes 16 bytes per plane (ds_write_b128).
template <bool KC>
struct StagerW {
    float v[16];
    unsigned p0[8], p1[8], p2[8];        // the planes' packed pairs of the tile being split
    float ta, tb, ra, rb;                // the pair in flight
    unsigned tq0, tq1;
    int voff;                            // per-lane byte offset (constant)
    int lds;                             // per-lane LDS byte address inside the operand's image; the stage bit (W_STAGE_BIT) toggles every tile
    int kpos;                            // KC: first k of the thread's 4 elements per row
    int step;                            // KC: bytes per 64 rows; MC: bytes per k row (wave-uniform)

    __device__ __forceinline__ void init(int tid, int ld, int img_off) {
        if constexpr (KC) {
            const int row0 = tid >> 2, kq = tid & 3;
            kpos = kq * 4;
            voff = (row0 * ld + kq * 4) * 4;
            lds = img_off + ((kq >> 1) * W_CSTRIDE + slot_of(row0)) * 16 + (kq & 1) * 8;     // rows row0 + 64 u: slot + 64 u
            step = 64 * ld * 4;
        } else {
            kpos = 0;
            voff = tid * 4;
            lds = img_off + slot_of(tid) * 16;
            step = ld * 4;
        }
    }
    // load unit u of the k-tile at scalar byte offset soff: KC 4 units (one per row-set), MC 4 units of 4 k rows
    __device__ __forceinline__ void load_unit(__amdgpu_buffer_rsrc_t rs, int soff, int u) {
        if constexpr (KC) {
            const f32x4 a = buf_load(rs, voff, soff + u * step);
            v[4 * u] = a.x; v[4 * u + 1] = a.y; v[4 * u + 2] = a.z; v[4 * u + 3] = a.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[4 * u + i] = bitsf(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff + (4 * u + i) * step, 0));
        }
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
    // MC only (the weight-gradient form's bias gradient): sums of the two 8-k chunks, associated like gemm_x3_kernel's
    __device__ __forceinline__ float sum8(int c) const {
        const float* x = &v[8 * c];
        return ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
    }
    // one third-ish of the split chain of element pair p (elements 2p, 2p+1): round-to-nearest-even at every level, remainders exact
    __device__ __forceinline__ void split_step(int p, int s) {
        if (s == 0) {
            ta = v[2 * p]; tb = v[2 * p + 1];
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
    // plane stores.  KC: unit u (row-set), MC: unit c (k-chunk); pl = plane; st = this operand's per-lane address with the stage bit applied
    static constexpr int NWRITE = KC ? 4 : 2;        // store units per tile (x 3 planes)
    static constexpr int PAIRS_PER_UNIT = KC ? 2 : 4;
    __device__ __forceinline__ void write_plane(int addr, int u, int pl) {
        extern __shared__ __attribute__((aligned(16))) char smem_c[];
        const unsigned* P = pl == 0 ? p0 : pl == 1 ? p1 : p2;
        if constexpr (KC) {
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            *reinterpret_cast<u32x2*>(smem_c + addr + u * 1024 + pl * W_PLANE) = (u32x2){P[2 * u], P[2 * u + 1]};
        } else {
            *reinterpret_cast<u32x4*>(smem_c + addr + u * (W_CSTRIDE * 16) + pl * W_PLANE) = (u32x4){P[4 * u], P[4 * u + 1], P[4 * u + 2], P[4 * u + 3]};
        }
    }
    // side work of gap q (0 .. 31) of this operand's split phase: a third of pair q / 4's chain, and the stores of the units already complete
    __device__ __forceinline__ void phase_gap(int addr, int q) {
        split_step(q >> 2, q & 3);
        // unit u's pairs are complete after gap 4 PAIRS_PER_UNIT (u + 1) - 1; its three stores follow, one per gap
        constexpr int G = 4 * PAIRS_PER_UNIT;
        if (q >= G) {
            const int r = q - G, u = r / G, w = r % G;
            if (w < 3) write_plane(addr, u, w);
        }
    }
    // the last unit's stores fall in the three gaps after the phase
    __device__ __forceinline__ void phase_tail(int addr, int w) { write_plane(addr, NWRITE - 1, w); }
};

// ---- epilogue: one 256 x 128 pass through the LDS image.  Image column c holds tile column (c >> 6) * 128 + 64 pass + (c & 63) -----------------------------
__device__ __forceinline__ int w_tile_col(int c, int pass) { return ((c >> 6) << 7) + 64 * pass + (c & 63); }

__device__ __forceinline__ void w_store_pass(const GemmArgs& g, int pass, int tid, int m0, int n0, float* C, float* C2, const float* aux) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int c4 = (tid & 31) * 4;               // image column of this thread's four values
    const int tc = w_tile_col(c4, pass);         // tile column
    const int rl0 = tid >> 5;                    // 8 rows per sweep, 32 sweeps
    if (g.vec_epi) {
        const bool fast = m0 + WT <= g.M && n0 + WT <= g.N && (g.epi == 1 || (g.epi == 0 && g.act != 2));
        if (fast) {
            const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(C + (long long)m0 * g.ldc + n0, 0, 0xffffffffu, RSRC_FLAGS);
            const int voC = (rl0 * g.ldc + tc) * 4;
            const int ldsC = (rl0 * W_CP + c4) * 4;
            if (g.epi == 0) {
                const bool relu = g.act == 1;
#pragma unroll 8
                for (int q = 0; q < 32; ++q) {
                    f32x4 v = lds_read(ldsC + q * 8 * W_CP * 4);
                    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    buf_store(v, rsC, voC, q * 8 * g.ldc * 4);
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
                        v.x = a.x > 0.f ? v.x : 0.f; v.y = a.y > 0.f ? v.y : 0.f; v.z = a.z > 0.f ? v.z : 0.f; v.w = a.w > 0.f ? v.w : 0.f;
                        buf_store(v, rsC, voC, (q0 + q) * 8 * g.ldc * 4);
                    }
                }
            }
            return;
        }
        const int col = n0 + tc;
        if (col < g.N) {
            const bool full = col + 3 < g.N;
#pragma unroll 4
            for (int q = 0; q < 32; ++q) {
                const int rl = rl0 + 8 * q;
                const int row = m0 + rl;
                if (row >= g.M) continue;
                const float4 v = *reinterpret_cast<const float4*>(smem + rl * W_CP + c4);
                float o[4] = {v.x, v.y, v.z, v.w};
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

    StagerW<AKC> sa;
    StagerW<BKC> sb;
    sa.init(tid, g.lda, 0);
    sb.init(tid, g.ldb, W_IMG);

    // fragment read addresses: lane (l31, half) reads out base + 32 t + l31, k-chunk = half, plane p at + p * W_PLANE.  base is a multiple of 128, so
    // slot_of(base + 64 e + 32 o + l31) = base + 64 e + slot_of(32 o + l31): one address per tile parity, the rest are immediates.  These are the
    // addresses in the CURRENT tile's stage; ^ W_STAGE_BIT is the stage being filled.
    int frA0 = (half * W_CSTRIDE + wm * 128 + slot_of(l31)) * 16;
    int frA1 = (half * W_CSTRIDE + wm * 128 + slot_of(32 + l31)) * 16;
    int frB0 = W_IMG + (half * W_CSTRIDE + wn * 128 + slot_of(l31)) * 16;
    int frB1 = W_IMG + (half * W_CSTRIDE + wn * 128 + slot_of(32 + l31)) * 16;
    auto rd = [&](int addr, int pl, int t) {
        extern __shared__ __attribute__((aligned(16))) char smem_c[];
        return *reinterpret_cast<const bf16x8*>(smem_c + addr + (t >> 1) * 1024 + pl * W_PLANE);
    };

    f32x16 acc[4][4];
    {
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (g.epi == 0 && g.bias) {
            const float* bias = g.bias + bz * g.sBias;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = n0 + wn * 128 + j * 32 + l31;
                if (c < g.N) bv[j] = bias[c];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = to_agpr(bv[j]);
    }
    float rs0 = 0.f, rs1 = 0.f;

    // fragments: fa[plane][tile], fb[plane][tile]; fbx = B plane 0 of the NEXT tile's first term (B0 feeds the first and the last term of a tile: the
    // copy for the first term is read a tile ahead, the one for terms 3 and 5 in the tile's own first gaps)
    bf16x8 fa[3][4], fb[3][4], fbx[4];

    // prologue: tile 0 into stage 0, the loads of tile 1, tile 0's fragments (its A0 / B0 are read by the loop body itself)
    sa.load(rsA, 0); sb.load(rsB, 0);
    sa.mask(nkt == 1 ? hi : WK); sb.mask(nkt == 1 ? hi : WK);
    if constexpr (!AKC) { rs0 += sa.sum8(0); rs1 += sa.sum8(1); }
#pragma unroll
    for (int p = 0; p < 8; ++p) sa.split_pair(p);
#pragma unroll
    for (int u = 0; u < StagerW<AKC>::NWRITE; ++u) { sa.write_plane(sa.lds, u, 0); sa.write_plane(sa.lds, u, 1); sa.write_plane(sa.lds, u, 2); }
#pragma unroll
    for (int p = 0; p < 8; ++p) sb.split_pair(p);
#pragma unroll
    for (int u = 0; u < StagerW<BKC>::NWRITE; ++u) { sb.write_plane(sb.lds, u, 0); sb.write_plane(sb.lds, u, 1); sb.write_plane(sb.lds, u, 2); }
    sa.load(rsA, kstepA); sb.load(rsB, kstepB);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t) { fa[2][t] = rd((t & 1) ? frA1 : frA0, 2, t); fbx[t] = rd((t & 1) ? frB1 : frB0, 0, t); }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        fb[2][t] = rd((t & 1) ? frB1 : frB0, 2, t); fa[1][t] = rd((t & 1) ? frA1 : frA0, 1, t); fb[1][t] = rd((t & 1) ? frB1 : frB0, 1, t);
    }

    // ONE loop, one body, no branch inside and no tail code: tile t's 96 MFMAs with the staging of tile t+1 in their gaps.  In the last trip the
    // staging works on a k-tile past the reduction's end (loads beyond the records read zero, or read operand memory that is never used) into the
    // stage nobody reads again -- free under the MFMAs, and the accumulators see a single back edge (with tails and parity copies the register
    // allocator moved them between AGPRs and VGPRs at every join and spilled 4,000 registers).
    int soffA = 2 * kstepA, soffB = 2 * kstepB;
    for (int t = 0; t < nkt; ++t) {
        const int wrA = sa.lds ^ W_STAGE_BIT, wrB = sb.lds ^ W_STAGE_BIT;                 // tile t+1's stage
        const int nA0 = frA0 ^ W_STAGE_BIT, nA1 = frA1 ^ W_STAGE_BIT, nB0 = frB0 ^ W_STAGE_BIT, nB1 = frB1 ^ W_STAGE_BIT;
        const int hin = (t + 2 == nkt) ? hi : WK;                                         // tile t+1 is the last one: its k tail is zeroed
        if (hin < WK) { sa.mask(hin); sb.mask(hin); }                                     // (a uniform branch that does not touch the accumulators)
        if constexpr (!AKC) {
            if (t + 1 < nkt) { rs0 += sa.sum8(0); rs1 += sa.sum8(1); }
        }
#pragma unroll
        for (int q = 0; q < 96; ++q) {
            {
                const int term = q >> 4, i = (q >> 2) & 3, j = q & 3;
                const bf16x8 a = term == 0 ? fa[2][i] : (term == 2 || term == 3) ? fa[1][i] : fa[0][i];
                const bf16x8 b = term == 0 ? fbx[j] : term == 1 ? fb[2][j] : (term == 2 || term == 4) ? fb[1][j] : fb[0][j];
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i][j], 0, 0, 0);
            }
            // this tile's A0 (first needed at MFMA 16; its registers fed the previous tile's last term) and B0 (needed at MFMA 48)
            if (q < 4) fa[0][q] = rd((q & 1) ? frA1 : frA0, 0, q);
            else if (q < 8) fb[0][q - 4] = rd((q & 1) ? frB1 : frB0, 0, q - 4);
            if (q < 32) sa.phase_gap(wrA, q);
            else if (q < 35) sa.phase_tail(wrA, q - 32);
            if (q >= 32 && q < 64) sb.phase_gap(wrB, q - 32);
            else if (q >= 64 && q < 67) sb.phase_tail(wrB, q - 64);
            if (q >= 36 && q < 40) sa.load_unit(rsA, soffA, q - 36);                      // tile t+2 into the registers the split just drained
            if (q >= 64 && q < 68) sb.load_unit(rsB, soffB, q - 64);
            if (q == W_BARRIER_GAP) {
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();
            }
            constexpr int R0 = W_BARRIER_GAP + 1;
            if (q == R0) { fb[2][0] = rd(nB0, 2, 0); fb[2][1] = rd(nB1, 2, 1); }
            else if (q == R0 + 1) { fb[2][2] = rd(nB0, 2, 2); fb[2][3] = rd(nB1, 2, 3); }
            else if (q == R0 + 2) { fa[2][0] = rd(nA0, 2, 0); fa[2][1] = rd(nA1, 2, 1); }
            else if (q == R0 + 3) { fa[2][2] = rd(nA0, 2, 2); fa[2][3] = rd(nA1, 2, 3); }
            else if (q == R0 + 4) { fbx[0] = rd(nB0, 0, 0); fbx[1] = rd(nB1, 0, 1); }
            else if (q == R0 + 5) { fbx[2] = rd(nB0, 0, 2); fbx[3] = rd(nB1, 0, 3); }
            else if (q == R0 + 6) { fa[1][0] = rd(nA0, 1, 0); fa[1][1] = rd(nA1, 1, 1); }
            else if (q == R0 + 7) { fa[1][2] = rd(nA0, 1, 2); fa[1][3] = rd(nA1, 1, 3); }
            else if (q >= 80 && q < 84) fb[1][q - 80] = rd((q & 1) ? nB1 : nB0, 1, q - 80);
            __builtin_amdgcn_sched_barrier(0);
        }
        sa.lds = wrA; sb.lds = wrB;
        frA0 = nA0; frA1 = nA1; frB0 = nB0; frB1 = nB1;
        soffA += kstepA; soffB += kstepB;
    }
    __syncthreads();                                              // the epilogue reuses the staging buffers
    if constexpr (!AKC) {
        if (g.rowsum != nullptr && tn == 0 && m0 + tid < g.M) g.rowsum[bz * g.sRowsum + sp * g.sSplit + m0 + tid] = rs0 + rs1;      // thread = out: it summed all 16 k rows of every tile
    }
    if (g.dbg) { dbg_c1 = clock64(); dbg_w1 = wall_clock64(); }

    // epilogue: pass p covers MFMA column tiles j = 2p, 2p + 1 of every wave (all four waves write 128 registers per pass)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* C = g.C + bz * g.sC + sp * g.sSplit;
    float* C2 = g.C2 ? g.C2 + bz * g.sC2 : nullptr;
    const float* aux = g.aux ? g.aux + bz * g.sAux : nullptr;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (pass) __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    smem[(wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * W_CP + wn * 64 + jj * 32 + l31] = acc[i][2 * pass + jj][r];
        __syncthreads();
        if (g.dbg && tid == 0 && pass == 0) g.dbg[8 * (blockIdx.y * gridDim.x + blockIdx.x) + 6] = clock64();
        w_store_pass(g, pass, tid, m0, n0, C, C2, aux);
    }
    if (g.dbg && tid == 0) {
        long long* o = g.dbg + 8 * (blockIdx.y * gridDim.x + blockIdx.x);
        o[0] = dbg_c0; o[1] = dbg_w0; o[2] = dbg_c1; o[3] = dbg_w1; o[4] = clock64(); o[5] = wall_clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        o[7] = wall_clock64();                    // the workgroup's stores acknowledged (tools/gemm_x3w_phases.py)
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
        if (e != hipSuccess) return fail(PULSE_ERR_LAUNCH, "pulse_gemm_f32 (256 x 256 tile): LDS attribute: %s", hipGetErrorString(e)); \
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
