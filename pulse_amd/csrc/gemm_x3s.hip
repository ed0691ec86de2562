// The x3 fp32 GEMM for SKINNY outputs (N <= 96 columns over a long M): the mu / value heads of the actor / critic pair and the latent-width
// layers of the PULSE VAE  (nn.Linear forward / input gradient: phc/learning/amp_network_builder.py:127-148 ``mu`` / ``value``,
// phc/learning/amp_network_z_builder.py:341-467 ``z_mu`` / ``z_logvar`` and the decoder's dX towards z).
//
// Why a kernel of its own.  On the 128 x 128 (64 x 128) tiling such a launch pads N to 128 columns and, per 64 rows of A, re-splits and
// re-stages a whole 128 x 16 B tile every k step: per k-tile and SIMD 2 waves x (12 MFMAs + ~120 VALU) -- the tile is bound by instruction
// issue at ~0.09 of the x3 peak (profiles/r06_ab_runs.txt), while the operand it streams (16384 x 512 fp32 x 2 nets = 67 MB) would take
// 11 us at HBM rate.  Here:
//   * a workgroup owns 128 rows x ALL N columns (three 32-column MFMA tiles), 4 waves = one per SIMD, wave w owns rows 32 w .. 32 w + 31;
//   * A never touches LDS: the lane that feeds row r, k group h of v_mfma_f32_32x32x16_bf16 loads exactly those 8 consecutive fp32 (two
//     16-byte loads, a full 128-byte line per row and k-step pair across the two k groups), splits them in registers (the same
//     round-to-nearest-even three-way split, gemm_shared.h) and uses them as the fragment -- loads run a whole B phase (8 k steps) ahead;
//   * B (N x K, either layout) is staged cooperatively, once per workgroup and 128-deep k phase, ALREADY SPLIT and in fragment order
//     (k step, column tile, plane -> 64 lanes x 16 B: one conflict-free ds_read_b128 per fragment), double buffered (2 x 72 KB): the split of
//     phase p + 1 is spread over the k steps of phase p, one barrier per phase;
//   * per k step and wave: 18 MFMAs beside 44 VALU of A split + 33 of B split + 9 ds_read_b128.
// Same arithmetic as gemm_x3_kernel / gemm_x3w_kernel: per 16-deep k step the six plane products (A plane, B plane) = (2,0) (0,2) (1,1) (1,0)
// (0,1) (0,0) into the same fp32 accumulator, bias as its initial value -- BIT-IDENTICAL outputs (tests/test_gemm_x3_skinny_gpu.py).
#include <type_traits>
#include "gemm_shared.h"

namespace pulse {

constexpr int S_ROWS = 128;                                  // rows per workgroup
constexpr int S_NT = 3;                                      // 32-column MFMA tiles: N <= 96
constexpr int S_KP = 128;                                    // k extent of a B phase
constexpr int S_STEPS = S_KP / 16;                           // k steps per phase
constexpr int S_FRAG = 64 * 16;                              // one fragment block: 64 lanes x 16 B
constexpr int S_STAGE = S_STEPS * S_NT * 3 * S_FRAG;         // 73,728 B
constexpr int S_LDS = 2 * S_STAGE;                           // 147,456 B: one workgroup per CU
constexpr int S_UNITS = S_NT * 32 * (S_KP / 8) / 256;        // B staging units (one column x 8 k) per thread and phase: 6

// raw fp32 of 8 consecutive k -> the three bf16x8 plane fragments
struct Split8 {
    u32x4 p0, p1, p2;
    __device__ __forceinline__ void pair(const float (&v)[8], int k) {      // elements 2k, 2k + 1: 11 VALU
        const float a = v[2 * k], b = v[2 * k + 1];
        const unsigned q0 = pack_rn(a, b);
        p0[k] = q0;
        const float ra = a - bitsf(q0 << 16), rb = b - bitsf(q0 & 0xffff0000u);
        const unsigned q1 = pack_rn(ra, rb);
        p1[k] = q1;
        const float sa = ra - bitsf(q1 << 16), sb = rb - bitsf(q1 & 0xffff0000u);
        p2[k] = pack_rn(sa, sb);
    }
    __device__ __forceinline__ void run(const float (&v)[8]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) pair(v, k);
    }
};

// EXP: timing experiments only (wrong results): 1 no MFMAs, 2 no A split, 4 no B staging, 8 no fragment reads, 16 no A loads after the prologue
template <bool BKC, int EXP = 0>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) gemm_x3s_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.x * S_ROWS;
    const int bz = blockIdx.y;
    const int K = g.K;
    const int nph = (K + S_KP - 1) / S_KP;

    // ---- operands: buffer resources with the true extents (what the range check catches reads as zero without touching memory)
    const int extA = min(S_ROWS, g.M - m0);
    const int k4 = (K + 3) & ~3;
    const float* Ab = g.A + bz * g.sA + (long long)m0 * g.lda;
    const float* Bb = g.B + bz * g.sB;
    const unsigned recA = (unsigned)((extA - 1) * g.lda + k4) * 4u;
    const unsigned recB = (unsigned)(BKC ? ((g.N - 1) * g.ldb + k4) : ((K - 1) * g.ldb + g.N)) * 4u;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Ab), 0, recA, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Bb), 0, recB, RSRC_FLAGS);

    // A: this lane feeds row 32 wave + l31, k group ``half`` of every k step
    const int voA = ((wave * 32 + l31) * g.lda + 8 * half) * 4;
    float a[S_STEPS][8];                                     // one phase of raw A (8 k steps x 8 floats), reloaded a phase ahead
    auto load_a = [&](int s, int kbase) {                    // k step s of the phase starting at kbase
        const int so = (kbase + 16 * s) * 4;
        const f32x4 x = buf_load(rsA, voA, so), y = buf_load(rsA, voA, so + 16);
        a[s][0] = x.x; a[s][1] = x.y; a[s][2] = x.z; a[s][3] = x.w; a[s][4] = y.x; a[s][5] = y.y; a[s][6] = y.z; a[s][7] = y.w;
    };

    // B staging: unit u of a thread = column n_u, k group kg_u (8 k) of the phase; its three plane fragments go to
    // [k step kg >> 1][column tile n >> 5][plane][lane (kg & 1) * 32 + (n & 31)]
    float braw[S_UNITS][8];
    int b_vo[S_UNITS], b_lds[S_UNITS], b_kpos[S_UNITS];
#pragma unroll
    for (int u = 0; u < S_UNITS; ++u) {
        const int id = tid + 256 * u;                        // 0 .. 1535 = 96 columns x 16 k groups
        const int n = BKC ? id / 16 : id % 96, kg = BKC ? id % 16 : id / 96;      // consecutive lanes walk the operand's contiguous dimension
        b_vo[u] = (BKC ? n * g.ldb + 8 * kg : 8 * kg * g.ldb + n) * 4;
        b_lds[u] = (((kg >> 1) * S_NT + (n >> 5)) * 3) * S_FRAG + ((kg & 1) * 32 + (n & 31)) * 16;
        b_kpos[u] = 8 * kg;
    }
    auto load_b = [&](int u, int kbase) {
        if constexpr (BKC) {
            const f32x4 x = buf_load(rsB, b_vo[u], kbase * 4), y = buf_load(rsB, b_vo[u], kbase * 4 + 16);
            braw[u][0] = x.x; braw[u][1] = x.y; braw[u][2] = x.z; braw[u][3] = x.w; braw[u][4] = y.x; braw[u][5] = y.y; braw[u][6] = y.z; braw[u][7] = y.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) braw[u][e] = bitsf(__builtin_amdgcn_raw_buffer_load_b32(rsB, b_vo[u], (kbase + e) * g.ldb * 4, 0));
        }
    };
    auto stage_b = [&](auto tail_tag, int u, int kbase, int stage) {     // split unit u (raw values of the phase at kbase) into LDS stage ``stage``
        if constexpr (BKC && decltype(tail_tag)::value) {    // the reduction ends inside that phase: reduction-contiguous rows may carry anything
#pragma unroll                                               // past K inside their pitch
            for (int e = 0; e < 8; ++e)
                if (kbase + b_kpos[u] + e >= K) braw[u][e] = 0.f;
        }
        Split8 sp;
        sp.run(braw[u]);
        char* dst = smem_c + stage * S_STAGE + b_lds[u];
        *reinterpret_cast<u32x4*>(dst) = sp.p0;
        *reinterpret_cast<u32x4*>(dst + S_FRAG) = sp.p1;
        *reinterpret_cast<u32x4*>(dst + 2 * S_FRAG) = sp.p2;
    };

    // accumulators: tile j = columns 32 j + l31, register r = row (r & 3) + 8 (r >> 2) + 4 half; the bias is their initial value
    f32x16 acc[S_NT];
#pragma unroll
    for (int j = 0; j < S_NT; ++j) {
        float b0 = 0.f;
        if (g.bias && 32 * j + l31 < g.N) b0 = g.bias[bz * g.sBias + 32 * j + l31];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = b0;
    }

    // ---- prologue: phase 0's A and B, B split into stage 0; phase 1's B raw values on their way
#pragma unroll
    for (int s = 0; s < S_STEPS; ++s) load_a(s, 0);
#pragma unroll
    for (int u = 0; u < S_UNITS; ++u) load_b(u, 0);
#pragma unroll
    for (int u = 0; u < S_UNITS; ++u) stage_b(std::true_type{}, u, 0, 0);
    if (nph > 1) {
#pragma unroll
        for (int u = 0; u < S_UNITS; ++u) load_b(u, S_KP);
    }
    __syncthreads();

    // One phase = 8 k steps in ONE basic block (no branch inside: the scheduler can put the split of k step s + 1 and the staging of the next
    // phase's B under the MFMAs of k step s).  MORE: a phase follows (its A / B are fetched here); TAIL: the reduction ends inside this phase.
    auto phase = [&](auto more_tag, auto tail_tag, auto next_tail_tag, int p) {
        constexpr bool MORE = decltype(more_tag)::value, TAIL = decltype(tail_tag)::value;
        const int kbase = p * S_KP;
        const int cur = (p & 1) * S_STAGE;
        Split8 fa;
        if constexpr (TAIL) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (kbase + 8 * half + e >= K) a[0][e] = 0.f;
        }
        fa.run(a[0]);
        // B fragments in two register sets (k step s + 1's nine ds_read_b128 are issued under the MFMAs of k step s), A fragments likewise.
        // One wave per SIMD: whatever is not placed in the gap after an MFMA runs with the matrix pipe idle, so the k step is written as 18
        // slots -- MFMA q, then slot q's share of the side work, then a scheduling barrier:
        //   slots 0 2 4 6     a split pair of A's next k step (11 VALU each)          slot 0 also: the two loads of A for the next phase
        //   slots 1 3 5 7 9   the next k step's B fragments (two reads each, the last one reads one)
        //   slots 8 10 12 14  a split pair of the next phase's B unit s               slots 15 16 17: its three plane stores; 17: its next loads
        bf16x8 fb[2][S_NT][3];
        Split8 fn, sb;                                          // the next k step's A fragments; the B unit being staged
        auto frag = [&](auto set_tag, int s, int u) {           // fragment u (0..8) of k step s into register set SET
            constexpr int SET = decltype(set_tag)::value;
            const int j = u / 3, pl = u % 3;
            fb[SET][j][pl] = *reinterpret_cast<const bf16x8*>(smem_c + cur + ((s * S_NT + j) * 3 + pl) * S_FRAG + lane * 16);
        };
        using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
#pragma unroll
        for (int u = 0; u < 9; ++u) frag(S0{}, 0, u);
#pragma unroll
        for (int s = 0; s < S_STEPS; ++s) {
            const int k0 = kbase + 16 * s;
            const bool live = !TAIL || k0 < K;                 // (TAIL only: uniform) k steps past the reduction's end do nothing
            const bf16x8 a0 = __builtin_bit_cast(bf16x8, fa.p0), a1 = __builtin_bit_cast(bf16x8, fa.p1), a2 = __builtin_bit_cast(bf16x8, fa.p2);
            const bool nxt = s + 1 < S_STEPS;
            const bool stg = MORE && s < S_UNITS;
            if (nxt) {
                if constexpr (TAIL) {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (k0 + 16 + 8 * half + e >= K) a[s + 1][e] = 0.f;
                }
            }
            if constexpr (MORE) {
                if (stg) {
                    if constexpr (BKC && decltype(next_tail_tag)::value) {
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (kbase + S_KP + b_kpos[s] + e >= K) braw[s][e] = 0.f;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 6 * S_NT; ++q) {
                if (live) {
                    // term order (A plane, B plane): (2,0) (0,2) (1,1) (1,0) (0,1) (0,0) -- gemm_x3_kernel's
                    const int term = q / S_NT, j = q % S_NT, sq = s & 1;
                    const bf16x8 av = term == 0 ? a2 : (term == 2 || term == 3) ? a1 : a0;
                    const bf16x8 bv = term == 1 ? fb[sq][j][2] : (term == 2 || term == 4) ? fb[sq][j][1] : fb[sq][j][0];
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[j], 0, 0, 0);
                }
                if (nxt && q < 8 && (q & 1) == 0) fn.pair(a[s + 1], q >> 1);
                if (q == 0) { if constexpr (MORE) load_a(s, kbase + S_KP); }          // (a[s] was consumed when this k step's fragments were built)
                if (nxt && (q & 1) == 1 && q < 10) {
                    const int u0 = q - 1;                                              // q = 1 3 5 7 9 -> fragments 0-1, 2-3, 4-5, 6-7, 8
                    if (s & 1) { frag(S0{}, s + 1, u0); if (u0 + 1 < 9) frag(S0{}, s + 1, u0 + 1); }
                    else { frag(S1{}, s + 1, u0); if (u0 + 1 < 9) frag(S1{}, s + 1, u0 + 1); }
                }
                if constexpr (MORE) {
                    if (stg) {
                        if (q >= 8 && q < 16 && (q & 1) == 0) sb.pair(braw[s], (q - 8) >> 1);
                        char* dst = smem_c + ((p + 1) & 1) * S_STAGE + b_lds[s];
                        if (q == 15) *reinterpret_cast<u32x4*>(dst) = sb.p0;
                        if (q == 16) *reinterpret_cast<u32x4*>(dst + S_FRAG) = sb.p1;
                        if (q == 17) { *reinterpret_cast<u32x4*>(dst + 2 * S_FRAG) = sb.p2; load_b(s, kbase + 2 * S_KP); }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (nxt) fa = fn;
        }
        __syncthreads();                                        // next phase's B is complete; nobody reads this phase's stage any more
    };
    using T = std::true_type; using F = std::false_type;
    const bool ktail = (K % S_KP) != 0;
    for (int p = 0; p + 2 < nph; ++p) phase(T{}, F{}, F{}, p);
    if (nph > 1) {                                              // the phase that stages the last one
        if (ktail) phase(T{}, F{}, T{}, nph - 2);
        else phase(T{}, F{}, F{}, nph - 2);
    }
    if (nph > 0) {
        if (ktail) phase(F{}, T{}, F{}, nph - 1);
        else phase(F{}, F{}, F{}, nph - 1);
    }

    // ---- epilogue: straight from the accumulators (N <= 96 columns: a row segment of 32 lanes is one 128-byte store)
    float* C = g.C + bz * g.sC;
    const bool relu = g.act == 1;
#pragma unroll
    for (int j = 0; j < S_NT; ++j) {
        const int col = 32 * j + l31;
        if (col < g.N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < g.M) {
                    float v = acc[j][r];
                    if (relu) v = fmaxf(v, 0.f);
                    C[(long long)row * g.ldc + col] = v;
                }
            }
        }
    }
}

int launch_gemm_x3s(const GemmArgs& g, bool bkc, hipStream_t stream) {
    const int exp_mode = gemm_option(7);
    if (exp_mode && bkc) {                                   // timing experiments (tools/bench_gemm_x3_skinny.py --exp): results are garbage
        const dim3 grid((unsigned)((g.M + S_ROWS - 1) / S_ROWS), (unsigned)g.batch);
#define LAUNCHE(E_)                                                                                                                        \
        { static bool once = false; if (!once) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x3s_kernel<true, E_>), hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS); once = true; } \
          hipLaunchKernelGGL((gemm_x3s_kernel<true, E_>), grid, dim3(256), S_LDS, stream, g); }
        switch (exp_mode) { case 1: LAUNCHE(1) break; case 2: LAUNCHE(2) break; case 4: LAUNCHE(4) break; case 8: LAUNCHE(8) break; case 16: LAUNCHE(16) break;
                            case 31: LAUNCHE(31) break; case 30: LAUNCHE(30) break; default: LAUNCHE(3) break; }
#undef LAUNCHE
        return check_launch("pulse_gemm_f32 (skinny-N tile, experiment)");
    }
    const dim3 grid((unsigned)((g.M + S_ROWS - 1) / S_ROWS), (unsigned)g.batch);
    static bool attr_done[2] = {false, false};
    hipError_t e = hipSuccess;
#define LAUNCHS(IDX, BK_)                                                                                                                  \
    if (!attr_done[IDX]) {                                                                                                                 \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x3s_kernel<BK_>), hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS);   \
        if (e != hipSuccess) { (void)hipGetLastError(); return kWideTileUnavailable; }                                                     \
        attr_done[IDX] = true;                                                                                                             \
    }                                                                                                                                      \
    hipLaunchKernelGGL((gemm_x3s_kernel<BK_>), grid, dim3(256), S_LDS, stream, g)
    if (bkc) { LAUNCHS(0, true); }
    else { LAUNCHS(1, false); }
#undef LAUNCHS
    return check_launch("pulse_gemm_f32 (skinny-N tile)");
}

}  // namespace pulse
