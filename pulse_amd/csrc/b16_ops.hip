// Support kernels of the bf16-STORAGE training path (mixed_precision, BASELINE.json configs[4]; the reference's autocast sites are
// phc/learning/amp_agent.py:671 and common_agent.py:426,461): activations, gradients and weight copies live in HBM as bf16 and feed
// pulse_gemm_x3p(planes = 1).  Everything here is HBM- or latency-bound glue around those GEMMs:
//
//   transpose_to_b16     fp32 W[out][in] -> bf16 W^T[in][out] once per optimiser step, so every input-gradient GEMM runs in the forward
//                        (both operands reduction-contiguous) form                       (nn.Linear backward, network_builder.py:105-124)
//   colsum_partial_b16   bias gradients = column sums of a bf16 gradient matrix          (same call sites as pulse_colsum_partial)
//   disc_penalty         AMPAgent._disc_loss gradient penalty (amp_agent.py:925-934): sum ||dD/dx||^2 partials + the scaled gradient dg
//                        that starts the penalty's backward pass, as fp32 and / or bf16
//   disc_reg             disc_logit_reg / disc_weight_decay gradient terms (amp_agent.py:919-923, 936-940): grad += alpha * w over up to
//                        four parameter ranges, with the ranges' sums of squares for the reported loss
//   disc_reward          AMPAgent._calc_disc_rewards (amp_agent.py:1027-1041): -log(max(1 - sigmoid(logit), 1e-4)) * scale
#include "common.h"

namespace pulse {

typedef unsigned int b16_u32x4 __attribute__((ext_vector_type(4)));

// out[z][c][r] = bf16(in[z][r][c]): 64 x 64 tiles through LDS, 16-byte reads of the fp32 rows, 8-byte writes of the bf16 rows.
__device__ __forceinline__ void transpose_tile_to_b16(const float* __restrict__ src, long long ld_in, int rows_in, int cols_in, unsigned short* __restrict__ dst,
                                                      long long ld_out, int r0, int c0, float (&tile)[64][65]) {
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;           // 16 x 16 threads, 4 columns x 4 rows each
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 16 * i, c = c0 + tx * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < rows_in) {
            if (c + 3 < cols_in) {
                const float4 t = *reinterpret_cast<const float4*>(src + (long long)r * ld_in + c);
                v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) if (c + k < cols_in) v[k] = src[(long long)r * ld_in + c + k];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) tile[ty + 16 * i][tx * 4 + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int oc = c0 + ty + 16 * i;                               // output row = input column
        const int orr = r0 + tx * 4;                                   // output columns = input rows
        if (oc >= cols_in || orr >= rows_in) continue;
        const float a = tile[tx * 4][ty + 16 * i], b = tile[tx * 4 + 1][ty + 16 * i], c = tile[tx * 4 + 2][ty + 16 * i], d = tile[tx * 4 + 3][ty + 16 * i];
        unsigned short* o = dst + (long long)oc * ld_out + orr;
        if (orr + 3 < rows_in) {
            *reinterpret_cast<uint2*>(o) = make_uint2(split_pack_rn(a, b), split_pack_rn(c, d));
        } else {
            const float v[4] = {a, b, c, d};
            for (int k = 0; k < 4 && orr + k < rows_in; ++k) o[k] = (unsigned short)(split_pack_rn(v[k], 0.f) & 0xffffu);
        }
    }
}

__global__ void __launch_bounds__(256) transpose_to_b16_kernel(const float* __restrict__ in, long long ld_in, int rows_in, int cols_in,
                                                              unsigned short* __restrict__ out, long long ld_out, long long stride_in, long long stride_out) {
    __shared__ float tile[64][65];
    transpose_tile_to_b16(in + blockIdx.z * stride_in, ld_in, rows_in, cols_in, out + blockIdx.z * stride_out, ld_out, blockIdx.y * 64, blockIdx.x * 64, tile);
}

// The bf16 images a training pass needs of a flat fp32 parameter buffer, in ONE launch: the straight image flat16[i] = bf16(flat[i]) (the first
// lin_blocks workgroups, grid-stride over 8-element pieces; elements [count, roundup8(count)) are written as zero) and up to four transposed
// W^T images (the remaining workgroups, one 64 x 64 tile each).
struct WeightsTr { const float* in; long long ld_in; int rows, cols; unsigned short* out; long long ld_out, stride_in, stride_out; int tiles_x, tiles_y, first_block; };
struct WeightsArgs { const float* flat; unsigned short* flat16; long long count; int lin_blocks, ntr; WeightsTr tr[4]; };
__global__ void __launch_bounds__(256) weights_to_b16_kernel(const WeightsArgs a) {
    __shared__ float tile[64][65];
    const int blk = blockIdx.x;
    if (blk < a.lin_blocks) {
        const long long pieces = (a.count + 7) >> 3;
        for (long long i = (long long)blk * 256 + threadIdx.x; i < pieces; i += (long long)a.lin_blocks * 256) {
            const long long e = i * 8;
            float v[8];
            if (e + 7 < a.count) {
                const float4 t0 = *reinterpret_cast<const float4*>(a.flat + e), t1 = *reinterpret_cast<const float4*>(a.flat + e + 4);
                v[0] = t0.x; v[1] = t0.y; v[2] = t0.z; v[3] = t0.w; v[4] = t1.x; v[5] = t1.y; v[6] = t1.z; v[7] = t1.w;
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = e + k < a.count ? a.flat[e + k] : 0.f;
            }
            b16_u32x4 q;
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = split_pack_rn(v[2 * k], v[2 * k + 1]);
            *reinterpret_cast<b16_u32x4*>(a.flat16 + e) = q;
        }
        return;
    }
    int t = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k) if (k < a.ntr && blk >= a.tr[k].first_block) t = k;
    const WeightsTr& w = a.tr[t];
    const int local = blk - w.first_block;
    const int bx = local % w.tiles_x, by = (local / w.tiles_x) % w.tiles_y, bz = local / (w.tiles_x * w.tiles_y);
    transpose_tile_to_b16(w.in + bz * w.stride_in, w.ld_in, w.rows, w.cols, w.out + bz * w.stride_out, w.ld_out, by * 64, bx * 64, tile);
}

// partial[chunk][n] = sum over the chunk's rows of X[m][n], X bf16.  256 threads = CG column groups (8 columns = 16 bytes each) x 256 / CG
// row lanes; four independent loads in flight per thread; fp32 accumulation in a fixed order.  CG = 32: 256 columns per workgroup (many
// chunks); CG = 4: 32 columns per workgroup, for the launches that write one partial row per GRADIENT SLAB (num_chunks = the book's
// split-K count, partial = the slabs themselves): the bias gradient then needs no reduce launch of its own -- the slab reduce that the
// weight gradients need anyway sums it -- and the few chunks still give every CU a workgroup.
// ``w`` (optional): per-row bf16 weights (stride ws elements): partial = sum_m w[m] X[m][n] -- the weight gradient of a one-output Linear
// (the discriminator's logit layer: d w3 = sum_m dlogit[m] H2[m][:]) without a 1 x n GEMM and its reduce.
template <int CG, int NACC>
__global__ void __launch_bounds__(256) colsum_partial_b16_kernel(const unsigned short* __restrict__ X, int M, int N, long long ld, int rows_per_chunk,
                                                                float* __restrict__ partial, long long ldp, const unsigned short* __restrict__ w,
                                                                long long ws) {
    constexpr int RL = 256 / CG;
    __shared__ float red[RL][CG][9];
    const int cg = threadIdx.x % CG, rl = threadIdx.x / CG;
    const int c = blockIdx.x * (CG * 8) + cg * 8;
    const int r0 = blockIdx.y * rows_per_chunk;
    const int r1 = min(M, r0 + rows_per_chunk);
    float s[NACC][8];
#pragma unroll
    for (int q = 0; q < NACC; ++q)
#pragma unroll
        for (int k = 0; k < 8; ++k) s[q][k] = 0.f;
    if (c < N) {
        const unsigned short* p = X + c;
        auto add = [&](int q, int r) {
            const b16_u32x4 t = *reinterpret_cast<const b16_u32x4*>(p + (long long)r * ld);
            if (w) {
                const float wv = split_bitsf((unsigned)w[(long long)r * ws] << 16);
#pragma unroll
                for (int k = 0; k < 4; ++k) { s[q][2 * k] += wv * split_bitsf(t[k] << 16); s[q][2 * k + 1] += wv * split_bitsf(t[k] & 0xffff0000u); }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) { s[q][2 * k] += split_bitsf(t[k] << 16); s[q][2 * k + 1] += split_bitsf(t[k] & 0xffff0000u); }
            }
        };
        int r = r0 + rl;
        for (; r + (NACC - 1) * RL < r1; r += NACC * RL) {
#pragma unroll
            for (int q = 0; q < NACC; ++q) add(q, r + q * RL);
        }
        for (; r < r1; r += RL) add(0, r);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < NACC; ++q) t += s[q][k];
        red[rl][cg][k] = t;
    }
    __syncthreads();
    // 8 * CG output columns, one thread each: the RL row lanes' sums added in lane order
    if (threadIdx.x < 8 * CG) {
        const int g = threadIdx.x >> 3, k = threadIdx.x & 7;
        const int col = blockIdx.x * (CG * 8) + g * 8 + k;
        if (col < N) {
            float t = 0.f;
#pragma unroll 8
            for (int w_ = 0; w_ < RL; ++w_) t += red[w_][g][k];
            partial[(long long)blockIdx.y * ldp + col] = t;
        }
    }
}

// dg = scale * G (fp32 and / or bf16), partials[block] = sum of G^2 over the block's elements.  G is (rows, ld) with its pad columns zero.
__global__ void __launch_bounds__(256) disc_penalty_kernel(const float* __restrict__ G, long long ldg, int rows, int cols4, float scale,
                                                          float* __restrict__ out32, long long ld32, unsigned short* __restrict__ out16, long long ld16,
                                                          float* __restrict__ partials) {
    __shared__ float red[4];
    float acc = 0.f;
    const long long total = (long long)rows * cols4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int r = (int)(i / cols4), c = (int)(i - (long long)r * cols4) * 4;
        const float4 v = *reinterpret_cast<const float4*>(G + (long long)r * ldg + c);
        acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        const float a = v.x * scale, b = v.y * scale, cc = v.z * scale, d = v.w * scale;
        if (out32) *reinterpret_cast<float4*>(out32 + (long long)r * ld32 + c) = make_float4(a, b, cc, d);
        if (out16) *reinterpret_cast<uint2*>(out16 + (long long)r * ld16 + c) = make_uint2(split_pack_rn(a, b), split_pack_rn(cc, d));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

struct DiscRegArgs { long long off[4]; long long len[4]; float alpha[4]; int n; };

// grad[off_r + i] += alpha_r * flat[off_r + i];  partials[block * 4 + r] = sum over the block's share of flat[range r]^2
__global__ void __launch_bounds__(256) disc_reg_kernel(const float* __restrict__ flat, float* __restrict__ grad, const DiscRegArgs a, float* __restrict__ partials) {
    __shared__ float red[4][4];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (r >= a.n) break;
        const float* w = flat + a.off[r];
        float* g = grad ? grad + a.off[r] : nullptr;
        const float al = a.alpha[r];
        if (((a.off[r] | a.len[r]) & 3) == 0) {                          // 16-byte path (every weight matrix of a ParamBook)
            const long long n4 = a.len[r] >> 2;
            const bool upd = g && al != 0.f;
            for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
                const float4 v = reinterpret_cast<const float4*>(w)[i];
                acc[r] += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                if (upd) {
                    float4 t = reinterpret_cast<float4*>(g)[i];
                    t.x += al * v.x; t.y += al * v.y; t.z += al * v.z; t.w += al * v.w;
                    reinterpret_cast<float4*>(g)[i] = t;
                }
            }
            continue;
        }
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.len[r]; i += (long long)gridDim.x * 256) {
            const float v = w[i];
            acc[r] += v * v;
            if (g && al != 0.f) g[i] += al * v;
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[r] += __shfl_xor(acc[r], o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][r] = acc[r];
    }
    __syncthreads();
    if (threadIdx.x < 4) partials[blockIdx.x * 4 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// out[off_r + i] = scale * sum_{s < nslabs_r} slabs[s * stride + off_r + i] (+ alpha_r * flat[off_r + i]) over up to 8 regions of a flat
// gradient buffer, each with its own slab count (a weight-gradient launch that split its reduction 4 ways wrote 4 slabs: the other slabs of
// that region are never read); sq_partials[block] = the block's share of sum out^2 (the gradient-norm clip needs it: no separate pass over
// the gradient), w2_partials[block * 8 + r] = its share of sum flat[region r]^2 (the regularisers' loss terms).
// A region may name its OWN source (src[r] != null: nslabs[r] partial rows of sstride[r] floats, element i of the region at src[r][s * sstride[r] + i]):
// the column-sum partials of a bias gradient or the scratch of a weight gradient that was split wider than the slab count are summed here
// instead of by a pulse_reduce_slabs launch of their own.
constexpr int kReduceRegions = 32;        // (8 until v29; the regularisers' w2 partials cover the first 8)
struct ReduceRegions { long long off[kReduceRegions]; long long count[kReduceRegions]; int nslabs[kReduceRegions]; float alpha[kReduceRegions];
                       const float* src[kReduceRegions]; long long sstride[kReduceRegions]; int n; };
__global__ void __launch_bounds__(256) reduce_grads_kernel(const float* __restrict__ slabs, long long slab_stride, const ReduceRegions rg, float* __restrict__ out,
                                                          float scale, const float* __restrict__ flat, float* __restrict__ sq_partials,
                                                          float* __restrict__ w2_partials) {
    __shared__ float red[4][9];
    float sq = 0.f, w2[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < rg.n; ++r) {
        float w2r = 0.f;
        const float* base = rg.src[r] ? rg.src[r] : slabs + rg.off[r];
        const long long stride = rg.src[r] ? rg.sstride[r] : slab_stride;
        float* o = out + rg.off[r];
        const float* w = flat ? flat + rg.off[r] : nullptr;
        const float al = rg.alpha[r];
        const int ns = rg.nslabs[r];
        const long long n4 = rg.count[r] >> 2;                          // offsets / counts are multiples of 4 floats (checked by the launcher)
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
            // the SAME association as reduce_slabs_kernel (gemm_f32.hip): slab 0 + four chains over the groups of four, remainder onto chain 0,
            // ((c0 + c1) + (c2 + c3)) -- a gradient reduced by either kernel is bit-identical (the data-parallel path reduces bucket by bucket
            // with pulse_reduce_slabs, the single-GPU path with this kernel)
            float4 a1 = make_float4(0.f, 0.f, 0.f, 0.f), a2 = a1, a3 = a1;
            float4 a = a1;
            if (ns > 0) a = reinterpret_cast<const float4*>(base)[i];         // ns == 0: a region no launch of this pass wrote (zero gradient)
            int k = 1;
            // [r6] sixteen slabs' loads in flight where a region has that many (the bias column-sum partials and the heads' wide split have 32 - 128
            // rows: at four loads per round trip a thread walked up to 32 dependent round trips while the big regions' threads had long finished);
            // the adds keep their order, so the association -- and every bit -- is unchanged
            for (; k + 15 < ns; k += 16) {
                float4 v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = reinterpret_cast<const float4*>(base + (k + u) * stride)[i];
#pragma unroll
                for (int u = 0; u < 16; u += 4) {
                    a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w;
                    a1.x += v[u + 1].x; a1.y += v[u + 1].y; a1.z += v[u + 1].z; a1.w += v[u + 1].w;
                    a2.x += v[u + 2].x; a2.y += v[u + 2].y; a2.z += v[u + 2].z; a2.w += v[u + 2].w;
                    a3.x += v[u + 3].x; a3.y += v[u + 3].y; a3.z += v[u + 3].z; a3.w += v[u + 3].w;
                }
            }
            for (; k + 3 < ns; k += 4) {
                const float4 v0 = reinterpret_cast<const float4*>(base + k * stride)[i];
                const float4 v1 = reinterpret_cast<const float4*>(base + (k + 1) * stride)[i];
                const float4 v2 = reinterpret_cast<const float4*>(base + (k + 2) * stride)[i];
                const float4 v3 = reinterpret_cast<const float4*>(base + (k + 3) * stride)[i];
                a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w;
                a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
                a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
                a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
            }
            // the last one to three slabs go onto chain 0 in order; their loads are issued together (uniform branches: ns belongs to the region)
            const int rem = ns - k;
            if (rem == 3) {
                const float4 t0 = reinterpret_cast<const float4*>(base + k * stride)[i];
                const float4 t1 = reinterpret_cast<const float4*>(base + (k + 1) * stride)[i];
                const float4 t2 = reinterpret_cast<const float4*>(base + (k + 2) * stride)[i];
                a.x += t0.x; a.y += t0.y; a.z += t0.z; a.w += t0.w;
                a.x += t1.x; a.y += t1.y; a.z += t1.z; a.w += t1.w;
                a.x += t2.x; a.y += t2.y; a.z += t2.z; a.w += t2.w;
            } else if (rem == 2) {
                const float4 t0 = reinterpret_cast<const float4*>(base + k * stride)[i];
                const float4 t1 = reinterpret_cast<const float4*>(base + (k + 1) * stride)[i];
                a.x += t0.x; a.y += t0.y; a.z += t0.z; a.w += t0.w;
                a.x += t1.x; a.y += t1.y; a.z += t1.z; a.w += t1.w;
            } else if (rem == 1) {
                const float4 t0 = reinterpret_cast<const float4*>(base + k * stride)[i];
                a.x += t0.x; a.y += t0.y; a.z += t0.z; a.w += t0.w;
            }
            a.x = (a.x + a1.x) + (a2.x + a3.x); a.y = (a.y + a1.y) + (a2.y + a3.y); a.z = (a.z + a1.z) + (a2.z + a3.z); a.w = (a.w + a1.w) + (a2.w + a3.w);
            a.x *= scale; a.y *= scale; a.z *= scale; a.w *= scale;
            if (w) {
                const float4 v = reinterpret_cast<const float4*>(w)[i];
                w2r += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                if (al != 0.f) { a.x += al * v.x; a.y += al * v.y; a.z += al * v.z; a.w += al * v.w; }
            }
            sq += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
            reinterpret_cast<float4*>(o)[i] = a;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (q == r) w2[q] += w2r;
    }
    float vals[9] = {sq, w2[0], w2[1], w2[2], w2[3], w2[4], w2[5], w2[6], w2[7]};
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        float v = vals[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 9) {
        const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        if (threadIdx.x == 0) { if (sq_partials) sq_partials[blockIdx.x] = t; }
        else if (w2_partials) w2_partials[blockIdx.x * 8 + threadIdx.x - 1] = t;
    }
}

__global__ void __launch_bounds__(256) disc_reward_kernel(const float* __restrict__ logits, long long ls, long long n, float scale, float* __restrict__ out, long long os) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = logits[i * ls];
    const float prob = 1.f / (1.f + expf(-x));                          // the reference's op sequence, op for op (amp_agent.py:1033-1038)
    out[i * os] = -logf(fmaxf(1.f - prob, 0.0001f)) * scale;
}

}  // namespace pulse

using namespace pulse;

extern "C" {

int pulse_transpose_to_b16(const float* in, int64_t ld_in, int32_t rows_in, int32_t cols_in, void* out, int64_t ld_out, int32_t batch,
                           int64_t stride_in, int64_t stride_out, pulse_stream_t s) {
    PULSE_REQUIRE(rows_in >= 0 && cols_in >= 0 && batch >= 0, "pulse_transpose_to_b16: negative size");
    if (rows_in == 0 || cols_in == 0 || batch == 0) return PULSE_OK;
    PULSE_REQUIRE(in && out, "pulse_transpose_to_b16: null pointer");
    PULSE_REQUIRE(ld_in >= cols_in && (ld_in % 4) == 0 && (stride_in % 4) == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0,
                  "pulse_transpose_to_b16: input rows must be 16-byte aligned");
    PULSE_REQUIRE(ld_out >= rows_in && (ld_out % 4) == 0 && (stride_out % 4) == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0,
                  "pulse_transpose_to_b16: output rows must be 8-byte aligned and hold rows_in columns");
    const dim3 grid((unsigned)((cols_in + 63) / 64), (unsigned)((rows_in + 63) / 64), (unsigned)batch);
    hipLaunchKernelGGL(transpose_to_b16_kernel, grid, dim3(256), 0, as_stream(s), in, (long long)ld_in, rows_in, cols_in,
                       reinterpret_cast<unsigned short*>(out), (long long)ld_out, (long long)stride_in, (long long)stride_out);
    return check_launch("pulse_transpose_to_b16");
}

int pulse_sizeof_b16_transpose(void) { return (int)sizeof(pulse_b16_transpose); }

int pulse_weights_to_b16(const float* flat, int64_t count, void* flat16, int32_t num_transposes, const pulse_b16_transpose* tr, pulse_stream_t s) {
    PULSE_REQUIRE(count >= 0 && num_transposes >= 0 && num_transposes <= 4, "pulse_weights_to_b16: count >= 0 and at most 4 transposes");
    PULSE_REQUIRE(num_transposes == 0 || tr != nullptr, "pulse_weights_to_b16: null transpose list");
    if (count == 0 && num_transposes == 0) return PULSE_OK;
    PULSE_REQUIRE(count == 0 || (flat && flat16 && (reinterpret_cast<uintptr_t>(flat) & 15) == 0 && (reinterpret_cast<uintptr_t>(flat16) & 15) == 0),
                  "pulse_weights_to_b16: flat / flat16 must be 16-byte aligned");
    WeightsArgs a;
    a.flat = flat; a.flat16 = reinterpret_cast<unsigned short*>(flat16); a.count = count; a.ntr = 0;
    const long long pieces = (count + 7) / 8;
    long long lin = (pieces + 255) / 256;
    if (lin > 2048) lin = 2048;
    a.lin_blocks = (int)lin;
    long long next = lin;
    for (int i = 0; i < 4; ++i) {
        WeightsTr& w = a.tr[i];
        w = WeightsTr{nullptr, 0, 0, 0, nullptr, 0, 0, 0, 1, 1, 0x7fffffff};
        if (i >= num_transposes) continue;
        const pulse_b16_transpose& d = tr[i];
        PULSE_REQUIRE(d.rows >= 0 && d.cols >= 0 && d.batch >= 0, "pulse_weights_to_b16: negative transpose size");
        if (d.rows == 0 || d.cols == 0 || d.batch == 0) continue;
        PULSE_REQUIRE(d.in && d.out, "pulse_weights_to_b16: null transpose pointer");
        PULSE_REQUIRE(d.ld_in >= d.cols && (d.ld_in % 4) == 0 && (d.stride_in % 4) == 0 && (reinterpret_cast<uintptr_t>(d.in) & 15) == 0,
                      "pulse_weights_to_b16: transpose input rows must be 16-byte aligned");
        PULSE_REQUIRE(d.ld_out >= d.rows && (d.ld_out % 4) == 0 && (d.stride_out % 4) == 0 && (reinterpret_cast<uintptr_t>(d.out) & 7) == 0,
                      "pulse_weights_to_b16: transpose output rows must be 8-byte aligned and hold `rows` columns");
        WeightsTr& o = a.tr[a.ntr++];
        o.in = d.in; o.ld_in = d.ld_in; o.rows = d.rows; o.cols = d.cols; o.out = reinterpret_cast<unsigned short*>(d.out); o.ld_out = d.ld_out;
        o.stride_in = d.stride_in; o.stride_out = d.stride_out;
        o.tiles_x = (d.cols + 63) / 64; o.tiles_y = (d.rows + 63) / 64; o.first_block = (int)next;
        next += (long long)o.tiles_x * o.tiles_y * d.batch;
        PULSE_REQUIRE(next < (1LL << 31), "pulse_weights_to_b16: too many tiles");
    }
    if (next == 0) return PULSE_OK;
    hipLaunchKernelGGL(weights_to_b16_kernel, dim3((unsigned)next), dim3(256), 0, as_stream(s), a);
    return check_launch("pulse_weights_to_b16");
}

static int colsum_b16_launch(const char* what, const void* x, int32_t m, int32_t n, int64_t ld, const void* w, int64_t w_stride, int32_t num_chunks,
                             float* partial, int64_t ld_partial, pulse_stream_t s) {
    PULSE_REQUIRE(m >= 0 && n >= 0 && num_chunks >= 1, "%s: bad sizes", what);
    if (n == 0) return PULSE_OK;
    PULSE_REQUIRE(x && partial && ld >= ((n + 7) & ~7) && ld_partial >= n, "%s: bad pointers / pitches (ld must cover roundup8(n))", what);
    PULSE_REQUIRE((ld % 8) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "%s: x rows must be 16-byte aligned", what);
    const int rows = (m + num_chunks - 1) / num_chunks;
    const unsigned short* xp = reinterpret_cast<const unsigned short*>(x);
    const unsigned short* wp = reinterpret_cast<const unsigned short*>(w);
    if ((long long)((n + 255) / 256) * num_chunks >= 256)
        hipLaunchKernelGGL((colsum_partial_b16_kernel<32, 4>), dim3((unsigned)((n + 255) / 256), (unsigned)num_chunks), dim3(256), 0, as_stream(s), xp, m, n,
                           (long long)ld, rows > 0 ? rows : 1, partial, (long long)ld_partial, wp, (long long)w_stride);
    else
        hipLaunchKernelGGL((colsum_partial_b16_kernel<4, 8>), dim3((unsigned)((n + 31) / 32), (unsigned)num_chunks), dim3(256), 0, as_stream(s), xp, m, n,
                           (long long)ld, rows > 0 ? rows : 1, partial, (long long)ld_partial, wp, (long long)w_stride);
    return check_launch(what);
}

int pulse_colsum_partial_b16(const void* x, int32_t m, int32_t n, int64_t ld, int32_t num_chunks, float* partial, int64_t ld_partial, pulse_stream_t s) {
    return colsum_b16_launch("pulse_colsum_partial_b16", x, m, n, ld, nullptr, 0, num_chunks, partial, ld_partial, s);
}

int pulse_colsum_weighted_b16(const void* x, int32_t m, int32_t n, int64_t ld, const void* w, int64_t w_stride, int32_t num_chunks, float* partial,
                              int64_t ld_partial, pulse_stream_t s) {
    PULSE_REQUIRE(w != nullptr && w_stride >= 1, "pulse_colsum_weighted_b16: weights missing");
    return colsum_b16_launch("pulse_colsum_weighted_b16", x, m, n, ld, w, w_stride, num_chunks, partial, ld_partial, s);
}

int pulse_disc_penalty(const float* g, int64_t ldg, int32_t rows, int32_t cols, float scale, float* out32, int64_t ld32, void* out16, int64_t ld16,
                       float* partials, int32_t num_blocks, pulse_stream_t s) {
    PULSE_REQUIRE(rows >= 1 && cols >= 1 && num_blocks >= 1, "pulse_disc_penalty: bad sizes");
    PULSE_REQUIRE(g && partials, "pulse_disc_penalty: null pointer");
    PULSE_REQUIRE((cols % 4) == 0 && (ldg % 4) == 0 && ldg >= cols && (reinterpret_cast<uintptr_t>(g) & 15) == 0,
                  "pulse_disc_penalty: cols / pitch must be multiples of 4 floats (the pad columns of G are zero and are processed with it)");
    PULSE_REQUIRE(!out32 || ((ld32 % 4) == 0 && ld32 >= cols && (reinterpret_cast<uintptr_t>(out32) & 15) == 0), "pulse_disc_penalty: fp32 output rows must be 16-byte aligned");
    PULSE_REQUIRE(!out16 || ((ld16 % 4) == 0 && ld16 >= cols && (reinterpret_cast<uintptr_t>(out16) & 7) == 0), "pulse_disc_penalty: bf16 output rows must be 8-byte aligned");
    hipLaunchKernelGGL(disc_penalty_kernel, dim3((unsigned)num_blocks), dim3(256), 0, as_stream(s), g, (long long)ldg, rows, cols / 4, scale, out32,
                       (long long)ld32, reinterpret_cast<unsigned short*>(out16), (long long)ld16, partials);
    return check_launch("pulse_disc_penalty");
}

int pulse_disc_reg(const float* flat, float* grad, int32_t num_ranges, const int64_t* offsets, const int64_t* lengths, const float* alphas,
                   float* partials, int32_t num_blocks, pulse_stream_t s) {
    PULSE_REQUIRE(num_ranges >= 1 && num_ranges <= 4 && num_blocks >= 1, "pulse_disc_reg: 1..4 ranges");
    PULSE_REQUIRE(flat && offsets && lengths && alphas && partials, "pulse_disc_reg: null pointer");
    PULSE_REQUIRE((reinterpret_cast<uintptr_t>(flat) & 15) == 0 && (!grad || (reinterpret_cast<uintptr_t>(grad) & 15) == 0), "pulse_disc_reg: flat / grad must be 16-byte aligned");
    DiscRegArgs a;
    a.n = num_ranges;
    for (int r = 0; r < 4; ++r) {
        a.off[r] = r < num_ranges ? offsets[r] : 0; a.len[r] = r < num_ranges ? lengths[r] : 0; a.alpha[r] = r < num_ranges ? alphas[r] : 0.f;
        PULSE_REQUIRE(a.off[r] >= 0 && a.len[r] >= 0, "pulse_disc_reg: negative range");
    }
    hipLaunchKernelGGL(disc_reg_kernel, dim3((unsigned)num_blocks), dim3(256), 0, as_stream(s), flat, grad, a, partials);
    return check_launch("pulse_disc_reg");
}

int pulse_reduce_grads(const float* slabs, int64_t slab_stride, int32_t num_regions, const int64_t* offsets, const int64_t* counts, const int32_t* nslabs,
                       const float* alphas, const float* const* region_src, const int64_t* region_src_stride, float* out, float scale, const float* flat,
                       float* sq_partials, float* w2_partials, int32_t num_blocks, pulse_stream_t s) {
    PULSE_REQUIRE(num_regions >= 1 && num_regions <= kReduceRegions && num_blocks >= 1, "pulse_reduce_grads: 1..32 regions");
    PULSE_REQUIRE(slabs && offsets && counts && nslabs && out, "pulse_reduce_grads: null pointer");
    PULSE_REQUIRE((slab_stride % 4) == 0 && (reinterpret_cast<uintptr_t>(slabs) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
                  (!flat || (reinterpret_cast<uintptr_t>(flat) & 15) == 0), "pulse_reduce_grads: 16-byte alignment required");
    ReduceRegions rg;
    rg.n = num_regions;
    for (int r = 0; r < kReduceRegions; ++r) {
        rg.off[r] = r < num_regions ? offsets[r] : 0; rg.count[r] = r < num_regions ? counts[r] : 0; rg.nslabs[r] = r < num_regions ? nslabs[r] : 1;
        rg.alpha[r] = (r < num_regions && alphas) ? alphas[r] : 0.f;
        rg.src[r] = (r < num_regions && region_src) ? region_src[r] : nullptr;
        rg.sstride[r] = (rg.src[r] && region_src_stride) ? region_src_stride[r] : 0;
        PULSE_REQUIRE(!rg.src[r] || ((reinterpret_cast<uintptr_t>(rg.src[r]) & 15) == 0 && (rg.sstride[r] % 4) == 0 && (rg.nslabs[r] == 1 || rg.sstride[r] >= rg.count[r])),
                      "pulse_reduce_grads: a region's own source must be 16-byte aligned with a row stride that is a multiple of 4 floats covering the region");
        PULSE_REQUIRE(rg.off[r] >= 0 && rg.count[r] >= 0 && (rg.off[r] % 4) == 0 && (rg.count[r] % 4) == 0 && rg.nslabs[r] >= 0,
                      "pulse_reduce_grads: region offsets / counts must be non-negative multiples of 4 floats, slab counts >= 0");
    }
    hipLaunchKernelGGL(reduce_grads_kernel, dim3((unsigned)num_blocks), dim3(256), 0, as_stream(s), slabs, (long long)slab_stride, rg, out, scale, flat,
                       sq_partials, w2_partials);
    return check_launch("pulse_reduce_grads");
}

int pulse_disc_reward(const float* logits, int64_t logit_stride, int64_t n, float scale, float* out, int64_t out_stride, pulse_stream_t s) {
    PULSE_REQUIRE(n >= 0, "pulse_disc_reward: negative size");
    if (n == 0) return PULSE_OK;
    PULSE_REQUIRE(logits && out && logit_stride >= 1 && out_stride >= 1, "pulse_disc_reward: bad arguments");
    hipLaunchKernelGGL(disc_reward_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(s), logits, (long long)logit_stride, (long long)n, scale, out,
                       (long long)out_stride);
    return check_launch("pulse_disc_reward");
}
}
