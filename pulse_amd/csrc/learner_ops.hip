// Learner-side elementwise / reduction kernels of the PPO update on gfx950 (all HBM- or latency-bound).
//
//   rms_normalize / rms_update   RunningMeanStd.forward, phc/utils/running_mean_std.py:56-109, fused with the
//                                AMPDataset minibatch row gather (phc/learning/amp_datasets.py:81-94)
//   policy_sample                rl_games ModelA2CContinuousLogStd eval branch + value un-normalise
//                                (phc/learning/common_agent.py:262-288)
//   ppo_loss                     _actor_loss/_critic_loss/bound_loss + neglogp + policy_kl and their analytic
//                                gradients (phc/learning/common_agent.py:400-491,512-520,564-587)
//   advantage_moments/normalize  _calc_advs (phc/learning/common_agent.py:589-599)
//   sqnorm_partial / adam_step   clip_grad_norm_ + Adam.step over one flat parameter buffer (:472-478, :66)
//
// Statistics are accumulated in fp64 (the reference keeps fp64 running buffers; fp64 partials make the
// batch moments independent of the block decomposition to ~1e-16).  Compiled with -ffp-contract=off.
#include "common.h"

namespace pulse {

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_sumf(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ------------------------------------------------------------------------------------------------
// RunningMeanStd: wide matrices (cols >= 64): threads own columns, blocks own row chunks.
// ------------------------------------------------------------------------------------------------
constexpr int kRmsMaxColsPerThread = 12;  // 256 threads x 12 = 3072 columns (AMP obs is 10 x 232 = 2320)

__global__ void __launch_bounds__(256) rms_normalize_wide_kernel(const float* __restrict__ x, long long x_stride,
                                                                const long long* __restrict__ row_idx, int rows, int cols,
                                                                const double* __restrict__ mean, const double* __restrict__ var,
                                                                float eps, float clip, int mode, float* __restrict__ y,
                                                                long long y_stride, int y_cols, double* __restrict__ partials) {
    const int tid = threadIdx.x;
    const int nblk = gridDim.x;
    const int per = (rows + nblk - 1) / nblk;
    const int r0 = blockIdx.x * per;
    const int r1 = min(rows, r0 + per);
    float mu[kRmsMaxColsPerThread], den[kRmsMaxColsPerThread];
    double s1[kRmsMaxColsPerThread], s2[kRmsMaxColsPerThread];
#pragma unroll
    for (int j = 0; j < kRmsMaxColsPerThread; ++j) {
        const int c = tid + 256 * j;
        s1[j] = 0.0; s2[j] = 0.0;
        if (c < cols) {
            mu[j] = (float)mean[c];
            den[j] = sqrtf((float)var[c] + eps);
        } else { mu[j] = 0.f; den[j] = 1.f; }
    }
    for (int r = r0; r < r1; ++r) {
        const long long src = row_idx ? row_idx[r] : (long long)r;
        const float* xr = x + src * x_stride;
        float* yr = y + (long long)r * y_stride;
#pragma unroll
        for (int j = 0; j < kRmsMaxColsPerThread; ++j) {
            const int c = tid + 256 * j;
            if (c < cols) {
                const float v = xr[c];
                s1[j] += (double)v; s2[j] += (double)v * (double)v;
                float o;
                if (mode == 0) o = clampf((v - mu[j]) / den[j], -clip, clip);
                else o = den[j] * clampf(v, -clip, clip) + mu[j];
                yr[c] = o;
            } else if (c < y_cols) {
                yr[c] = 0.f;
            }
        }
    }
    if (partials) {
        double* p = partials + (long long)blockIdx.x * 2 * cols;
#pragma unroll
        for (int j = 0; j < kRmsMaxColsPerThread; ++j) {
            const int c = tid + 256 * j;
            if (c < cols) { p[c] = s1[j]; p[cols + c] = s2[j]; }
        }
    }
}

// 16-byte variant of the wide kernel: each thread owns groups of 4 adjacent columns (a wave reads 1 KiB
// of a row per instruction).  Needs 16-byte aligned rows on both sides and y_cols % 4 == 0.
constexpr int kRmsVecGroups = 3;   // 256 threads x 4 columns x 3 groups = 3072 columns
constexpr int kRmsRowsPerPass = 64;
constexpr int kRmsRowsInFlight = 4;     // rows whose loads are in flight per workgroup (4 KB each): the row loop is a latency x concurrency product

// OUT = 1: besides y, the three bf16 planes of every output element (common.h: split_pair3) go to planes[p * plane_stride + row * planes_ld + col]
// -- the layer-1 operand of the planar GEMM (gemm_x3p.hip) written by its producer instead of a separate split pass.
// OUT = 2: the output IS a bf16 matrix (planes[row * planes_ld + col], y unused): the layer-1 operand of the bf16-storage training path
// (mixed_precision; a bf16 autocast Linear rounds its fp32 input exactly like this).
// G: column groups per thread.  G = 1 (rows of <= 1024 columns: every observation of this path) needs 62 - 70 VGPRs instead of 141 - 146, so
// three workgroups share a CU instead of one and the 512 workgroups of a 16384-row minibatch are resident at once: 28.4 -> 22.0 us for
// 16384 x 934 -> 960 (5.6 TB/s of read + write; eight rows in flight per half instead of four measured 24.9).  [r6]
template <int OUT, int G = kRmsVecGroups>
__global__ void __launch_bounds__(512) rms_normalize_vec4_kernel(const float* __restrict__ x, long long x_stride,
                                                                const long long* __restrict__ row_idx, int rows, int cols,
                                                                const double* __restrict__ mean, const double* __restrict__ var,
                                                                float eps, float clip, int mode, float* __restrict__ y,
                                                                long long y_stride, int y_cols, double* __restrict__ partials,
                                                                unsigned short* __restrict__ planes, long long plane_stride, long long planes_ld,
                                                                float* __restrict__ raw = nullptr, long long raw_stride = 0) {
    // 512 threads = two halves of 256 column owners: the halves take alternate groups of kRmsRowsInFlight rows of the workgroup's row range and
    // their moment sums are combined through LDS at the end.  (One workgroup per CU with four waves could not hide the row loads' latency:
    // 4096 x 1960 rows took 29 us = 1.6 TB/s; the partials per WORKGROUP, which rms_update has to read back, stay what they were.)
    const int tid = threadIdx.x & 255, hf = threadIdx.x >> 8;
    const int nblk = gridDim.x;
    const int per = (rows + nblk - 1) / nblk;
    const int r0 = blockIdx.x * per;
    const int r1 = min(rows, r0 + per);
    float mu[G][4], den[G][4];
    double s1[G][4], s2[G][4];
#pragma unroll
    for (int j = 0; j < G; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = (tid + 256 * j) * 4 + k;
            s1[j][k] = 0.0; s2[j][k] = 0.0;
            if (c < cols) { mu[j][k] = (float)mean[c]; den[j][k] = sqrtf((float)var[c] + eps); }
            else { mu[j][k] = 0.f; den[j][k] = 1.f; }
        }
    // the gather indices of this block's rows go through LDS once (a dependent index load per row would put a full
    // memory latency in front of every row), then rows are processed two at a time so two rows' loads are in flight
    __shared__ long long s_src[kRmsRowsPerPass];
    for (int rb = r0; rb < r1; rb += kRmsRowsPerPass) {
        const int nr = min(kRmsRowsPerPass, r1 - rb);
        __syncthreads();
        if ((int)threadIdx.x < nr) s_src[threadIdx.x] = row_idx ? row_idx[rb + threadIdx.x] : (long long)(rb + threadIdx.x);
        __syncthreads();
        for (int q = hf * kRmsRowsInFlight; q < nr; q += 2 * kRmsRowsInFlight) {
            // kRmsRowsInFlight rows' loads are issued before the first is consumed: a workgroup's row loop is latency-bound
            // (one 3.7 KB row per memory round trip otherwise)
            float4 v[kRmsRowsInFlight][G];
#pragma unroll
            for (int h = 0; h < kRmsRowsInFlight; ++h) {
                const float* xr = x + s_src[q + h < nr ? q + h : q] * x_stride;
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    const int c = (tid + 256 * j) * 4;
                    v[h][j] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (c < cols) v[h][j] = *reinterpret_cast<const float4*>(xr + c);
                }
            }
#pragma unroll
            for (int h = 0; h < kRmsRowsInFlight; ++h) {
                if (q + h >= nr) break;
                float* yr = OUT == 2 ? nullptr : y + (long long)(rb + q + h) * y_stride;
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    const int c = (tid + 256 * j) * 4;
                    if (c < y_cols) {
                        const float4 vv = v[h][j];
                        if (raw && c < cols) *reinterpret_cast<float4*>(raw + (long long)(rb + q + h) * raw_stride + c) = vv;     // the rollout's record of the raw row
                        float in[4] = {vv.x, vv.y, vv.z, vv.w}, o[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            if (c + k < cols) {
                                s1[j][k] += (double)in[k]; s2[j][k] += (double)in[k] * (double)in[k];
                                o[k] = mode == 0 ? clampf((in[k] - mu[j][k]) / den[j][k], -clip, clip) : den[j][k] * clampf(in[k], -clip, clip) + mu[j][k];
                            } else {
                                o[k] = 0.f;
                            }
                        }
                        if constexpr (OUT == 2) {
                            *reinterpret_cast<uint2*>(planes + (long long)(rb + q + h) * planes_ld + c) = make_uint2(split_pack_rn(o[0], o[1]), split_pack_rn(o[2], o[3]));
                        } else {
                            *reinterpret_cast<float4*>(yr + c) = make_float4(o[0], o[1], o[2], o[3]);
                        }
                        if constexpr (OUT == 1) {
                            unsigned a0, a1, a2, b0, b1, b2;
                            split_pair3(o[0], o[1], a0, a1, a2);
                            split_pair3(o[2], o[3], b0, b1, b2);
                            unsigned short* pr = planes + (long long)(rb + q + h) * planes_ld + c;
                            *reinterpret_cast<uint2*>(pr) = make_uint2(a0, b0);
                            *reinterpret_cast<uint2*>(pr + plane_stride) = make_uint2(a1, b1);
                            *reinterpret_cast<uint2*>(pr + 2 * plane_stride) = make_uint2(a2, b2);
                        }
                    }
                }
            }
        }
    }
    if (partials) {
        __shared__ double s_red[2 * G * 4][256];           // 48 KB: the upper half's sums, [moment, group, column-in-group][owner]
        __syncthreads();
        if (hf == 1) {
#pragma unroll
            for (int j = 0; j < G; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) { s_red[j * 4 + k][tid] = s1[j][k]; s_red[G * 4 + j * 4 + k][tid] = s2[j][k]; }
        }
        __syncthreads();
        if (hf == 0) {
            double* p = partials + (long long)blockIdx.x * 2 * cols;
#pragma unroll
            for (int j = 0; j < G; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = (tid + 256 * j) * 4 + k;
                    if (c < cols) { p[c] = s1[j][k] + s_red[j * 4 + k][tid]; p[cols + c] = s2[j][k] + s_red[G * 4 + j * 4 + k][tid]; }
                }
        }
    }
}

// narrow matrices (cols < 64, e.g. the (B,1) value tensor): threads own rows.
__global__ void __launch_bounds__(256) rms_normalize_narrow_kernel(const float* __restrict__ x, long long x_stride,
                                                                  const long long* __restrict__ row_idx, int rows, int cols,
                                                                  const double* __restrict__ mean, const double* __restrict__ var,
                                                                  float eps, float clip, int mode, float* __restrict__ y,
                                                                  long long y_stride, int y_cols, double* __restrict__ partials) {
    __shared__ double red[2][4];
    const int nblk = gridDim.x;
    const int per = (rows + nblk - 1) / nblk;
    const int r0 = blockIdx.x * per;
    const int r1 = min(rows, r0 + per);
    for (int c = 0; c < cols; ++c) {
        const float mu = (float)mean[c];
        const float den = sqrtf((float)var[c] + eps);
        double s1 = 0.0, s2 = 0.0;
        for (int r = r0 + threadIdx.x; r < r1; r += 256) {
            const long long src = row_idx ? row_idx[r] : (long long)r;
            const float v = x[src * x_stride + c];
            s1 += (double)v; s2 += (double)v * (double)v;
            float o;
            if (mode == 0) o = clampf((v - mu) / den, -clip, clip);
            else o = den * clampf(v, -clip, clip) + mu;
            y[(long long)r * y_stride + c] = o;
        }
        if (partials) {
            s1 = wave_sum(s1); s2 = wave_sum(s2);
            if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; }
            __syncthreads();
            if (threadIdx.x == 0) {
                double* p = partials + (long long)blockIdx.x * 2 * cols;
                p[c] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
                p[cols + c] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
            }
            __syncthreads();
        }
    }
    for (int r = r0 + threadIdx.x; r < r1; r += 256)
        for (int c = cols; c < y_cols; ++c) y[(long long)r * y_stride + c] = 0.f;
}

__global__ void __launch_bounds__(1024) rms_update_kernel(double* __restrict__ mean, double* __restrict__ var, double* __restrict__ count_out,
                                                        const double* __restrict__ partials, int nblk, int cols, double count,
                                                        double n) {
    // 16 columns x 64 groups over the partial blocks per workgroup (it was 64 x 16: 15 workgroups for the 934-column policy observation, each
    // walking 32 dependent rounds of loads -- 13 us, latency-bound; 59 workgroups of 8 rounds now).  [r5]
    __shared__ double red[2][64][16];
    const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double s1 = 0.0, s2 = 0.0;
    if (c < cols) {
        for (int b = pl; b < nblk; b += 64) {
            s1 += partials[(long long)b * 2 * cols + c];
            s2 += partials[(long long)b * 2 * cols + cols + c];
        }
    }
    red[0][pl][cl] = s1; red[1][pl][cl] = s2;
    __syncthreads();
    if (pl == 0 && c < cols) {
        s1 = 0.0; s2 = 0.0;
#pragma unroll
        for (int g = 0; g < 64; ++g) { s1 += red[0][g][cl]; s2 += red[1][g][cl]; }
        const double bm = s1 / n;
        // unbiased variance like torch.var: sum (x - mean)^2 / (n - 1)
        double bv = (s2 - n * bm * bm) / (n - 1.0);
        if (bv < 0.0) bv = 0.0;
        const double m = mean[c], v = var[c];
        const double delta = bm - m;
        const double tot = count + n;
        const double new_mean = m + delta * n / tot;
        const double m2 = v * count + bv * n + delta * delta * count * n / tot;
        mean[c] = new_mean;
        var[c] = m2 / tot;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && count_out) *count_out = count + n;
}

// ------------------------------------------------------------------------------------------------
// Gaussian policy head, rollout (eval) branch.  16 lanes per sample.
// ------------------------------------------------------------------------------------------------
constexpr int kLanesPerSample = 16;
constexpr float kHalfLog2Pi = 0.91893853320467274178f;  // 0.5 * ln(2 pi)

__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, kLanesPerSample);
    return v;
}

__global__ void __launch_bounds__(256) policy_sample_kernel(const float* __restrict__ mu, long long mu_stride, const float* __restrict__ logstd,
                                                           const float* __restrict__ noise, long long noise_stride,
                                                           const float* __restrict__ value_raw, long long value_stride,
                                                           const double* __restrict__ vmean, const double* __restrict__ vvar, int rows,
                                                           int A, float* __restrict__ actions, long long actions_stride,
                                                           float* __restrict__ sigmas, long long sigmas_stride, float* __restrict__ neglogp,
                                                           long long neglogp_stride, float* __restrict__ values, long long values_stride,
                                                           float* __restrict__ mus_out, long long mus_out_stride) {
    const int g = (blockIdx.x * 256 + threadIdx.x) / kLanesPerSample;
    const int l = threadIdx.x % kLanesPerSample;
    if (g >= rows) return;
    float quad = 0.f, lsum = 0.f;
    for (int j = l; j < A; j += kLanesPerSample) {
        const float ls = logstd[j];
        const float sg = expf(ls);
        const float m = mu[(long long)g * mu_stride + j];
        const float a = m + sg * noise[(long long)g * noise_stride + j];   // Normal(mu, sigma).sample()
        const float zz = (a - m) / sg;
        quad += zz * zz;
        lsum += ls;
        actions[(long long)g * actions_stride + j] = a;
        if (mus_out) mus_out[(long long)g * mus_out_stride + j] = m;
        if (sigmas) sigmas[(long long)g * sigmas_stride + j] = m * 0.0f + sg;   // amp_network_builder.py:142-148
    }
    quad = group16_sum(quad);
    lsum = group16_sum(lsum);
    if (l == 0) {
        neglogp[(long long)g * neglogp_stride] = 0.5f * quad + kHalfLog2Pi * (float)A + lsum;
        if (values) {
            float v = value_raw[(long long)g * value_stride];
            if (vmean) v = sqrtf((float)vvar[0] + 1e-5f) * clampf(v, -5.f, 5.f) + (float)vmean[0];
            values[(long long)g * values_stride] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// PPO losses + gradients w.r.t. (mu, value).  16 lanes per sample, 16 samples per 256-thread block.
// HOIST [r6]: what depends on the action column only -- sigma = exp(logstd), sigma^2, the KL's log(s1 / sigma + 1e-5) and 2 (s1^2 + 1e-5), the row
// sum of logstd -- is evaluated once per lane instead of once per sample (three exp, a log and a divide per element were two thirds of the kernel's
// instructions), and a sample's mu / action values stay in registers between the loss pass and the gradient pass.  Same expressions on the same
// values in the same order: the results are bit-identical to the plain form, which stays for num_actions > 16 * kLossCols.
// ------------------------------------------------------------------------------------------------
constexpr int kLossCols = 10;          // hoisted columns per lane: num_actions <= 160 (SMPL 69, SMPL-X 153)

template <bool HOIST>
__global__ void __launch_bounds__(256) ppo_loss_kernel(const pulse_ppo_loss_args a) {
    __shared__ float red[5][16];
    const int samples_per_block = 256 / kLanesPerSample;
    const int slot = threadIdx.x / kLanesPerSample;
    const int l = threadIdx.x % kLanesPerSample;
    const int A = a.num_actions;
    const float invB = 1.0f / (float)a.rows;
    float acc_a = 0.f, acc_c = 0.f, acc_b = 0.f, acc_clip = 0.f, acc_kl = 0.f;
    float c_sg[kLossCols], c_sg2[kLossCols], c_klog[kLossCols], c_kden[kLossCols];
    float c_lsum = 0.f;
    if constexpr (HOIST) {
#pragma unroll
        for (int t = 0; t < kLossCols; ++t) {
            const int j = l + kLanesPerSample * t;
            c_sg[t] = 1.f; c_sg2[t] = 1.f; c_klog[t] = 0.f; c_kden[t] = 1.f;
            if (j < A) {
                const float ls = a.logstd[j], sg = expf(ls);
                const float s1 = expf(a.old_logstd[j]);
                c_sg[t] = sg; c_sg2[t] = sg * sg;
                c_klog[t] = logf(s1 / sg + 1e-5f);
                c_kden[t] = 2.0f * (s1 * s1 + 1e-5f);
                c_lsum += ls;
            }
        }
        c_lsum = group16_sum(c_lsum);
    }
    for (int base = blockIdx.x * samples_per_block; base < a.rows; base += gridDim.x * samples_per_block) {
        const int i = base + slot;
        if (i >= a.rows) continue;
        const long long d = a.idx ? a.idx[i] : (long long)i;
        const float* mu = a.mu + (long long)i * a.mu_stride;
        const float* act = a.actions + d * a.actions_stride;
        const float* omu = a.old_mu + d * a.old_mu_stride;
        float quad = 0.f, lsum = 0.f, bl = 0.f, kl = 0.f;
        float r_m[kLossCols], r_da[kLossCols];                  // HOIST: mu and (action - mu) of this lane's columns
        if constexpr (HOIST) {
            float r_o[kLossCols];
#pragma unroll
            for (int t = 0; t < kLossCols; ++t) {               // all loads of the sample first: one memory round trip, not one per column
                const int j = l + kLanesPerSample * t;
                r_m[t] = 0.f; r_da[t] = 0.f; r_o[t] = 0.f;
                if (j < A) { r_m[t] = mu[j]; r_da[t] = act[j]; r_o[t] = omu[j]; }
            }
#pragma unroll
            for (int t = 0; t < kLossCols; ++t) {
                const int j = l + kLanesPerSample * t;
                if (j < A) {
                    const float m = r_m[t];
                    r_da[t] = r_da[t] - m;
                    const float zz = r_da[t] / c_sg[t];
                    quad += zz * zz;
                    if (a.has_bounds_loss) {
                        const float hi = fmaxf(m - 1.0f, 0.f), lo = fminf(m + 1.0f, 0.f);
                        bl += lo * lo + hi * hi;
                    }
                    const float dm = r_o[t] - m;
                    kl += c_klog[t] + (c_sg2[t] + dm * dm) / c_kden[t] + (-0.5f);
                }
            }
            lsum = c_lsum;
            quad = group16_sum(quad); bl = group16_sum(bl); kl = group16_sum(kl);
        } else {
            for (int j = l; j < A; j += kLanesPerSample) {
                const float ls = a.logstd[j], sg = expf(ls);
                const float m = mu[j];
                const float zz = (act[j] - m) / sg;
                quad += zz * zz;
                lsum += ls;
                if (a.has_bounds_loss) {
                    const float hi = fmaxf(m - 1.0f, 0.f), lo = fminf(m + 1.0f, 0.f);
                    bl += lo * lo + hi * hi;
                }
                // rl_games policy_kl(p0 = new, p1 = old)
                const float s1 = expf(a.old_logstd[j]);
                const float dm = omu[j] - m;
                kl += logf(s1 / sg + 1e-5f) + (sg * sg + dm * dm) / (2.0f * (s1 * s1 + 1e-5f)) + (-0.5f);
            }
            quad = group16_sum(quad); lsum = group16_sum(lsum); bl = group16_sum(bl); kl = group16_sum(kl);
        }
        const float nlp = 0.5f * quad + kHalfLog2Pi * (float)A + lsum;
        const float adv = a.advantages[d];
        const float ratio = expf(a.old_neglogp[d] - nlp);
        const float rc = clampf(ratio, 1.0f - a.e_clip, 1.0f + a.e_clip);
        const float l1 = -(adv * ratio), l2 = -(adv * rc);
        const float a_loss = fmaxf(l1, l2);
        // d a_loss / d ratio: -adv where the unclipped branch is active (or tied inside the clip range)
        const bool inside = (ratio >= 1.0f - a.e_clip) && (ratio <= 1.0f + a.e_clip);
        const float g_ratio = (inside || l1 > l2) ? -adv : 0.f;
        const float g_nlp = g_ratio * (-ratio);                      // d ratio / d nlp = -ratio
        const float clipped = fabsf(ratio - 1.0f) > a.e_clip ? 1.f : 0.f;
        // critic
        const float v = a.value[(long long)i * a.value_stride];
        const float ret = a.returns[d];
        float c_loss, g_v;
        if (a.clip_value) {
            const float vp = a.old_values[d];
            const float dv = v - vp;
            const float vpc = vp + clampf(dv, -a.e_clip, a.e_clip);
            const float e1 = (v - ret) * (v - ret), e2 = (vpc - ret) * (vpc - ret);
            c_loss = fmaxf(e1, e2);
            if (e1 >= e2) g_v = 2.0f * (v - ret);
            else g_v = (fabsf(dv) <= a.e_clip) ? 2.0f * (vpc - ret) : 0.f;
        } else {
            c_loss = (ret - v) * (ret - v);
            g_v = -2.0f * (ret - v);
        }
        // gradients (mean over the batch folded in)
        float* dmu = a.dmu + (long long)i * a.dmu_stride;
        if constexpr (HOIST) {
#pragma unroll
            for (int t = 0; t < kLossCols; ++t) {
                const int j = l + kLanesPerSample * t;
                if (j < A) {
                    const float m = r_m[t];
                    float gmu = g_nlp * (-r_da[t] / c_sg2[t]);
                    if (a.has_bounds_loss) gmu += a.bounds_loss_coef * (2.0f * fmaxf(m - 1.0f, 0.f) + 2.0f * fminf(m + 1.0f, 0.f));
                    dmu[j] = gmu * invB;
                    if (a.dmu16) reinterpret_cast<unsigned short*>(a.dmu16)[(long long)i * a.dmu16_stride + j] = (unsigned short)(split_pack_rn(gmu * invB, 0.f) & 0xffffu);
                }
            }
        } else {
            for (int j = l; j < A; j += kLanesPerSample) {
                const float sg = expf(a.logstd[j]);
                const float m = mu[j];
                float gmu = g_nlp * (-(act[j] - m) / (sg * sg));
                if (a.has_bounds_loss) gmu += a.bounds_loss_coef * (2.0f * fmaxf(m - 1.0f, 0.f) + 2.0f * fminf(m + 1.0f, 0.f));
                dmu[j] = gmu * invB;
                if (a.dmu16) reinterpret_cast<unsigned short*>(a.dmu16)[(long long)i * a.dmu16_stride + j] = (unsigned short)(split_pack_rn(gmu * invB, 0.f) & 0xffffu);
            }
        }
        if (l == 0) {
            a.dvalue[(long long)i * a.dvalue_stride] = a.critic_coef * g_v * invB;
            if (a.dvalue16) reinterpret_cast<unsigned short*>(a.dvalue16)[(long long)i * a.dvalue16_stride] = (unsigned short)(split_pack_rn(a.critic_coef * g_v * invB, 0.f) & 0xffffu);
            acc_a += a_loss; acc_c += c_loss; acc_b += bl; acc_clip += clipped; acc_kl += kl;
        }
    }
    if (l == 0) { red[0][slot] = acc_a; red[1][slot] = acc_c; red[2][slot] = acc_b; red[3][slot] = acc_clip; red[4][slot] = acc_kl; }
    __syncthreads();
    if (threadIdx.x < 8) {
        float s = 0.f;
        if (threadIdx.x < 5)
            for (int k = 0; k < 16; ++k) s += red[threadIdx.x][k];
        a.partials[(long long)blockIdx.x * 8 + threadIdx.x] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// advantages
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) adv_moments_kernel(const float* __restrict__ ret, const float* __restrict__ val, long long count,
                                                         float* __restrict__ adv, double* __restrict__ partials) {
    __shared__ double red[2][4];
    double s1 = 0.0, s2 = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256) {
        const float a = ret[i] - val[i];
        adv[i] = a;
        s1 += (double)a; s2 += (double)a * (double)a;
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        partials[2 * blockIdx.x + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

__global__ void __launch_bounds__(256) adv_normalize_kernel(float* __restrict__ adv, long long count, const double* __restrict__ partials, int nblk) {
    __shared__ float s_mean, s_den;
    if (threadIdx.x == 0) {
        double s1 = 0.0, s2 = 0.0;
        for (int b = 0; b < nblk; ++b) { s1 += partials[2 * b]; s2 += partials[2 * b + 1]; }
        const double n = (double)count;
        const double m = s1 / n;
        double var = (s2 - n * m * m) / (n - 1.0);
        if (var < 0.0) var = 0.0;
        s_mean = (float)m;
        s_den = (float)sqrt(var) + 1e-8f;
    }
    __syncthreads();
    const float m = s_mean, den = s_den;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256) adv[i] = (adv[i] - m) / den;
}

// ------------------------------------------------------------------------------------------------
// gradient norm + Adam over the flat parameter buffer
// ------------------------------------------------------------------------------------------------
// AMPAgent._disc_loss head (phc/learning/amp_agent.py:895-952) on the logits of the 3b stacked rows [agent | replay | demo]:
// prediction loss 0.5 (BCEWithLogits(agent U replay, 0) + BCEWithLogits(demo, 1)), its gradient w.r.t. every logit (times
// ``scale`` = disc_coef / world_size), the two accuracies and the two logit means -- one launch instead of ~45 tiny tensor ops and an
// autograd pass per minibatch.  One workgroup: 3b logits are a few hundred KB; reductions in double, fixed order.
__global__ void __launch_bounds__(1024) disc_head_kernel(const float* __restrict__ logits, long long ls, int b, float scale,
                                                         float* __restrict__ dlogits, long long ds, float* __restrict__ stats,
                                                         unsigned short* __restrict__ dl16, long long ds16, float* __restrict__ bias_grad) {
    __shared__ double red[16][7];
    const int n = 3 * b, na = 2 * b;
    double acc[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0}; // BCE agent, BCE demo, #agent < 0, #demo > 0, sum agent logit, sum demo logit, sum of d loss / d logit
    const float ga = scale * 0.5f / (float)na, gd = scale * 0.5f / (float)b;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float x = logits[(long long)i * ls];
        const float sp = fmaxf(x, 0.f) + log1pf(expf(-fabsf(x)));        // softplus(x) = BCEWithLogits(x, 0); BCE(x, 1) = softplus(x) - x
        const float sg = 1.f / (1.f + expf(-x));
        if (i < na) {
            acc[0] += (double)sp; acc[2] += x < 0.f ? 1.0 : 0.0; acc[4] += (double)x;
            const float gl = ga * sg;
            if (dlogits) dlogits[(long long)i * ds] = gl;
            const unsigned q = split_pack_rn(gl, 0.f) & 0xffffu;
            if (dl16) dl16[(long long)i * ds16] = (unsigned short)q;
            acc[6] += dl16 ? (double)split_bitsf(q << 16) : (double)gl;      // the logit bias' gradient sums what the backward pass reads
        } else {
            acc[1] += (double)(sp - x); acc[3] += x > 0.f ? 1.0 : 0.0; acc[5] += (double)x;
            const float gl = gd * (sg - 1.f);
            if (dlogits) dlogits[(long long)i * ds] = gl;
            const unsigned q = split_pack_rn(gl, 0.f) & 0xffffu;
            if (dl16) dl16[(long long)i * ds16] = (unsigned short)q;
            acc[6] += dl16 ? (double)split_bitsf(q << 16) : (double)gl;
        }
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) acc[k] = wave_sum(acc[k]);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 7; ++k) red[threadIdx.x >> 6][k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[7];
        for (int k = 0; k < 7; ++k) { t[k] = 0.0; for (int w = 0; w < 16; ++w) t[k] += red[w][k]; }
        if (bias_grad) *bias_grad = (float)t[6];
        const double la = t[0] / na, ld = t[1] / b;
        stats[0] = (float)(0.5 * (la + ld)); stats[1] = (float)la; stats[2] = (float)ld;
        stats[3] = (float)(t[2] / na); stats[4] = (float)(t[3] / b); stats[5] = (float)(t[4] / na); stats[6] = (float)(t[5] / b); stats[7] = 0.f;
    }
}

__global__ void __launch_bounds__(256) sqnorm_partial_kernel(const float* __restrict__ x, long long count, float* __restrict__ partials) {
    __shared__ float red[4];
    float s = 0.f;
    const long long n4 = count >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = x4[i];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0) for (long long i = (n4 << 2) + threadIdx.x; i < count; i += 256) s += x[i] * x[i];
    s = wave_sumf(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__device__ __forceinline__ void adam_body(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                          long long count, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                          float max_norm, const float* __restrict__ sq_partials, int npart, float* __restrict__ norm_out) {
    __shared__ float red[4];
    __shared__ float s_coef;
    float coef = 1.0f;
    if (sq_partials) {
        double s = 0.0;
        for (int i = threadIdx.x; i < npart; i += 256) s += (double)sq_partials[i];
        s = wave_sum(s);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = (float)s;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float norm = sqrtf(red[0] + red[1] + red[2] + red[3]);
            if (norm_out && blockIdx.x == 0) *norm_out = norm;
            float c = 1.0f;
            if (max_norm > 0.f) c = fminf(max_norm / (norm + 1e-6f), 1.0f);   // clip_grad_norm_
            s_coef = c;
        }
        __syncthreads();
        coef = s_coef;
    }
    const float step_size = lr / bc1;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256) {
        float gi = g[i] * coef;
        const float pi = p[i];
        if (wd != 0.f) gi += wd * pi;
        const float mi = m[i] + (gi - m[i]) * (1.0f - b1);           // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = v[i] * b2 + (1.0f - b2) * (gi * gi);        // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - step_size * (mi / denom);
    }
}

__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                  long long count, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                                  float max_norm, const float* __restrict__ sq_partials, int npart, float* __restrict__ norm_out) {
    adam_body(p, g, m, v, count, lr, b1, b2, eps, wd, bc1, bc2_sqrt, max_norm, sq_partials, npart, norm_out);
}

// the same step over up to four flat buffers in ONE launch (blockIdx.y = buffer): one optimiser over several parameter groups
// (AMPAgent: policy + discriminator, amp_agent.py:136-140) with the joint gradient-norm clip; per element exactly adam_kernel's arithmetic
struct AdamGroups { float* p[4]; const float* g[4]; float* m[4]; float* v[4]; long long count[4]; };
__global__ void __launch_bounds__(256) adam_multi_kernel(const AdamGroups a, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                                        float max_norm, const float* __restrict__ sq_partials, int npart, float* __restrict__ norm_out) {
    const int k = blockIdx.y;
    adam_body(a.p[k], a.g[k], a.m[k], a.v[k], a.count[k], lr, b1, b2, eps, wd, bc1, bc2_sqrt, max_norm, sq_partials, npart, k == 0 ? norm_out : nullptr);
}

}  // namespace pulse

using namespace pulse;

// the normaliser's wide-row kernel with as many column groups per thread as the row needs (1 / 2 / 3 groups: 62 - 70 / ~100 / 141 - 146 VGPRs)
template <int OUT>
static void launch_rms_vec4(int num_blocks, hipStream_t st, const float* x, long long x_stride, const long long* row_idx, int rows, int cols,
                            const double* mean, const double* var, float eps, float clip, int mode, float* y, long long y_stride, int y_cols,
                            double* partials, unsigned short* planes, long long plane_stride, long long planes_ld, float* raw = nullptr,
                            long long raw_stride = 0) {
    if (y_cols <= 1024)
        hipLaunchKernelGGL((rms_normalize_vec4_kernel<OUT, 1>), dim3(num_blocks), dim3(512), 0, st, x, x_stride, row_idx, rows, cols, mean, var, eps, clip, mode,
                           y, y_stride, y_cols, partials, planes, plane_stride, planes_ld, raw, raw_stride);
    else if (y_cols <= 2048)
        hipLaunchKernelGGL((rms_normalize_vec4_kernel<OUT, 2>), dim3(num_blocks), dim3(512), 0, st, x, x_stride, row_idx, rows, cols, mean, var, eps, clip, mode,
                           y, y_stride, y_cols, partials, planes, plane_stride, planes_ld, raw, raw_stride);
    else
        hipLaunchKernelGGL((rms_normalize_vec4_kernel<OUT, 3>), dim3(num_blocks), dim3(512), 0, st, x, x_stride, row_idx, rows, cols, mean, var, eps, clip, mode,
                           y, y_stride, y_cols, partials, planes, plane_stride, planes_ld, raw, raw_stride);
}

extern "C" {

int pulse_rms_normalize(const float* x, int64_t x_stride, const int64_t* row_idx, int32_t rows, int32_t cols, const double* mean,
                        const double* var, float eps, float clip, int32_t mode, float* y, int64_t y_stride, int32_t y_cols,
                        double* moment_partials, int32_t num_blocks, pulse_stream_t s) {
    PULSE_REQUIRE(rows >= 0 && cols >= 0, "pulse_rms_normalize: negative size");
    if (rows == 0 || cols == 0) return PULSE_OK;
    PULSE_REQUIRE(x && y && mean && var, "pulse_rms_normalize: null pointer");
    PULSE_REQUIRE(cols <= 256 * kRmsMaxColsPerThread, "pulse_rms_normalize: cols %d > %d", cols, 256 * kRmsMaxColsPerThread);
    PULSE_REQUIRE(num_blocks >= 1, "pulse_rms_normalize: num_blocks < 1");
    PULSE_REQUIRE(y_cols >= cols && y_stride >= y_cols && x_stride >= cols, "pulse_rms_normalize: bad pitches");
    PULSE_REQUIRE(mode == 0 || mode == 1, "pulse_rms_normalize: bad mode");
    const bool vec_ok = cols >= 64 && cols <= 256 * 4 * kRmsVecGroups && (x_stride % 4) == 0 && (y_stride % 4) == 0 && (y_cols % 4) == 0 &&
                        (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 &&
                        x_stride >= ((cols + 3) & ~3);
    if (vec_ok)
        launch_rms_vec4<0>(num_blocks, as_stream(s), x, (long long)x_stride, (const long long*)row_idx, rows, cols, mean, var, eps, clip, mode, y,
                           (long long)y_stride, y_cols, moment_partials, nullptr, 0LL, 0LL);
    else if (cols >= 64)
        hipLaunchKernelGGL(rms_normalize_wide_kernel, dim3(num_blocks), dim3(256), 0, as_stream(s), x, (long long)x_stride,
                           (const long long*)row_idx, rows, cols, mean, var, eps, clip, mode, y, (long long)y_stride, y_cols, moment_partials);
    else
        hipLaunchKernelGGL(rms_normalize_narrow_kernel, dim3(num_blocks), dim3(256), 0, as_stream(s), x, (long long)x_stride,
                           (const long long*)row_idx, rows, cols, mean, var, eps, clip, mode, y, (long long)y_stride, y_cols, moment_partials);
    return check_launch("pulse_rms_normalize");
}

int pulse_rms_normalize_copy(const float* x, int64_t x_stride, const int64_t* row_idx, int32_t rows, int32_t cols, const double* mean,
                             const double* var, float eps, float clip, float* y, int64_t y_stride, int32_t y_cols, double* moment_partials,
                             int32_t num_blocks, float* raw_out, int64_t raw_stride, pulse_stream_t s) {
    PULSE_REQUIRE(rows >= 0 && cols >= 0, "pulse_rms_normalize_copy: negative size");
    if (rows == 0 || cols == 0) return PULSE_OK;
    PULSE_REQUIRE(x && y && mean && var && raw_out, "pulse_rms_normalize_copy: null pointer");
    PULSE_REQUIRE(num_blocks >= 1, "pulse_rms_normalize_copy: num_blocks < 1");
    PULSE_REQUIRE(y_cols >= cols && y_stride >= y_cols && x_stride >= cols, "pulse_rms_normalize_copy: bad pitches");
    PULSE_REQUIRE(cols >= 64 && y_cols <= 256 * 4 * kRmsVecGroups && (x_stride % 4) == 0 && (y_stride % 4) == 0 && (y_cols % 4) == 0 &&
                  (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 && x_stride >= ((cols + 3) & ~3),
                  "pulse_rms_normalize_copy: needs the wide-row form (64 <= cols, y_cols <= %d, 16-byte aligned rows)", 256 * 4 * kRmsVecGroups);
    PULSE_REQUIRE((raw_stride % 4) == 0 && raw_stride >= ((cols + 3) & ~3) && (reinterpret_cast<uintptr_t>(raw_out) & 15) == 0,
                  "pulse_rms_normalize_copy: raw_out rows must be 16-byte aligned and hold cols rounded up to 4 floats");
    launch_rms_vec4<0>(num_blocks, as_stream(s), x, (long long)x_stride, (const long long*)row_idx, rows, cols, mean, var, eps, clip, 0, y,
                       (long long)y_stride, y_cols, moment_partials, nullptr, 0LL, 0LL, raw_out, (long long)raw_stride);
    return check_launch("pulse_rms_normalize_copy");
}

int pulse_rms_normalize_planes(const float* x, int64_t x_stride, const int64_t* row_idx, int32_t rows, int32_t cols, const double* mean,
                               const double* var, float eps, float clip, float* y, int64_t y_stride, int32_t y_cols, double* moment_partials,
                               int32_t num_blocks, void* planes, int64_t plane_stride, int64_t planes_ld, pulse_stream_t s) {
    PULSE_REQUIRE(rows >= 0 && cols >= 0, "pulse_rms_normalize_planes: negative size");
    if (rows == 0 || cols == 0) return PULSE_OK;
    PULSE_REQUIRE(x && y && mean && var && planes, "pulse_rms_normalize_planes: null pointer");
    PULSE_REQUIRE(num_blocks >= 1, "pulse_rms_normalize_planes: num_blocks < 1");
    PULSE_REQUIRE(y_cols >= cols && y_stride >= y_cols && x_stride >= cols, "pulse_rms_normalize_planes: bad pitches");
    PULSE_REQUIRE(cols >= 64 && cols <= 256 * 4 * kRmsVecGroups && (x_stride % 4) == 0 && (y_stride % 4) == 0 && (y_cols % 4) == 0 &&
                  (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 && x_stride >= ((cols + 3) & ~3),
                  "pulse_rms_normalize_planes: needs the wide-row form (64 <= cols <= %d, 16-byte aligned rows)", 256 * 4 * kRmsVecGroups);
    // the planes cover the same y_cols columns as y (columns [cols, y_cols) are written as zeros: the GEMM's k padding)
    PULSE_REQUIRE((y_cols % 32) == 0 && planes_ld >= y_cols && (planes_ld % 8) == 0 && (plane_stride % 8) == 0 && plane_stride >= (int64_t)rows * planes_ld &&
                  (reinterpret_cast<uintptr_t>(planes) & 15) == 0,
                  "pulse_rms_normalize_planes: y_cols must be a multiple of 32 (zero-padded k extent), planes rows 16-byte aligned and covering it");
    launch_rms_vec4<1>(num_blocks, as_stream(s), x, (long long)x_stride, (const long long*)row_idx, rows, cols, mean, var, eps, clip, 0, y,
                       (long long)y_stride, y_cols, moment_partials, reinterpret_cast<unsigned short*>(planes), (long long)plane_stride, (long long)planes_ld);
    return check_launch("pulse_rms_normalize_planes");
}

int pulse_rms_normalize_b16(const float* x, int64_t x_stride, const int64_t* row_idx, int32_t rows, int32_t cols, const double* mean,
                            const double* var, float eps, float clip, void* y16, int64_t y_stride, int32_t y_cols, double* moment_partials,
                            int32_t num_blocks, pulse_stream_t s) {
    PULSE_REQUIRE(rows >= 0 && cols >= 0, "pulse_rms_normalize_b16: negative size");
    if (rows == 0 || cols == 0) return PULSE_OK;
    PULSE_REQUIRE(x && y16 && mean && var, "pulse_rms_normalize_b16: null pointer");
    PULSE_REQUIRE(num_blocks >= 1, "pulse_rms_normalize_b16: num_blocks < 1");
    PULSE_REQUIRE(y_cols >= cols && y_stride >= y_cols && x_stride >= cols, "pulse_rms_normalize_b16: bad pitches");
    PULSE_REQUIRE(cols >= 64 && y_cols <= 256 * 4 * kRmsVecGroups && (x_stride % 4) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && x_stride >= ((cols + 3) & ~3),
                  "pulse_rms_normalize_b16: needs the wide-row form (64 <= cols, y_cols <= %d, 16-byte aligned input rows)", 256 * 4 * kRmsVecGroups);
    PULSE_REQUIRE((y_cols % 4) == 0 && (y_stride % 4) == 0 && (reinterpret_cast<uintptr_t>(y16) & 7) == 0, "pulse_rms_normalize_b16: output rows must be 8-byte aligned, y_cols a multiple of 4");
    launch_rms_vec4<2>(num_blocks, as_stream(s), x, (long long)x_stride, (const long long*)row_idx, rows, cols, mean, var, eps, clip, 0, nullptr, 0LL,
                       y_cols, moment_partials, reinterpret_cast<unsigned short*>(y16), 0LL, (long long)y_stride);
    return check_launch("pulse_rms_normalize_b16");
}

int pulse_rms_update(double* mean, double* var, double* count_out, const double* moment_partials, int32_t num_blocks, int32_t cols,
                     double count_old, double batch_count, pulse_stream_t s) {
    PULSE_REQUIRE(cols >= 0 && num_blocks >= 1, "pulse_rms_update: bad sizes");
    if (cols == 0) return PULSE_OK;
    PULSE_REQUIRE(mean && var && moment_partials, "pulse_rms_update: null pointer");
    PULSE_REQUIRE(batch_count >= 2.0, "pulse_rms_update: batch of %g rows has no unbiased variance", batch_count);
    hipLaunchKernelGGL(rms_update_kernel, dim3((cols + 15) / 16), dim3(1024), 0, as_stream(s), mean, var, count_out, moment_partials, num_blocks,
                       cols, count_old, batch_count);
    return check_launch("pulse_rms_update");
}

int pulse_policy_sample(const float* mu, int64_t mu_stride, const float* logstd, const float* noise, int64_t noise_stride,
                        const float* value_raw, int64_t value_stride, const double* value_mean, const double* value_var, int32_t rows,
                        int32_t num_actions, float* actions, int64_t actions_stride, float* sigmas, int64_t sigmas_stride, float* neglogp,
                        int64_t neglogp_stride, float* values, int64_t values_out_stride, float* mus_out, int64_t mus_out_stride,
                        pulse_stream_t s) {
    PULSE_REQUIRE(rows >= 0 && num_actions >= 1, "pulse_policy_sample: bad sizes");
    if (rows == 0) return PULSE_OK;
    PULSE_REQUIRE(mu && logstd && noise && actions && neglogp, "pulse_policy_sample: null pointer");
    PULSE_REQUIRE(values == nullptr || value_raw != nullptr, "pulse_policy_sample: values requested without value_raw");
    PULSE_REQUIRE((value_mean == nullptr) == (value_var == nullptr), "pulse_policy_sample: value mean/var must come together");
    const long long threads = (long long)rows * kLanesPerSample;
    hipLaunchKernelGGL(policy_sample_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, as_stream(s), mu, (long long)mu_stride,
                       logstd, noise, (long long)noise_stride, value_raw, (long long)value_stride, value_mean, value_var, rows, num_actions,
                       actions, (long long)actions_stride, sigmas, (long long)sigmas_stride, neglogp, (long long)neglogp_stride, values,
                       (long long)values_out_stride, mus_out, (long long)mus_out_stride);
    return check_launch("pulse_policy_sample");
}

int pulse_sizeof_ppo_loss_args(void) { return (int)sizeof(pulse_ppo_loss_args); }

int pulse_ppo_loss(const pulse_ppo_loss_args* args, pulse_stream_t s) {
    PULSE_REQUIRE(args != nullptr, "pulse_ppo_loss: null args");
    const pulse_ppo_loss_args& a = *args;
    PULSE_REQUIRE(a.rows >= 1 && a.num_actions >= 1 && a.num_blocks >= 1, "pulse_ppo_loss: bad sizes");
    PULSE_REQUIRE(a.mu && a.value && a.logstd && a.old_logstd && a.actions && a.old_mu && a.old_neglogp && a.advantages && a.returns &&
                      a.dmu && a.dvalue && a.partials,
                  "pulse_ppo_loss: null pointer");
    PULSE_REQUIRE(!a.clip_value || a.old_values, "pulse_ppo_loss: clip_value needs old_values");
    if (a.num_actions <= kLanesPerSample * kLossCols && gemm_option(8) == 0)
        hipLaunchKernelGGL(ppo_loss_kernel<true>, dim3(a.num_blocks), dim3(256), 0, as_stream(s), a);
    else
        hipLaunchKernelGGL(ppo_loss_kernel<false>, dim3(a.num_blocks), dim3(256), 0, as_stream(s), a);
    return check_launch("pulse_ppo_loss");
}

int pulse_advantage_moments(const float* returns, const float* values, int64_t count, float* adv, double* partials, int32_t num_blocks,
                            pulse_stream_t s) {
    PULSE_REQUIRE(count >= 2 && num_blocks >= 1, "pulse_advantage_moments: need >= 2 samples");
    PULSE_REQUIRE(returns && values && adv && partials, "pulse_advantage_moments: null pointer");
    hipLaunchKernelGGL(adv_moments_kernel, dim3(num_blocks), dim3(256), 0, as_stream(s), returns, values, (long long)count, adv, partials);
    return check_launch("pulse_advantage_moments");
}

int pulse_advantage_normalize(float* adv, int64_t count, const double* partials, int32_t num_blocks, pulse_stream_t s) {
    PULSE_REQUIRE(count >= 2 && num_blocks >= 1 && adv && partials, "pulse_advantage_normalize: bad arguments");
    long long blocks = (count + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(adv_normalize_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(s), adv, (long long)count, partials, num_blocks);
    return check_launch("pulse_advantage_normalize");
}

int pulse_sqnorm_partial(const float* x, int64_t count, float* partials, int32_t num_blocks, pulse_stream_t s) {
    PULSE_REQUIRE(count >= 0 && num_blocks >= 1 && x && partials, "pulse_sqnorm_partial: bad arguments");
    PULSE_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, "pulse_sqnorm_partial: x must be 16-byte aligned");
    hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(num_blocks), dim3(256), 0, as_stream(s), x, (long long)count, partials);
    return check_launch("pulse_sqnorm_partial");
}

int pulse_disc_head(const float* logits, int64_t logit_stride, int32_t b, float scale, float* dlogits, int64_t dlogit_stride, float* stats,
                    pulse_stream_t s) {
    PULSE_REQUIRE(b >= 1 && logit_stride >= 1 && dlogit_stride >= 1, "pulse_disc_head: bad sizes");
    PULSE_REQUIRE(logits && dlogits && stats, "pulse_disc_head: null pointer");
    hipLaunchKernelGGL(disc_head_kernel, dim3(1), dim3(1024), 0, as_stream(s), logits, (long long)logit_stride, b, scale, dlogits,
                       (long long)dlogit_stride, stats, (unsigned short*)nullptr, 0LL, (float*)nullptr);
    return check_launch("pulse_disc_head");
}

int pulse_disc_head_b16(const float* logits, int64_t logit_stride, int32_t b, float scale, float* dlogits, int64_t dlogit_stride, void* dlogits16,
                        int64_t dlogit16_stride, float* stats, float* bias_grad, pulse_stream_t s) {
    PULSE_REQUIRE(b >= 1 && logit_stride >= 1, "pulse_disc_head_b16: bad sizes");
    PULSE_REQUIRE(logits && stats && (dlogits || dlogits16), "pulse_disc_head_b16: null pointer");
    PULSE_REQUIRE((!dlogits || dlogit_stride >= 1) && (!dlogits16 || dlogit16_stride >= 1), "pulse_disc_head_b16: bad strides");
    hipLaunchKernelGGL(disc_head_kernel, dim3(1), dim3(1024), 0, as_stream(s), logits, (long long)logit_stride, b, scale, dlogits,
                       (long long)dlogit_stride, stats, reinterpret_cast<unsigned short*>(dlogits16), (long long)dlogit16_stride, bias_grad);
    return check_launch("pulse_disc_head_b16");
}

int pulse_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t count, float lr, float beta1, float beta2,
                    float eps, float weight_decay, int32_t step, float max_norm, const float* sqnorm_partials, int32_t num_partials,
                    float* grad_norm_out, pulse_stream_t s) {
    PULSE_REQUIRE(count >= 0 && step >= 1, "pulse_adam_step: bad count / step");
    if (count == 0) return PULSE_OK;
    PULSE_REQUIRE(params && grads && exp_avg && exp_avg_sq, "pulse_adam_step: null pointer");
    PULSE_REQUIRE(sqnorm_partials == nullptr || num_partials >= 1, "pulse_adam_step: bad partial count");
    // bias corrections in double like torch (python floats), handed to the kernel as fp32 scalars
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    long long blocks = (count + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(s), params, grads, exp_avg, exp_avg_sq, (long long)count, lr,
                       beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), max_norm, sqnorm_partials, num_partials, grad_norm_out);
    return check_launch("pulse_adam_step");
}

int pulse_adam_step_multi(int32_t num_groups, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                          const int64_t* counts, float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step, float max_norm,
                          const float* sqnorm_partials, int32_t num_partials, float* grad_norm_out, pulse_stream_t s) {
    PULSE_REQUIRE(num_groups >= 1 && num_groups <= 4 && step >= 1, "pulse_adam_step_multi: 1..4 groups, step >= 1");
    PULSE_REQUIRE(params && grads && exp_avg && exp_avg_sq && counts, "pulse_adam_step_multi: null pointer");
    PULSE_REQUIRE(sqnorm_partials == nullptr || num_partials >= 1, "pulse_adam_step_multi: bad partial count");
    AdamGroups a;
    long long most = 0;
    for (int k = 0; k < 4; ++k) {
        const bool on = k < num_groups;
        a.p[k] = on ? params[k] : nullptr; a.g[k] = on ? grads[k] : nullptr; a.m[k] = on ? exp_avg[k] : nullptr; a.v[k] = on ? exp_avg_sq[k] : nullptr;
        a.count[k] = on ? counts[k] : 0;
        PULSE_REQUIRE(!on || (counts[k] >= 0 && (counts[k] == 0 || (params[k] && grads[k] && exp_avg[k] && exp_avg_sq[k]))), "pulse_adam_step_multi: bad group");
        if (a.count[k] > most) most = a.count[k];
    }
    if (most == 0) return PULSE_OK;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    long long blocks = (most + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)blocks, (unsigned)num_groups), dim3(256), 0, as_stream(s), a, lr, beta1, beta2, eps, weight_decay,
                       (float)bc1, (float)sqrt(bc2), max_norm, sqnorm_partials, num_partials, grad_norm_out);
    return check_launch("pulse_adam_step_multi");
}
}
