// PULSE VAE head algebra in three launches (no autograd tape on the product path).
//
// Replaces the tensor-op chains (and their autograd backward) of
//   AMPZBuilder.Network.form_embedding        phc/learning/amp_network_z_builder.py:79-121   (logvar clamp, re-parameterisation)
//   AMPAgent._optimize_kin                    phc/learning/amp_agent.py:771-849             (action RMSE, KL vs the learned prior,
//                                                                                            AR(1) latent smoothness with seam mask, regulariser)
//   kl_multi                                  phc/learning/loss_functions.py:3-10
// One 32-lane half-wave per minibatch row (lane = latent dimension, embedding_size <= 32), 8 rows per 256-thread workgroup; row
// reductions are 32-lane butterflies, loss sums go through per-workgroup partials that the host-side sum reduces in a fixed order
// (deterministic).  Compiled with -ffp-contract=off: the expressions are written in the reference's operation order.
#include "common.h"

namespace pulse {

constexpr int kVL = 32;   // lanes per row

__device__ __forceinline__ float vsum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, kVL);
    return v;
}
__device__ __forceinline__ float vclamp(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

// ---- form_embedding: z = mu + exp(0.5 clamp(logvar)) eps, written with the self observation into the decoder's (and the critic's)
// concat buffer.  eps == NULL: z = mu (flags.test, amp_network_z_builder.py:94-95).
__global__ void __launch_bounds__(256) vae_embed_kernel(const pulse_vae_embed_args a) {
    const int E = a.embedding_size, S = a.self_obs_size;
    const long long total = (long long)a.rows * (S + E);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / (S + E)), c = (int)(i - (long long)r * (S + E));
        if (c < S) {
            const float v = a.x[(long long)r * a.x_stride + c];
            a.ain[(long long)r * a.ain_stride + c] = v;
            if (a.cin) a.cin[(long long)r * a.cin_stride + c] = v;
        } else {
            const int e = c - S;
            const float* h = a.heads + (long long)r * a.heads_stride;
            const float mu = h[e];
            float lv = h[E + e];
            if (a.clamp_logvar) lv = vclamp(lv, -5.0f, a.clamp_max);
            const float z = a.eps ? mu + expf(0.5f * lv) * a.eps[(long long)r * a.eps_stride + e] : mu;
            a.ain[(long long)r * a.ain_stride + a.z_col + e] = z;
        }
    }
}

// keep-mask of AR(1) error row (sequence s, step i -> i + 1): consecutive progress values and neither step within the first frames
__device__ __forceinline__ bool ar1_keep(const int64_t* prog, long long r0) {
    const long long p0 = (long long)prog[r0], p1 = (long long)prog[r0 + 1];
    return (p1 - p0) == 1 && !(p0 <= 2 || p1 <= 2);
}

// ---- losses of _optimize_kin + d loss / d pred_action.  partials[block][8] = sums over the block's rows of
//   [0] ||pred - gt||   [1] kl_multi row   [2] ||masked AR(1) error||   [3] prior_mu^2   [4] vae_mu^2   [5] prior_logvar^2   [6] vae_logvar^2
__global__ void __launch_bounds__(256) vae_kin_loss_kernel(const pulse_vae_kin_args a) {
    __shared__ float red[8][8];
    const int slot = threadIdx.x / kVL, l = threadIdx.x % kVL;
    const int E = a.embedding_size, A = a.num_actions, T = a.horizon;
    const float inv_rows = 1.0f / (float)a.rows;
    float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long long base = (long long)blockIdx.x * 8; base < a.rows; base += (long long)gridDim.x * 8) {
        const long long i = base + slot;
        if (i >= a.rows) continue;
        // action loss: torch.norm(pred - gt, dim=-1).mean()
        const float* pr = a.pred + i * a.pred_stride;
        const float* gt = a.gt + i * a.gt_stride;
        float sq = 0.f;
        for (int j = l; j < A; j += kVL) { const float d = pr[j] - gt[j]; sq += d * d; }
        const float nrm = sqrtf(vsum(sq));
        float* dmu = a.dmu + i * a.dmu_stride;
        for (int j = l; j < A; j += kVL) dmu[j] = nrm > 0.f ? ((pr[j] - gt[j]) / nrm) * inv_rows : 0.f;
        // KL(q || learned prior), loss_functions.py:3-10
        float kl = 0.f, e_pm = 0.f, e_qm = 0.f, e_pv = 0.f, e_qv = 0.f, err2 = 0.f;
        const float* zh = a.zheads + i * a.zheads_stride;
        if (l < E) {
            const float* ph = a.pheads + i * a.pheads_stride;
            const float qm = zh[l], pm = ph[l];
            float qv = zh[E + l], pv = ph[E + l];
            if (a.clamp_logvar) { qv = vclamp(qv, -5.0f, a.clamp_max); pv = vclamp(pv, -5.0f, a.clamp_max); }
            const float ep = expf(pv);
            const float dm = qm - pm;
            kl = 0.5f * (pv - qv + expf(qv) / ep + (dm * dm) / ep - 1.0f);
            e_pm = pm * pm; e_qm = qm * qm; e_pv = pv * pv; e_qv = qv * qv;
            // AR(1) error row (i -> i + 1) of this sequence, amp_agent.py:792-808
            if (a.use_ar1 && (i % T) < T - 1 && ar1_keep(a.progress, i)) {
                const float er = zh[a.zheads_stride + l] - qm * 0.99f;
                err2 = er * er;
            }
        }
        kl = vsum(kl);
        const float ar = a.use_ar1 ? sqrtf(vsum(err2)) : 0.f;
        if (a.use_regu) { e_pm = vsum(e_pm); e_qm = vsum(e_qm); e_pv = vsum(e_pv); e_qv = vsum(e_qv); }
        acc[0] += nrm; acc[1] += kl; acc[2] += ar; acc[3] += e_pm; acc[4] += e_qm; acc[5] += e_pv; acc[6] += e_qv;
    }
    if (l == 0) {
#pragma unroll
        for (int k = 0; k < 7; ++k) red[k][slot] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += red[threadIdx.x][k];
        a.partials[(long long)blockIdx.x * 8 + threadIdx.x] = s;
    }
    if (threadIdx.x == 7) a.partials[(long long)blockIdx.x * 8 + 7] = 0.f;
}

// ---- head backward: d loss / d (encoder heads) and d loss / d (prior heads) from the decoder's dz (re-parameterisation), the KL term,
// the AR(1) term and the regulariser.  Coefficients arrive with the batch means folded in.
__global__ void __launch_bounds__(256) vae_head_backward_kernel(const pulse_vae_head_bwd_args a) {
    const int slot = threadIdx.x / kVL, l = threadIdx.x % kVL;
    const int E = a.embedding_size, T = a.horizon;
    for (long long base = (long long)blockIdx.x * 8; base < a.rows; base += (long long)gridDim.x * 8) {
        const long long i = base + slot;
        if (i >= a.rows) continue;
        const float* zh = a.zheads + i * a.zheads_stride;
        const bool on = l < E;
        float qm = 0.f, qv_raw = 0.f, qv = 0.f;
        bool in_q = true;
        if (on) {
            qm = zh[l]; qv_raw = zh[E + l]; qv = qv_raw;
            if (a.clamp_logvar) { qv = vclamp(qv_raw, -5.0f, a.clamp_max); in_q = qv_raw >= -5.0f && qv_raw <= a.clamp_max; }
        }
        float g_qm = 0.f, g_qv = 0.f;
        if (on && a.dz) {                                           // z = mu + exp(0.5 logvar) eps
            const float dz = a.dz[i * a.dz_stride + l];
            g_qm = dz;
            if (a.eps) g_qv = dz * (0.5f * expf(0.5f * qv) * a.eps[i * a.eps_stride + l]);
        }
        if (a.pheads) {
            const float* ph = a.pheads + i * a.pheads_stride;
            float g_pm = 0.f, g_pv = 0.f;
            if (on) {
                const float pm = ph[l], pv_raw = ph[E + l];
                float pv = pv_raw;
                bool in_p = true;
                if (a.clamp_logvar) { pv = vclamp(pv_raw, -5.0f, a.clamp_max); in_p = pv_raw >= -5.0f && pv_raw <= a.clamp_max; }
                const float ep = expf(pv), dm = qm - pm, rq = expf(qv) / ep;
                g_qm += a.c_kl * (dm / ep);
                g_qv += a.c_kl * (0.5f * (rq - 1.0f));
                g_pm = -a.c_kl * (dm / ep) + a.c_regu * 2.0f * pm;
                g_pv = in_p ? a.c_kl * (0.5f * (1.0f - rq - (dm * dm) / ep)) + a.c_regu * 2.0f * pv : 0.f;
                g_qm += a.c_regu * 2.0f * qm;
                g_qv += a.c_regu * 2.0f * qv;
            }
            if (on && a.dpheads) {
                float* o = a.dpheads + i * a.dpheads_stride;
                o[l] = g_pm; o[E + l] = g_pv;
            }
        }
        if (a.c_ar1 != 0.f) {
            // error rows touching this step: (i - 1 -> i) contributes + err / ||err||, (i -> i + 1) contributes - 0.99 err / ||err||
            const int ti = (int)(i % T);
            float e_prev = 0.f, e_next = 0.f;
            const bool has_prev = ti > 0 && ar1_keep(a.progress, i - 1), has_next = ti < T - 1 && ar1_keep(a.progress, i);
            if (on && has_prev) e_prev = qm - zh[l - a.zheads_stride] * 0.99f;
            if (on && has_next) e_next = zh[a.zheads_stride + l] - qm * 0.99f;
            const float n_prev = sqrtf(vsum(e_prev * e_prev)), n_next = sqrtf(vsum(e_next * e_next));
            if (on) {
                if (n_prev > 0.f) g_qm += a.c_ar1 * (e_prev / n_prev);
                if (n_next > 0.f) g_qm += a.c_ar1 * (-0.99f * (e_next / n_next));
            }
        }
        if (on) {
            float* o = a.dzheads + i * a.dzheads_stride;
            o[l] = g_qm;
            o[E + l] = in_q ? g_qv : 0.f;
        }
    }
}

}  // namespace pulse

using namespace pulse;

extern "C" {

int pulse_sizeof_vae_embed_args(void) { return (int)sizeof(pulse_vae_embed_args); }
int pulse_sizeof_vae_kin_args(void) { return (int)sizeof(pulse_vae_kin_args); }
int pulse_sizeof_vae_head_bwd_args(void) { return (int)sizeof(pulse_vae_head_bwd_args); }

int pulse_vae_embed(const pulse_vae_embed_args* args, pulse_stream_t s) {
    PULSE_REQUIRE(args != nullptr, "pulse_vae_embed: null args");
    const pulse_vae_embed_args& a = *args;
    if (a.rows == 0) return PULSE_OK;
    PULSE_REQUIRE(a.rows > 0 && a.embedding_size >= 1 && a.self_obs_size >= 0, "pulse_vae_embed: bad sizes");
    PULSE_REQUIRE(a.heads && a.x && a.ain, "pulse_vae_embed: null pointer");
    PULSE_REQUIRE(a.heads_stride >= 2 * a.embedding_size && a.x_stride >= a.self_obs_size && a.ain_stride >= a.z_col + a.embedding_size &&
                  a.z_col >= a.self_obs_size, "pulse_vae_embed: strides / column offsets do not cover the rows");
    PULSE_REQUIRE(!a.eps || a.eps_stride >= a.embedding_size, "pulse_vae_embed: eps_stride too small");
    PULSE_REQUIRE(!a.cin || a.cin_stride >= a.self_obs_size, "pulse_vae_embed: cin_stride too small");
    long long blocks = ((long long)a.rows * (a.self_obs_size + a.embedding_size) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(vae_embed_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(s), a);
    return check_launch("pulse_vae_embed");
}

int pulse_vae_kin_loss(const pulse_vae_kin_args* args, pulse_stream_t s) {
    PULSE_REQUIRE(args != nullptr, "pulse_vae_kin_loss: null args");
    const pulse_vae_kin_args& a = *args;
    PULSE_REQUIRE(a.rows >= 1 && a.num_actions >= 1 && a.embedding_size >= 1 && a.embedding_size <= kVL && a.num_blocks >= 1, "pulse_vae_kin_loss: bad sizes (embedding_size <= 32)");
    PULSE_REQUIRE(a.pred && a.gt && a.zheads && a.pheads && a.dmu && a.partials, "pulse_vae_kin_loss: null pointer");
    PULSE_REQUIRE(!a.use_ar1 || (a.progress && a.horizon >= 2 && a.rows % a.horizon == 0), "pulse_vae_kin_loss: the AR(1) term needs progress and rows = sequences x horizon");
    PULSE_REQUIRE(a.zheads_stride >= 2 * a.embedding_size && a.pheads_stride >= 2 * a.embedding_size, "pulse_vae_kin_loss: head strides too small");
    hipLaunchKernelGGL(vae_kin_loss_kernel, dim3((unsigned)a.num_blocks), dim3(256), 0, as_stream(s), a);
    return check_launch("pulse_vae_kin_loss");
}

int pulse_vae_head_backward(const pulse_vae_head_bwd_args* args, pulse_stream_t s) {
    PULSE_REQUIRE(args != nullptr, "pulse_vae_head_backward: null args");
    const pulse_vae_head_bwd_args& a = *args;
    if (a.rows == 0) return PULSE_OK;
    PULSE_REQUIRE(a.rows > 0 && a.embedding_size >= 1 && a.embedding_size <= kVL, "pulse_vae_head_backward: bad sizes (embedding_size <= 32)");
    PULSE_REQUIRE(a.zheads && a.dzheads, "pulse_vae_head_backward: null pointer");
    PULSE_REQUIRE(a.zheads_stride >= 2 * a.embedding_size && a.dzheads_stride >= 2 * a.embedding_size, "pulse_vae_head_backward: head strides too small");
    PULSE_REQUIRE(!a.pheads || (a.pheads_stride >= 2 * a.embedding_size && (!a.dpheads || a.dpheads_stride >= 2 * a.embedding_size)), "pulse_vae_head_backward: prior strides too small");
    PULSE_REQUIRE(a.pheads || (a.c_kl == 0.f && a.c_regu == 0.f), "pulse_vae_head_backward: KL / regulariser terms need the prior heads");
    PULSE_REQUIRE(a.c_ar1 == 0.f || (a.progress && a.horizon >= 2 && a.rows % a.horizon == 0), "pulse_vae_head_backward: the AR(1) term needs progress and rows = sequences x horizon");
    PULSE_REQUIRE(!a.dz || a.dz_stride >= a.embedding_size, "pulse_vae_head_backward: dz_stride too small");
    long long blocks = ((long long)a.rows + 7) / 8;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(vae_head_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(s), a);
    return check_launch("pulse_vae_head_backward");
}
}
