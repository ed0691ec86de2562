// GAE reverse scan + returns for gfx950.
//
// Replaces CommonAgent.discount_values (phc/learning/common_agent.py:493-505) and the
// `mb_returns = mb_advs + mb_values` that follows it (:346-347): T dependent PyTorch
// iterations over (N,1) tensors become one launch.
//
// One thread per environment walks the horizon backwards.  The recurrence
//     A_t = delta_t + (gamma*tau) * (1 - done_t) * A_{t+1},  delta_t = r_t + gamma * V'_t - V_t
// is latency-, not bandwidth-bound (21 B per element, 2.75 MB at 4096 x 32), so the loads of a
// CHUNK of steps are issued together before the dependent chain consumes them (HBM/L2 latency
// is paid once per chunk instead of once per step).  Strides are explicit so the same kernel
// serves the reference's (T, N, 1) layout and this framework's env-major (N, T, 1) buffers.
// Compiled with -ffp-contract=off: delta/adv round as in the eager reference.
#include "common.h"

namespace pulse {

constexpr int kGaeChunk = 8;

__global__ void __launch_bounds__(64) gae_kernel(const float* __restrict__ rewards, const float* __restrict__ values,
                                                 const float* __restrict__ next_values, const uint8_t* __restrict__ dones,
                                                 int horizon, int num_envs, int64_t st, int64_t sn, float gamma,
                                                 float gamma_tau, float* __restrict__ advs, float* __restrict__ returns) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= num_envs) return;
    const int64_t base = (int64_t)n * sn;
    float last = 0.0f;
    for (int t1 = horizon; t1 > 0; t1 -= kGaeChunk) {
        float r[kGaeChunk], v[kGaeChunk], nv[kGaeChunk], nd[kGaeChunk];
#pragma unroll
        for (int k = 0; k < kGaeChunk; ++k) {
            const int t = t1 - 1 - k;
            if (t >= 0) {
                const int64_t o = base + (int64_t)t * st;
                r[k] = rewards[o]; v[k] = values[o]; nv[k] = next_values[o];
                nd[k] = 1.0f - (float)dones[o];
            }
        }
#pragma unroll
        for (int k = 0; k < kGaeChunk; ++k) {
            const int t = t1 - 1 - k;
            if (t >= 0) {
                const int64_t o = base + (int64_t)t * st;
                const float delta = r[k] + gamma * nv[k] - v[k];
                last = delta + gamma_tau * nd[k] * last;
                advs[o] = last;
                if (returns) returns[o] = last + v[k];
            }
        }
    }
}

}  // namespace pulse

using namespace pulse;

extern "C" int pulse_gae(const float* rewards, const float* values, const float* next_values, const uint8_t* dones,
                         int32_t horizon, int32_t num_envs, int64_t stride_t, int64_t stride_n, float gamma,
                         float gamma_tau, float* advs, float* returns, pulse_stream_t s) {
    PULSE_REQUIRE(horizon >= 0 && num_envs >= 0, "pulse_gae: negative size");
    if (horizon == 0 || num_envs == 0) return PULSE_OK;
    PULSE_REQUIRE(rewards && values && next_values && dones && advs, "pulse_gae: null pointer");
    // 64-thread blocks: N=4096 -> 64 workgroups spread over 64 CUs; each wave keeps 4*kGaeChunk loads in flight
    const int block = 64;
    const unsigned grid = (unsigned)((num_envs + block - 1) / block);
    hipLaunchKernelGGL(gae_kernel, dim3(grid), dim3(block), 0, as_stream(s), rewards, values, next_values, dones, horizon,
                       num_envs, stride_t, stride_n, gamma, gamma_tau, advs, returns);
    return check_launch("pulse_gae");
}
