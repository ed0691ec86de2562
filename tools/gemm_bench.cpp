// Stand-alone micro-benchmark of pulse_gemm_f32 through the C ABI (dev tool, not the official bench).
//   hipcc -O2 --offload-arch=gfx950 -Iinclude tools/gemm_bench.cpp -Lpulse_amd/csrc -lpulse_hip -Wl,-rpath,'$ORIGIN/../pulse_amd/csrc' -o tools/gemm_bench
// No torch: a gpurun call of this binary costs seconds of box time.  Shapes are the cfg2 launch mix of
// pulse_amd/learning/network.py (update M = 16384, rollout M = 4096) plus a K sweep that separates the per-tile
// overhead from the main-loop rate.  Every case is spot-checked against a naive fp32 dot-product kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>
#include "pulse_hip.h"

extern "C" int pulse_gemm_set_option(int key, int value);
extern "C" int pulse_gemm_set_debug_buffer(long long* device_buffer);

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void fill_kernel(float* p, long long n, unsigned seed, float scale, int relu = 0) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (long long)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
        const float v = ((h & 0xffffff) / 8388608.0f - 1.0f) * scale;
        p[i] = relu && v < 0.f ? 0.f : v;              // --relu: activations / masked gradients as the networks see them (half zeros)
    }
}

struct CaseP {
    int M, N, K, lda, ldb, ldc;
    int a_layout, b_layout, batch;
    long long sa, sb, sc;
    int splitk; long long split_stride;
    int act, epi; bool bias; int ldaux; long long saux;
    double algo_flops;
    int bf16;
};
struct Case : CaseP { std::string name; };

// naive check: out-of-place recompute of sampled elements of C for batch z / split 0 (splitk==1 only)
__global__ void check_kernel(const float* A, const float* B, const float* C, const float* bias, const float* aux, CaseP* c_, int nsamp,
                             float* maxerr) {
    const CaseP& c = *c_;
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsamp) return;
    unsigned h = s * 747796405u + 2891336453u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    int z = h % c.batch; h = h * 1664525u + 1013904223u;
    int m = (h >> 4) % c.M; h = h * 1664525u + 1013904223u;
    int n = (h >> 4) % c.N;
    const float* a = A + z * c.sa; const float* b = B + z * c.sb;
    double acc = 0.0;
    for (int k = 0; k < c.K; ++k) {
        float av = c.a_layout == PULSE_GEMM_RED_CONTIG ? a[(long long)m * c.lda + k] : a[(long long)k * c.lda + m];
        float bv = c.b_layout == PULSE_GEMM_RED_CONTIG ? b[(long long)n * c.ldb + k] : b[(long long)k * c.ldb + n];
        if (c.bf16 == 1) { av = (float)(__bf16)av; bv = (float)(__bf16)bv; }
        acc += (double)av * (double)bv;
    }
    float ref = (float)acc;
    if (c.splitk == 1) {
        if (c.epi == 0) { if (c.bias) ref += bias[z * c.N + n]; if (c.act == 1) ref = fmaxf(ref, 0.f); }
        else if (c.epi == 1) ref = aux[z * c.saux + (long long)m * c.ldaux + n] > 0.f ? ref : 0.f;
    }
    float got = 0.f;
    for (int sp = 0; sp < c.splitk; ++sp) got += C[z * c.sc + sp * c.split_stride + (long long)m * c.ldc + n];
    float err = fabsf(got - ref) / (1e-3f + fabsf(ref));
    atomicMax(reinterpret_cast<int*>(maxerr), __float_as_int(err));
}

int main(int argc, char** argv) {
    int iters = 20, warm = 3;
    std::string only;
    std::vector<std::pair<int, int>> opts;
    bool sweep = false, clocks = false, bf16_cases = false, x3_cases = false; int relu_data = 0;
    std::vector<std::vector<int>> custom;      // --fwd M N K lda ldb ldc
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--iters")) iters = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--only")) only = argv[++i];
        else if (!strcmp(argv[i], "--warm")) warm = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--opt")) { int k = atoi(argv[++i]); int v = atoi(argv[++i]); opts.push_back({k, v}); }
        else if (!strcmp(argv[i], "--sweep")) sweep = true;
        else if (!strcmp(argv[i], "--clocks")) clocks = true;
        else if (!strcmp(argv[i], "--bf16")) bf16_cases = true;
        else if (!strcmp(argv[i], "--x3")) x3_cases = true;
        else if (!strcmp(argv[i], "--relu")) relu_data = 1;
        else if (!strcmp(argv[i], "--fwd")) { std::vector<int> v; for (int j = 0; j < 6; ++j) v.push_back(atoi(argv[++i])); custom.push_back(v); }
    }
    for (auto& o : opts) pulse_gemm_set_option(o.first, o.second);
    printf("# options:"); for (auto& o : opts) printf(" %d=%d", o.first, o.second); printf("\n");

    const long long big = 16384LL * 4096 + 4096;
    float *X, *W, *H, *AUX, *BIAS, *maxerr; CaseP* dcase;
    CK(hipMalloc(&X, big * 4)); CK(hipMalloc(&W, big * 4)); CK(hipMalloc(&H, big * 4)); CK(hipMalloc(&AUX, big * 4));
    CK(hipMalloc(&BIAS, 8192 * 4)); CK(hipMalloc(&maxerr, 4)); CK(hipMalloc(&dcase, sizeof(CaseP)));
    fill_kernel<<<1024, 256>>>(X, big, 1u, 1.0f, relu_data);
    fill_kernel<<<1024, 256>>>(W, big, 2u, 0.05f);
    fill_kernel<<<1024, 256>>>(AUX, big, 3u, 1.0f, relu_data);
    fill_kernel<<<8, 256>>>(BIAS, 8192, 4u, 0.5f);
    CK(hipDeviceSynchronize());

    const int R = PULSE_GEMM_RED_CONTIG, O = PULSE_GEMM_OUT_CONTIG;
    std::vector<Case> cases;
    auto add = [&](const char* nm, int M, int N, int K, int lda, int ldb, int ldc, int al, int bl, int batch, long long sa, long long sb,
                   long long sc, int splitk, long long ss, int act, int epi, bool bias, int ldaux, long long saux, double algoK, double algoN) {
        Case c;
        static_cast<CaseP&>(c) = CaseP{M, N, K, lda, ldb, ldc, al, bl, batch, sa, sb, sc, splitk, ss, act, epi, bias, ldaux, saux,
                                       2.0 * M * (algoN > 0 ? algoN : N) * (algoK > 0 ? algoK : K) * batch, 0};
        c.name = nm;
        cases.push_back(c);
    };
    for (int M : {16384, 4096}) {
        const char* p = M == 16384 ? "upd " : "roll";
        char nm[96];
        snprintf(nm, 96, "%s fwdL1 %dx2048x960", p, M);
        add(nm, M, 2048, 960, 960, 960, 2048, R, R, 1, 0, 0, 0, 1, 0, 1, 0, true, 0, 0, 934, 0);
        snprintf(nm, 96, "%s fwdL2 2x(%dx512x1024)", p, M);
        add(nm, M, 512, 1024, 2048, 1024, 1024, R, R, 2, 1024, 512 * 1024, 512, 1, 0, 1, 0, true, 0, 0, 0, 0);
        snprintf(nm, 96, "%s heads 2x(%dx69x512)", p, M);
        add(nm, M, 69, 512, 1024, 512, 144, R, R, 2, 512, 69 * 512, 72, 1, 0, 0, 0, true, 0, 0, 0, 35);
    }
    {
        const int M = 16384;
        add("upd  dXhd  2x(16384x512x69)", M, 512, 69, 144, 512, 1024, R, O, 2, 72, 69 * 512, 512, 1, 0, 0, 1, false, 1024, 512, 35, 0);
        add("upd  dXL2  2x(16384x1024x512)", M, 1024, 512, 1024, 1024, 2048, R, O, 2, 512, 512 * 1024, 1024, 1, 0, 0, 1, false, 2048, 1024, 0, 0);
        add("upd  dWhd  2x(69x512x16384) s32", 69, 512, M, 144, 1024, 512, O, O, 2, 72, 512, 69 * 512, 32, 2 * 69 * 512 + 256, 0, 0, false, 0, 0, M * 35.0 / 69, 0);
        add("upd  dWL2  2x(512x1024x16384) s8", 512, 1024, M, 1024, 2048, 1024, O, O, 2, 512, 1024, 512 * 1024, 8, 2 * 512 * 1024, 0, 0, false, 0, 0, 0, 0);
        add("upd  dWL1  2048x960x16384 s8", 2048, 960, M, 2048, 960, 960, O, O, 1, 0, 0, 0, 8, 2048 * 960, 0, 0, false, 0, 0, 0, 934);
        add("alt  dWL1  2048x960x16384 s4", 2048, 960, M, 2048, 960, 960, O, O, 1, 0, 0, 0, 4, 2048 * 960, 0, 0, false, 0, 0, 0, 934);
        add("alt  dWL1  2048x960x16384 s16", 2048, 960, M, 2048, 960, 960, O, O, 1, 0, 0, 0, 16, 2048 * 960, 0, 0, false, 0, 0, 0, 934);
        add("alt  dWL2  2x(512x1024x16384) s4", 512, 1024, M, 1024, 2048, 1024, O, O, 2, 512, 1024, 512 * 1024, 4, 2 * 512 * 1024, 0, 0, false, 0, 0, 0, 0);
        add("alt  dWL2  2x(512x1024x16384) s16", 512, 1024, M, 1024, 2048, 1024, O, O, 2, 512, 1024, 512 * 1024, 16, 2 * 512 * 1024, 0, 0, false, 0, 0, 0, 0);
        add("crit fwdL1 16384x1024x960", M, 1024, 960, 960, 960, 2048, R, R, 1, 0, 0, 0, 1, 0, 1, 0, true, 0, 0, 934, 0);
        add("crit fwdL2 16384x512x1024", M, 512, 1024, 2048, 1024, 1024, R, R, 1, 0, 0, 0, 1, 0, 1, 0, true, 0, 0, 0, 0);
        add("big  4096^3", 4096, 4096, 4096, 4096, 4096, 4096, R, R, 1, 0, 0, 0, 1, 0, 0, 0, false, 0, 0, 0, 0);
    }
    for (auto& v : custom) {
        char nm[96];
        snprintf(nm, 96, "cust fwd %dx%dx%d ld %d/%d/%d", v[0], v[1], v[2], v[3], v[4], v[5]);
        add(nm, v[0], v[1], v[2], v[3], v[4], v[5], R, R, 1, 0, 0, 0, 1, 0, 1, 0, true, 0, 0, 0, 0);
    }
    if (sweep)
        for (int K : {32, 64, 128, 256, 512, 1024, 2048, 4096}) {
            char nm[96];
            snprintf(nm, 96, "swp  fwd 16384x2048xK%d", K);
            add(nm, 16384, 2048, K, 4096, 4096, 2048, R, R, 1, 0, 0, 0, 1, 0, 1, 0, true, 0, 0, 0, 0);
        }

    if (x3_cases) {
        const size_t n0 = cases.size();
        for (size_t i = 0; i < n0; ++i) { Case c = cases[i]; c.bf16 = 2; c.name = "x3   " + c.name; cases.push_back(c); }
    }
    if (bf16_cases) {
        const size_t n0 = cases.size();
        for (size_t i = 0; i < n0; ++i) { Case c = cases[i]; if (c.bf16) continue; c.bf16 = 1; c.name = "bf16 " + c.name; cases.push_back(c); }
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    {   // DVFS ramp: the chip needs sustained load before its clock settles; measured cases right after idle run ~10 % slow
        pulse_gemm_desc d; memset(&d, 0, sizeof d);
        d.A = X; d.B = W; d.C = H; d.M = 4096; d.N = 4096; d.K = 4096; d.lda = d.ldb = d.ldc = 4096; d.batch = 1; d.split_k = 1;
        for (int i = 0; i < 300; ++i) pulse_gemm_f32(&d, nullptr);
        CK(hipDeviceSynchronize());
    }
    double tot_flops = 0, tot_time = 0;
    for (auto& c : cases) {
        if (!only.empty() && c.name.find(only) == std::string::npos) continue;
        pulse_gemm_desc d; memset(&d, 0, sizeof d);
        d.A = X; d.B = W; d.C = H; d.bias = c.bias ? BIAS : nullptr; d.aux = c.epi ? AUX : nullptr;
        if (c.a_layout == O) { d.A = AUX; d.B = X; d.C = H; }       // dW: dY = AUX-like, X = activations
        d.M = c.M; d.N = c.N; d.K = c.K; d.lda = c.lda; d.ldb = c.ldb; d.ldc = c.ldc; d.ldaux = c.ldaux;
        d.a_layout = c.a_layout; d.b_layout = c.b_layout; d.batch = c.batch;
        d.stride_a = c.sa; d.stride_b = c.sb; d.stride_c = c.sc; d.stride_aux = c.saux; d.stride_bias = c.N;
        d.split_k = c.splitk; d.split_stride = c.split_stride; d.activation = c.act; d.epilogue = c.epi;
        d.compute_type = c.bf16 == 2 ? PULSE_GEMM_COMPUTE_F32X3 : c.bf16 ? PULSE_GEMM_COMPUTE_BF16 : PULSE_GEMM_COMPUTE_F32;
        for (int i = 0; i < warm; ++i)
            if (pulse_gemm_f32(&d, nullptr) != PULSE_OK) { fprintf(stderr, "%s: %s\n", c.name.c_str(), pulse_last_error()); exit(1); }
        CK(hipDeviceSynchronize());
        const int nwg = ((c.M + 127) / 128) * ((c.N + 127) / 128) * c.batch * c.splitk;
        long long* dbg = nullptr;
        if (clocks) { CK(hipMalloc(&dbg, (size_t)nwg * 64)); CK(hipMemset(dbg, 0, (size_t)nwg * 64)); pulse_gemm_set_debug_buffer(dbg); CK(hipDeviceSynchronize()); }
        CK(hipEventRecord(e0));
        for (int i = 0; i < iters; ++i) pulse_gemm_f32(&d, nullptr);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double t = ms * 1e-3 / iters;
        CK(hipMemset(maxerr, 0, 4));
        CK(hipMemcpy(dcase, static_cast<const CaseP*>(&c), sizeof(CaseP), hipMemcpyHostToDevice));
        check_kernel<<<8, 256>>>(d.A, d.B, d.C, d.bias, d.aux, dcase, 2048, maxerr);
        float err; CK(hipMemcpy(&err, maxerr, 4, hipMemcpyDeviceToHost));
        if (clocks) {
            std::vector<long long> h((size_t)nwg * 8);
            CK(hipMemcpy(h.data(), dbg, (size_t)nwg * 64, hipMemcpyDeviceToHost));
            CK(hipFree(dbg));
            pulse_gemm_set_debug_buffer(nullptr);
            double sum_main_c = 0, sum_main_w = 0, sum_epi_c = 0, sum_all_c = 0, sum_all_w = 0, sum_epi1 = 0; long long wmin = 1LL << 62, wmax = 0;
            for (int i = 0; i < nwg; ++i) {
                const long long* o = &h[(size_t)i * 8];
                sum_main_c += (double)(o[2] - o[0]); sum_main_w += (double)(o[3] - o[1]); sum_epi_c += (double)(o[4] - o[2]);
                if (o[6]) sum_epi1 += (double)(o[6] - o[2]);
                sum_all_c += (double)(o[4] - o[0]); sum_all_w += (double)(o[5] - o[1]);
                if (o[1] < wmin) wmin = o[1];
                if (o[5] > wmax) wmax = o[5];
            }
            {   // per-CU census: how many workgroups each CU ran, when its last one ended
                std::vector<int> cnt(4096, 0); std::vector<long long> last(4096, 0), first(4096, 1LL << 62);
                for (int i = 0; i < nwg; ++i) {
                    const long long* o = &h[(size_t)i * 8];
                    const int key = (int)((((o[7] >> 32) & 15) << 8) | ((o[7] >> 8) & 255));
                    cnt[key]++; if (o[5] > last[key]) last[key] = o[5]; if (o[1] < first[key]) first[key] = o[1];
                }
                int ncu = 0, cmin = 1 << 30, cmax = 0; long long lmin = 1LL << 62, lmax = 0; int hist[16] = {0};
                for (int k = 0; k < 4096; ++k) if (cnt[k]) { ++ncu; if (cnt[k] < cmin) cmin = cnt[k]; if (cnt[k] > cmax) cmax = cnt[k];
                    if (last[k] < lmin) lmin = last[k]; if (last[k] > lmax) lmax = last[k]; hist[cnt[k] < 15 ? cnt[k] : 15]++; }
                printf("    census: %d CUs, WGs per CU min %d max %d, last-end spread %.1f us, hist:", ncu, cmin, cmax, (lmax - lmin) * 0.01);
                for (int k = 0; k < 16; ++k) if (hist[k]) printf(" %dx%d", k, hist[k]);
                printf("\n");
            }
            const int kc = (c.K + c.splitk - 1) / c.splitk;
            const int cyc_tile = c.bf16 == 2 ? 768 : c.bf16 ? 512 : 4096;      // MFMA-pipe cycles per wave and k-tile
            const int ktiles = c.bf16 == 2 ? (kc + 15) / 16 : c.bf16 ? (kc + 63) / 64 : (kc + 31) / 32;
            const double ghz = sum_all_c / (sum_all_w * 10.0);          // shader cycles per ns, sustained (last of the timed launches)
            const double ideal_cyc = (double)nwg * ktiles * (double)cyc_tile / 256.0;   // MFMA-pipe cycles per SIMD if all 1024 SIMDs stay busy
            printf("    clocks: %d WGs, main %.0f cyc/WG (solo ideal %d), epilogue %.0f cyc (to LDS+barrier %.0f), sustained %.3f GHz, span %.1f us, pipe-cycle efficiency %.3f\n",
                   nwg, sum_main_c / nwg, ktiles * cyc_tile, sum_epi_c / nwg, sum_epi1 / nwg, ghz, (wmax - wmin) * 0.01, ideal_cyc / (t * 1e9 * ghz));
        }
        printf("%-36s %9.1f us %7.1f TF/s  (exec %7.1f)  maxrelerr %.2e%s\n", c.name.c_str(), t * 1e6, c.algo_flops / t / 1e12,
               2.0 * c.M * c.N * c.K * c.batch / t / 1e12, err, err > (c.bf16 == 1 ? 2e-2 : 2e-4) ? "  <-- MISMATCH" : "");
        fflush(stdout);
        if (c.name.rfind("upd", 0) == 0) { tot_flops += c.algo_flops; tot_time += t; }
    }
    if (tot_time > 0) printf("update mix (one launch each, heads..dW): %.1f us  %.1f TF/s\n", tot_time * 1e6, tot_flops / tot_time / 1e12);
    return 0;
}
