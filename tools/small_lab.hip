// Phase timing of the 16-token layer kernel (development tool): s_memtime stamps of workgroup 0.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DIRS_LAB -DIRS_SMALL_TIMING tools/small_lab.hip -o tools/small_lab
#include <vector>
#include <cstdlib>
#include "../influentialrs_amd/csrc/decoder.hip"
void irs_prof_begin(irs_ctx *, int, hipStream_t) {}
void irs_prof_end(irs_ctx *, int, hipStream_t, double, double) {}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// Weight-fetch probe: every workgroup reads `kb` KB of weights (4 waves x kb/4 KB), no MFMA.
// PAT 0: the MFMA A-fragment pattern (16 rows x 64 B per wave instruction, row stride 512 B);
// PAT 1: 1 KB contiguous per wave instruction.
template <int PAT>
__global__ void __launch_bounds__(1024) k_wload(const float *W, float *out, int n_inst, int shared) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lq = lane & 15, gq = lane >> 4;
    const float *base = W + (shared ? 0 : (size_t)blockIdx.x * (blockDim.x / 64) * n_inst * 256) + (size_t)wave * n_inst * 256;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 16
    for (int i = 0; i < n_inst; ++i) {
        const float *p;
        if (PAT == 0) { // tile = 16 rows x 128 floats (8 instructions per tile)
            const int tile = i >> 3, j = i & 7;
            p = base + (size_t)tile * 2048 + lq * 128 + 16 * j + 4 * gq;
        } else
            p = base + (size_t)i * 256 + lane * 4;
        const float4 v = *reinterpret_cast<const float4 *>(p);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[0] = acc.x;
}

// Is a launch's first touch of the weights slower than a second pass inside the same kernel (cold caches per launch)?
__device__ unsigned long long g_pass_t[4];
__global__ void __launch_bounds__(256) k_wload2(const float *W, float *out, int n_inst) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *base = W + (size_t)wave * n_inst * 256;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int pass = 0; pass < 3; ++pass) {
        if (blockIdx.x == 0 && threadIdx.x == 0) g_pass_t[pass] = __builtin_amdgcn_s_memtime();
#pragma unroll 16
        for (int i = 0; i < n_inst; ++i) {
            const float4 v = *reinterpret_cast<const float4 *>(base + (size_t)i * 256 + lane * 4);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) g_pass_t[3] = __builtin_amdgcn_s_memtime();
    if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[0] = acc.x;
}
int main(int argc, char **argv) {
    int M = argc > 1 ? atoi(argv[1]) : 165;
    float *buf; CK(hipMalloc(&buf, (size_t)(M * 1024 + 400000) * 4));
    std::vector<float> h((size_t)M * 1024 + 400000);
    for (auto &v : h) v = (rand() / (float)RAND_MAX) * 0.2f - 0.1f;
    CK(hipMemcpy(buf, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    float *W = buf + (size_t)M * 1024;
    SmallBlockArgs a{};
    a.AO = buf; a.X = buf + (size_t)M * 128; a.Xo = buf + (size_t)M * 256; a.QKV = buf + (size_t)M * 384;
    a.Wo = W; a.W1 = W + 16384; a.W2 = W + 49152; a.Win = W + 81920; a.bo = W + 131072; a.b1 = a.bo + 128; a.b2 = a.b1 + 256;
    a.bin = a.b2 + 128; a.g1 = a.bin + 384; a.b1n = a.g1 + 128; a.c = a.b1n + 128; a.g2 = a.c + 128; a.b2n = a.g2 + 128; a.g3 = a.b2n + 128;
    a.b3n = a.g3 + 128; a.M = M;
    float *wf; CK(hipMalloc(&wf, (size_t)(SMALL_WF_LAYER + SMALL_WF_WIN) * 4));
    hipLaunchKernelGGL(k_pack_frag16, dim3(16), dim3(256), 0, 0, a.Wo, wf + SMALL_WF_WO, 128, 128);
    hipLaunchKernelGGL(k_pack_frag16, dim3(32), dim3(256), 0, 0, a.W1, wf + SMALL_WF_W1, 256, 128);
    hipLaunchKernelGGL(k_pack_frag16, dim3(32), dim3(256), 0, 0, a.W2, wf + SMALL_WF_W2, 128, 256);
    hipLaunchKernelGGL(k_pack_frag16, dim3(48), dim3(256), 0, 0, a.Win, wf + SMALL_WF_LAYER, 384, 128);
    a.Wf = wf; a.Wfin = wf + SMALL_WF_LAYER;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k_block_small16<true, 1>), dim3((M + 15) / 16), dim3(64 * SB_NW), 0, 0, a);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long t[16]; CK(hipMemcpyFromSymbol(t, HIP_SYMBOL(g_small_t), sizeof t));
        printf("M=%d: %.2f us per launch back to back; stamps (ticks from entry):", M, ms * 1e3 / 20);
        for (int i = 1; i <= 13; ++i) printf(" %llu", t[i] - t[0]);
        printf("\n");
    }
    for (int dd : {30, 64}) { // the generic kernel: phase stamps at the reference's default d and at config 1's
        const int F = 256, dp = small_any_dp(dd), Fp = 256, Qp = (3 * dd + 15) & ~15;
        float *wfa; CK(hipMalloc(&wfa, (size_t)2 * small_any_layer_floats(dd, F) * 4));
        auto pk = [&](const float *Wsrc, float *out, int N, int K, int Np, int Kp) {
            hipLaunchKernelGGL(k_pack_frag16_any, dim3((Np * Kp / 4 + 255) / 256), dim3(256), 0, 0, Wsrc, out, N, K, Np, Kp);
        };
        pk(a.Wo, wfa, dd, dd, dp, dp);
        pk(a.W1, wfa + dp * dp, F, dd, Fp, dp);
        pk(a.W2, wfa + dp * dp + Fp * dp, dd, F, dp, Fp);
        pk(a.Win, wfa + small_any_win_off(dd, F), 3 * dd, dd, Qp, dp);
        SmallBlockArgs b = a;
        b.Wf = wfa; b.Wfin = wfa + small_any_win_off(dd, F); b.M = 50;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 20; ++i)
                launch_small_any(true, b.M, dd, F, b, 0);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long t[16]; CK(hipMemcpyFromSymbol(t, HIP_SYMBOL(g_small_t), sizeof t));
            printf("any d=%d M=%d: %.2f us per launch; stamps:", dd, b.M, ms * 1e3 / 20);
            for (int i = 1; i <= 9; ++i) printf(" %llu", t[i] - t[0]);
            printf("\n");
        }
    }
    {
        float *big; CK(hipMalloc(&big, (size_t)64 << 20));
        CK(hipMemset(big, 0, (size_t)64 << 20));
        const int n_inst = 128; // 128 KB per wave, 512 KB per workgroup
        for (int shared = 1; shared >= 0; --shared)
            for (int wgs : {1, 11, 44}) {
                for (int pat = 0; pat < 2; ++pat) {
                    float best = 1e9;
                    for (int rep = 0; rep < 5; ++rep) {
                        CK(hipEventRecord(e0, 0));
                        if (pat == 0) hipLaunchKernelGGL(k_wload<0>, dim3(wgs), dim3(256), 0, 0, big, buf, n_inst, shared);
                        else hipLaunchKernelGGL(k_wload<1>, dim3(wgs), dim3(256), 0, 0, big, buf, n_inst, shared);
                        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                        best = ms < best ? ms : best;
                    }
                    printf("wload shared=%d wgs=%2d pat=%d: %.2f us for 512 KB per workgroup\n", shared, wgs, pat, best * 1e3);
                }
            }
    }
    {
        float *big; CK(hipMalloc(&big, (size_t)64 << 20));
        CK(hipMemset(big, 0, (size_t)64 << 20));
        for (int threads : {256, 512, 1024}) {
            const int n_inst = 128 * 256 / threads;
            float best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(k_wload<1>, dim3(11), dim3(threads), 0, 0, big, buf, n_inst, 1);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            printf("wload contiguous, 11 wgs x %d threads: %.2f us for 512 KB per workgroup\n", threads, best * 1e3);
        }
    }
    {
        float *big; CK(hipMalloc(&big, (size_t)64 << 20));
        CK(hipMemset(big, 0, (size_t)64 << 20));
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(k_wload2, dim3(11), dim3(256), 0, 0, big, buf, 128);
            CK(hipDeviceSynchronize());
            unsigned long long t[4]; CK(hipMemcpyFromSymbol(t, HIP_SYMBOL(g_pass_t), sizeof t));
            printf("512 KB per workgroup, three passes inside one launch: %llu %llu %llu ticks\n", t[1] - t[0], t[2] - t[1], t[3] - t[2]);
        }
    }
    return 0;
}
