// Standalone timing lab for the decoder GEMM kernels (development tool, not shipped in the .so).
#include <functional>
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_lab.hip -o tools/gemm_lab
// run  : tools/gemm_lab [M]
#include <vector>
#include <cstdlib>
#include "../influentialrs_amd/csrc/decoder.hip"

void irs_prof_begin(irs_ctx *, int, hipStream_t) {}
void irs_prof_end(irs_ctx *, int, hipStream_t, double, double) {}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

static float time_it(std::function<void()> f, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

// fp32 MFMA issue-rate probe: no memory traffic, NACC independent accumulator tiles per wave.
template <int NACC>
__global__ void __launch_bounds__(256) k_mfma_peak(float *out, int iters) {
    typedef float f16v __attribute__((ext_vector_type(16)));
    f16v acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float sum = 0.f;
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) sum += acc[i][j];
    if (sum == 12345.f) out[0] = sum;
}

int main(int argc, char **argv) {
    int M = argc > 1 ? atoi(argv[1]) : 204800;
    {
        float *o; CK(hipMalloc(&o, 4));
        const int iters = 4096;
        for (int wgs : {256, 512, 1024, 2048}) {
            float ms = time_it([&] { hipLaunchKernelGGL(k_mfma_peak<4>, dim3(wgs), dim3(256), 0, 0, o, iters); }, 5);
            double tf = (double)wgs * 4 * iters * 4 * 4096.0 / (ms * 1e-3) / 1e12;
            printf("mfma f32 32x32x2 peak probe: %4d WGs x 4 waves, 4 acc: %8.1f us  %6.1f TF\n", wgs, ms * 1e3, tf);
        }
    }
    struct Shape { int N, K; bool ln, relu; const char *name; } shapes[] = {
        {384, 128, false, false, "qkv    "}, {128, 128, true, false, "out+ln "}, {256, 128, false, true, "ffn1   "}, {128, 256, true, false, "ffn2+ln"}};
    float *X, *W, *B, *R, *Y, *G;
    CK(hipMalloc(&X, (size_t)M * 256 * 4)); CK(hipMalloc(&W, 384 * 256 * 4)); CK(hipMalloc(&B, 384 * 4));
    float *Y2; CK(hipMalloc(&Y2, (size_t)M * 128 * 4));
    CK(hipMalloc(&R, (size_t)M * 384 * 4)); CK(hipMalloc(&Y, (size_t)M * 384 * 4)); CK(hipMalloc(&G, 384 * 4));
    std::vector<float> h((size_t)M * 256);
    for (auto &v : h) v = (rand() / (float)RAND_MAX) * 2 - 1;
    CK(hipMemcpy(X, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(R, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, h.data(), 384 * 256 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, h.data(), 384 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(G, h.data(), 384 * 4, hipMemcpyHostToDevice));
    for (auto &sh : shapes) {
        LinArgs a{X, W, B, sh.ln ? R : nullptr, Y, M, sh.N, sh.K, sh.relu ? 1 : 0, sh.ln ? G : nullptr, G, sh.ln ? G : nullptr, G, G, nullptr, nullptr};
        LinArgs af = a; af.Rf = R; af.Yf = Y2;
        double gf = 2.0 * M * sh.N * sh.K / 1e9;
        for (int bk : {32, 16}) {
            g_lin_bk = bk; g_ln_bk = bk;
            float ms = time_it([&] { launch_linear(nullptr, a.X, a.W, a.bias, a.R, a.Y, M, sh.N, sh.K, sh.relu, 0, a.g1, a.b1, a.c, a.g2, a.b2, nullptr, nullptr); }, 20);
            printf("%s BK=%2d : %8.1f us  %6.1f TF\n", sh.name, bk, ms * 1e3, gf / ms);
        }
    }
    {   // fused block kernel (fragment-major in/out): compare with out+ln, ffn1, ffn2+ln and the next qkv above
        BlockArgs ba{};
        ba.Yf = R, ba.W1 = W, ba.b1 = B, ba.W2 = W, ba.b2 = B, ba.g = G, ba.b = G, ba.Xf = Y, ba.M = M;
        float ms = time_it([&] { hipLaunchKernelGGL((k_block<false, false>), dim3((M + 127) / 128), dim3(256), 0, 0, ba); }, 20);
        printf("ffn fused           : %8.1f us  %6.1f TF\n", ms * 1e3, 4.0 * M * 128 * 256 / 1e9 / ms);
        ba.Win = W, ba.bin = B, ba.QKV = R, ba.qkv_nt1 = 6;
        ms = time_it([&] { hipLaunchKernelGGL((k_block<false, true>), dim3((M + 127) / 128), dim3(256), 0, 0, ba); }, 20);
        printf("ffn + next qkv      : %8.1f us  %6.1f TF\n", ms * 1e3, (4.0 * M * 128 * 256 + 6.0 * M * 128 * 128) / 1e9 / ms);
        ba.Af = X, ba.Rf = Y2, ba.Wo = W, ba.bo = B, ba.g1 = G, ba.b1n = G, ba.c = G, ba.g2 = G, ba.b2n = G;
        ms = time_it([&] { hipLaunchKernelGGL((k_block<true, true>), dim3((M + 127) / 128), dim3(256), 0, 0, ba); }, 20);
        printf("out + ffn + next qkv: %8.1f us  %6.1f TF\n", ms * 1e3, (4.0 * M * 128 * 256 + 8.0 * M * 128 * 128) / 1e9 / ms);
    }
    CK(hipDeviceSynchronize());
    return 0;
}
