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

int main(int argc, char **argv) {
    int M = argc > 1 ? atoi(argv[1]) : 204800;
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
    CK(hipDeviceSynchronize());
    return 0;
}
