// Phase timing of k_topk_direct (development tool): s_memtime stamps of the last workgroup of row 0.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DIRS_LAB -DIRS_DIRECT_TIMING -Iinclude tools/direct_lab.hip -o tools/direct_lab
#include <vector>
#include <cstdlib>
#include "../influentialrs_amd/csrc/score.hip"
void irs_prof_begin(irs_ctx *, int, hipStream_t) {}
void irs_prof_end(irs_ctx *, int, hipStream_t, double, double) {}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
int main() {
    const int n = 3415, d = 128, k = 100;
    std::vector<float> h((size_t)n * d + n + d);
    for (auto &v : h) v = (rand() / (float)RAND_MAX) * 0.2f - 0.1f;
    float *W; CK(hipMalloc(&W, h.size() * 4)); CK(hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    float *bias = W + (size_t)n * d, *x = bias + n;
    unsigned long long *gk; CK(hipMalloc(&gk, (size_t)32 * IRS_CAND_CAP * 8));
    unsigned int *arrive; CK(hipMalloc(&arrive, 256)); CK(hipMemset(arrive, 0, 256));
    float *val; int64_t *ids; int32_t *st;
    CK(hipMalloc(&val, 32 * k * 4)); CK(hipMalloc(&ids, 32 * k * 8)); CK(hipMalloc(&st, 128));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 20; ++i)
            hipLaunchKernelGGL(k_topk_direct<0>, dim3((n + DIRECT_TILE - 1) / DIRECT_TILE, 1), dim3(256), 0, 0, x, d, W, bias, n, (int64_t)0, k, gk, arrive, val, ids, st, irs_path_args{}, 1);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long t[16]; CK(hipMemcpyFromSymbol(t, HIP_SYMBOL(g_direct_t), sizeof t));
        printf("%.2f us per launch; last-workgroup stamps:", ms * 1e3 / 20);
        for (int i = 1; i < 10; ++i) printf(" %llu", t[i] - t[0]);
        printf("\n");
    }
    return 0;
}
