// Store-pattern microbenchmark (development tool): how fast can 256 CUs write a row-major fp32 [M][N] matrix
// when every wave owns a 64x64 tile and uses (0) dword stores, one 128-byte row segment per half-wave (the MFMA
// C layout with columns on lanes), (1) float4 stores, 32 B per row per instruction (the transposed C layout),
// (2) float4 stores that cover whole 256-byte tile rows (what an LDS-staged epilogue would issue).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/store_lab.hip -o tools/store_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k_store(float *Y, int M, int N, float v) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int m0 = blockIdx.y * 128 + (wave >> 1) * 64, n0 = blockIdx.x * 128 + (wave & 1) * 64;
    if (MODE == 0) {
        for (int tn = 0; tn < 2; ++tn)
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk, n = n0 + tn * 32 + li;
                    Y[(int64_t)m * N + n] = v + r;
                }
    } else if (MODE == 1) {
        for (int tn = 0; tn < 2; ++tn)
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int m = m0 + tm * 32 + li, n = n0 + tn * 32 + 8 * g + 4 * lk;
                    *reinterpret_cast<float4 *>(Y + (int64_t)m * N + n) = make_float4(v, v + 1, v + 2, v + g);
                }
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = m0 + i * 4 + (lane >> 4), n = n0 + (lane & 15) * 4;
            *reinterpret_cast<float4 *>(Y + (int64_t)m * N + n) = make_float4(v, v + 1, v + 2, v + i);
        }
    }
}

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 204800, N = 384;
    float *Y; CK(hipMalloc(&Y, (size_t)M * N * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    dim3 grid(N / 128, M / 128);
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 10; ++rep) {
            CK(hipEventRecord(a, 0));
            if (mode == 0) hipLaunchKernelGGL(k_store<0>, grid, dim3(256), 0, 0, Y, M, N, 1.f);
            if (mode == 1) hipLaunchKernelGGL(k_store<1>, grid, dim3(256), 0, 0, Y, M, N, 1.f);
            if (mode == 2) hipLaunchKernelGGL(k_store<2>, grid, dim3(256), 0, 0, Y, M, N, 1.f);
            CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
        }
        printf("store pattern %d: %8.1f us  %6.2f TB/s\n", mode, best * 1e3, (double)M * N * 4 / (best * 1e-3) / 1e12);
    }
    return 0;
}
