// Development lab for the split-bf16 fused layer kernel (not product code): k_block<true, true> (float32 MFMAs) against
// k_block_x6 on the same random tile inputs and weights -- element-wise difference of x' and qkv', and kernel times.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DIRS_LAB -DX6_STAMP | -DX6_DUMP | -DX6_NO_MFMA ...] tools/x6_lab.hip -o tools/x6_lab ; run: tools/x6_lab [tokens=131072] [zero-mask=0] [grid=0: one workgroup per 128 tokens]
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "../influentialrs_amd/csrc/decoder.hip"
void irs_prof_begin(irs_ctx *, int, hipStream_t) {}
void irs_prof_end(irs_ctx *, int, hipStream_t, double, double) {}
#ifndef X6_LAB_NPL
#define X6_LAB_NPL 3 // planes per float32 operand: 3 = bf16 (x6), 2 = float16 (h3)
#endif
constexpr int LNPL = X6_LAB_NPL;
constexpr size_t LAB_LAYER_B = x6_layer_bytes(4, LNPL);
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

static float *dev_rand(size_t n, float scale, unsigned long long &z, float offset = 0.f) {
    std::vector<float> h(n);
    for (auto &v : h) {
        z ^= z << 13; z ^= z >> 7; z ^= z << 17;
        v = ((float)((z >> 40) & 0xFFFFFF) * (1.0f / 16777216.0f) * 2 - 1) * scale + offset;
    }
    float *d;
    CK(hipMalloc(&d, n * 4));
    CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    return d;
}

int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int M = argc > 1 ? atoi(argv[1]) : 131072;
    const int zero = argc > 2 ? atoi(argv[2]) : 0; // bit 0: W_o = 0, bit 1: W1 = W2 = 0, bit 2: W_in = 0 (bisecting a wrong result)
    const int full_grid = (M + 32 * X6_NW - 1) / (32 * X6_NW);
    const int grid = argc > 3 && atoi(argv[3]) > 0 ? std::min(atoi(argv[3]), full_grid) : full_grid; // (< full: persistent workgroups, float16 planes only)
    unsigned long long z = 88172645463325252ull;
    const int D = 128, F = 256;
    float *Af = dev_rand((size_t)M * D, 1.0f, z), *Rf = dev_rand((size_t)M * D, 1.0f, z);
    float *Wo = dev_rand(D * D, (zero & 1) ? 0.f : 0.09f, z), *W1 = dev_rand(F * D, (zero & 2) ? 0.f : 0.09f, z);
    float *W2 = dev_rand(D * F, (zero & 2) ? 0.f : 0.06f, z), *Win = dev_rand(3 * D * D, (zero & 4) ? 0.f : 0.1f, z);
    float *bo = dev_rand(D, 0.1f, z), *b1 = dev_rand(F, 0.1f, z), *b2 = dev_rand(D, 0.1f, z), *bin = dev_rand(3 * D, 0.1f, z);
    float *g1 = dev_rand(D, 0.05f, z, 1.f), *b1n = dev_rand(D, 0.05f, z), *g2 = dev_rand(D, 0.05f, z, 1.f), *b2n = dev_rand(D, 0.05f, z);
    float *g3 = dev_rand(D, 0.05f, z, 1.f), *b3 = dev_rand(D, 0.05f, z), *c = dev_rand(D, 0.1f, z);
    float *Xf[2], *QKV[2];
    for (int i = 0; i < 2; ++i) {
        CK(hipMalloc(&Xf[i], (size_t)M * D * 4));
        CK(hipMalloc(&QKV[i], (size_t)M * 3 * D * 4));
        CK(hipMemset(Xf[i], 0, (size_t)M * D * 4));
        CK(hipMemset(QKV[i], 0, (size_t)M * 3 * D * 4));
    }
    uint4 *Wx;
    CK(hipMalloc(&Wx, LAB_LAYER_B));
    hipLaunchKernelGGL((k_pack_x6<4, LNPL>), dim3(X6_NSTEP * 8 * LNPL * 64 / 256), dim3(256), 0, 0, Wo, W1, W2, Win, Wx);
    CK(hipDeviceSynchronize());
    BlockArgs ba{};
    ba.Af = Af, ba.Rf = Rf, ba.Wo = Wo, ba.bo = bo, ba.g1 = g1, ba.b1n = b1n, ba.c = c, ba.g2 = g2, ba.b2n = b2n;
    ba.W1 = W1, ba.b1 = b1, ba.W2 = W2, ba.b2 = b2, ba.g = g3, ba.b = b3, ba.Xf = Xf[0], ba.Y = nullptr, ba.M = M, ba.m_dev = nullptr;
    ba.Win = Win, ba.bin = bin, ba.QKV = QKV[0], ba.qkv_n0 = 0, ba.qkv_nt1 = 6;
    BlockX6Args xa{};
    xa.Af = Af, xa.Rf = Rf, xa.Wx = Wx, xa.bo = bo, xa.g1 = g1, xa.b1n = b1n, xa.c = c, xa.g2 = g2, xa.b2n = b2n, xa.b1 = b1, xa.b2 = b2;
    xa.g = g3, xa.b = b3, xa.bin = bin, xa.Xf = Xf[1], xa.QKV = QKV[1], xa.M = M;
#ifdef X6_STAMP
    const size_t nwave = (size_t)((M + 32 * X6_NW - 1) / (32 * X6_NW)) * X6_NW;
    CK(hipMalloc(&xa.stamps, nwave * 64));
    CK(hipMemset(xa.stamps, 0, nwave * 64));
#endif
#ifdef X6_DUMP
    CK(hipMalloc(&xa.dbg, LAB_LAYER_B));
    CK(hipMemset(xa.dbg, 0xEE, LAB_LAYER_B));
#endif
    constexpr int x6_lds = x6_lds_bytes(4, LNPL);
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_block_x6<0, X6_NW, false, 4, LNPL>), hipFuncAttributeMaxDynamicSharedMemorySize, x6_lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        float ms;
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_block<true, true>), dim3((M + 127) / 128), dim3(256), 0, 0, ba);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("k_block<f32>  %8.1f us\n", ms * 1e3);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_block_x6<0, X6_NW, false, 4, LNPL>), dim3(grid), dim3(64 * X6_NW), x6_lds, 0, xa);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("k_block_x6    %8.1f us\n", ms * 1e3);
    }
    auto cmp = [&](const char *name, float *a_, float *b_, size_t n) {
        std::vector<float> ha(n), hb(n);
        CK(hipMemcpy(ha.data(), a_, n * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hb.data(), b_, n * 4, hipMemcpyDeviceToHost));
        double mx = 0, sum = 0;
        size_t nan = 0, first = n;
        for (size_t i = 0; i < n; ++i) {
            if (hb[i] != hb[i]) { if (first == n) first = i; ++nan; continue; }
            const double d = fabs((double)ha[i] - hb[i]);
            mx = d > mx ? d : mx;
            sum += d;
        }
        printf("%s: max |f32 - x6| = %.3e, mean %.3e, NaN %zu of %zu (first at %zu), f32[0..3] = %g %g %g %g, x6[0..3] = %g %g %g %g\n", name, mx,
               sum / n, nan, n, first, ha[0], ha[1], ha[2], ha[3], hb[0], hb[1], hb[2], hb[3]);
    };
#ifdef X6_DUMP
    {
        std::vector<unsigned> hw(LAB_LAYER_B / 4), hd(LAB_LAYER_B / 4);
        CK(hipMemcpy(hw.data(), Wx, LAB_LAYER_B, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hd.data(), xa.dbg, LAB_LAYER_B, hipMemcpyDeviceToHost));
        for (int st = 0; st < X6_NSTEP; ++st) {
            printf("step %2d:", st);
            for (int f = 0; f < 8 * LNPL; ++f) {
                int bad = 0;
                for (int i = 0; i < 256; ++i) bad += hw[(st * 8 * LNPL + f) * 256 + i] != hd[(st * 8 * LNPL + f) * 256 + i];
                printf(" %s", bad == 0 ? "." : bad == 256 ? "X" : "x");
            }
            printf("\n");
        }
    }
#endif
#ifdef X6_STAMP
    {
        std::vector<unsigned long long> hs(nwave * 8);
        CK(hipMemcpy(hs.data(), xa.stamps, nwave * 64, hipMemcpyDeviceToHost));
        double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (size_t w = 0; w < nwave; ++w)
            for (int k = 0; k < 8; ++k) t[k] += (double)hs[w * 8 + k] / nwave;
        printf("stamps (s_memtime ticks per wave, mean over %zu waves): kernel %.0f = prologue %.0f + layer norms and splits %.0f + q|k|v stores %.0f + "
               "steps %.0f (%.1f per step; of which DMA wait %.0f, barrier %.0f, DMA issue %.0f) + rest %.0f\n", nwave, t[0], t[5], t[6], t[7], t[4],
               t[4] / 32, t[1], t[2], t[3], t[0] - t[5] - t[6] - t[7] - t[4]);
    }
#endif
    cmp("x' (fragment-major)", Xf[0], Xf[1], (size_t)M * D);
    cmp("qkv'", QKV[0], QKV[1], (size_t)M * 3 * D);
    return 0;
}
