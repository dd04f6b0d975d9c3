// Development lab for the bf16 catalog sweep (not product code): times the production sweep kernels and
// candidate re-designs on one catalog shape, with the REAL thresholds of the production pipeline, and checks that
// every emitting variant emits exactly the production kernel's candidate set size.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/sweep_lab.hip -o tools/sweep_lab
// run:   tools/sweep_lab [N=1000000] [d=128] [M=1024] [reps=10]
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "../influentialrs_amd/csrc/score.hip"
void irs_prof_begin(irs_ctx *, int, hipStream_t) {}
void irs_prof_end(irs_ctx *, int, hipStream_t, double, double) {}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

// ring kernel instantiations with explicit parameters (the product picks its own in launch_sweep_bf16)
template <int KS, int RT, int TPS, int NW, int WPS, int NSLOT, int MODE>
static void launch_ring_lab(SweepArgs a, int nt, int rounds_x10, hipStream_t s) {
    a.n_ublocks = (a.UT + NW * RT - 1) / (NW * RT);
    const size_t lds = (size_t)NSLOT * TPS * KS * 1024 + (size_t)NSLOT * TPS * 256 + (size_t)NW * (EMIT_Q * 12 + 16);
    auto kern = k_sweep_ring<KS, RT, TPS, NW, WPS, NSLOT, MODE>;
    static int slots = 0;
    if (!slots) {
        if (lds > 65536) CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        slots = resident_workgroups(kern, NW * 64, lds);
        printf("   [%d resident workgroups]\n", slots);
    }
    int strips = (int)((long long)rounds_x10 * slots / 10 / a.n_ublocks) & ~7;
    if (strips < 8) strips = 8;
    a.tile_begin = 0, a.tile_end = nt, a.tile_stride = 1;
    if (MODE == MODE_PRE) { // group structure: strips of 4 * tpw tiles
        a.tiles_per_wg = 0;
        a.tiles_per_wave = ((nt + strips - 1) / strips + 3) / 4;
        a.n_strips = (nt + 4 * a.tiles_per_wave - 1) / (4 * a.tiles_per_wave);
        if (8 * a.n_strips > IRS_MAX_GROUPS) {
            printf("   (PRE variant skipped: %d groups > %d)\n", 8 * a.n_strips, IRS_MAX_GROUPS);
            return;
        }
        a.n_groups = 8 * a.n_strips;
    } else {
        a.tiles_per_wg = (nt + strips - 1) / strips;
        a.n_strips = (nt + a.tiles_per_wg - 1) / a.tiles_per_wg;
    }
    dim3 grid(((a.n_strips + 7) / 8) * 8 * a.n_ublocks);
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, s, a);
}

// PRE group maxima of a ring instantiation against the streaming kernel's, same decomposition (bitwise)
template <int KS, int RT, int TPS, int NW, int NSLOT, int DBG = 0>
static void check_pre(irs_ctx *ctx, SweepArgs a, int nt, int stride, int tpw, hipStream_t s, const char *name) {
    a.tile_begin = 0, a.tile_end = nt, a.tile_stride = stride, a.tiles_per_wave = tpw, a.tiles_per_wg = 0;
    const int nts = (nt + stride - 1) / stride;
    a.n_strips = (nts + 4 * tpw - 1) / (4 * tpw);
    const size_t G = (size_t)8 * a.n_strips, n = G * a.M_pad;
    a.n_groups = (int)G;
    std::vector<float> ref(n), got(n);
    CK(hipMemset(ctx->gm, 0xFF, n * 4));
    SweepArgs b = a;
    ctx->sweep_variant = 3; // streaming kernel for any row count (lab only)
    launch_sweep_bf16<MODE_PRE>(ctx, b, s);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(ref.data(), ctx->gm, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemset(ctx->gm, 0xFF, n * 4));
    ctx->sweep_variant = 0;
    b = a;
    b.n_ublocks = (a.UT + NW * RT - 1) / (NW * RT);
    const size_t lds = (size_t)NSLOT * TPS * KS * 1024 + (size_t)NSLOT * TPS * 256 + (size_t)NW * (EMIT_Q * 12 + 16);
    auto kern = k_sweep_ring<KS, RT, TPS, NW, 2, NSLOT, MODE_PRE, DBG>;
    if (lds > 65536) CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(((b.n_strips + 7) / 8) * 8 * b.n_ublocks), dim3(NW * 64), lds, s, b);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(got.data(), ctx->gm, n * 4, hipMemcpyDeviceToHost));
    size_t bad = 0, first = n;
    for (size_t i = 0; i < n; ++i)
        if (memcmp(&ref[i], &got[i], 4)) {
            if (first == n) first = i;
            ++bad;
        }
    printf("  PRE check %-28s stride %d tpw %d: %zu of %zu group maxima differ", name, stride, tpw, bad, n);
    if (bad) printf(" (first: group %zu row %zu: %g vs %g)", first / a.M_pad, first % a.M_pad, got[first], ref[first]);
    printf("\n");
    if (bad > 10000) {
        std::vector<size_t> by_rt(a.UT, 0), by_g8(8, 0), by_lane(32, 0);
        size_t hi = 0, lo = 0;
        for (size_t i = 0; i < n; ++i)
            if (memcmp(&ref[i], &got[i], 4) && ref[i] == ref[i] && got[i] == got[i]) {
                by_rt[(i % a.M_pad) / 32]++;
                by_g8[(i / a.M_pad) % 8]++;
                by_lane[i % 32]++;
                (got[i] > ref[i] ? hi : lo)++;
            }
        printf("     higher %zu lower %zu; by row tile:", hi, lo);
        for (int t = 0; t < a.UT; ++t) printf(" %zu", by_rt[t]);
        printf("\n     by group mod 8 (quarter * 2 + half):");
        for (int t = 0; t < 8; ++t) printf(" %zu", by_g8[t]);
        printf("\n     by row mod 32:");
        for (int t = 0; t < 32; ++t) printf(" %zu", by_lane[t]);
        printf("\n");
    }
}

// 16x16x32 ring instantiations with explicit parameters
template <int KS, int RT16, int NW, int NSLOT, int MODE>
static void launch_ring16_lab(SweepArgs a, int nt, int rounds_x10, hipStream_t s) {
    a.n_ublocks = (a.UT + NW * (RT16 / 2) - 1) / (NW * (RT16 / 2));
    const size_t lds = ring16_lds_bytes(KS, RT16, NW, NSLOT);
    auto kern = k_sweep_ring16<KS, RT16, NW, NSLOT, MODE>;
    static int slots = 0;
    if (!slots) {
        if (lds > 65536) CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        slots = resident_workgroups(kern, NW * 64, lds);
        printf("   [%d resident workgroups]\n", slots);
    }
    int strips = (int)((long long)rounds_x10 * slots / 10 / a.n_ublocks) & ~7;
    if (strips < 8) strips = 8;
    a.tile_begin = 0, a.tile_end = nt, a.tile_stride = 1;
    if (MODE == MODE_PRE) {
        a.tiles_per_wg = 0;
        a.tiles_per_wave = ((nt + strips - 1) / strips + 3) / 4;
        a.n_strips = (nt + 4 * a.tiles_per_wave - 1) / (4 * a.tiles_per_wave);
        if (8 * a.n_strips > IRS_MAX_GROUPS) {
            printf("   (PRE variant skipped: %d groups > %d)\n", 8 * a.n_strips, IRS_MAX_GROUPS);
            return;
        }
        a.n_groups = 8 * a.n_strips;
    } else {
        a.tiles_per_wg = (nt + strips - 1) / strips;
        a.n_strips = (nt + a.tiles_per_wg - 1) / a.tiles_per_wg;
    }
    dim3 grid(((a.n_strips + 7) / 8) * 8 * a.n_ublocks);
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, s, a);
}

// in-kernel stamps of the 16x16x32 ring (DBG = 1 build): where a wave's cycles go
template <int KS, int RT16, int NW, int NSLOT>
static void stamps_ring16(irs_ctx *ctx, SweepArgs a, int nt, hipStream_t s) {
    a.n_ublocks = (a.UT + NW * (RT16 / 2) - 1) / (NW * (RT16 / 2));
    const size_t lds = ring16_lds_bytes(KS, RT16, NW, NSLOT);
    auto kern = k_sweep_ring16<KS, RT16, NW, NSLOT, MODE_EMIT, 1>;
    if (lds > 65536) CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int slots = resident_workgroups(kern, NW * 64, lds);
    a.tile_begin = 0, a.tile_end = nt;
    ring_emit_grid(a, slots);
    const unsigned grid = ((a.n_strips + 7) / 8) * 8 * a.n_ublocks;
    CK(hipMemsetAsync(ctx->cand_cnt, 0, (size_t)a.M_pad * IRS_CAND_BUCKETS * 4, s));
    CK(hipMemsetAsync(ctx->gm, 0, (size_t)grid * NW * 64, s));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, s, a);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h((size_t)grid * NW * 8);
    CK(hipMemcpy(h.data(), ctx->gm, h.size() * 8, hipMemcpyDeviceToHost));
    double tot = 0, tv = 0, tb = 0, th = 0, nh = 0, ns = 0, n = 0;
    for (size_t i = 0; i < h.size(); i += 8) {
        if (!h[i + 5]) continue;
        tot += h[i], tv += h[i + 1], tb += h[i + 2], th += h[i + 3], nh += h[i + 4], ns += h[i + 5], n += 1;
    }
    printf("  stamps ring16<KS=%d,RT16=%d>: %0.f waves, %.1f steps each; per STEP: total %.0f cycles, DMA wait %.0f, barrier %.0f, "
           "hit handling %.0f (%.2f handled sub-tiles with hits per step, %.0f cycles each incl. ~80 of stamps)\n",
           KS, RT16, n, ns / n, tot / ns, tv / ns, tb / ns, th / ns, nh / ns, nh > 0 ? th / nh : 0.0);
}

static float median(std::vector<float> v) {
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int64_t N = argc > 1 ? atoll(argv[1]) : 1000000;
    const int d = argc > 2 ? atoi(argv[2]) : 128;
    const int M = argc > 3 ? atoi(argv[3]) : 1024;
    const int reps = argc > 4 ? atoi(argv[4]) : 10;
    const int k = 100;
    irs_ctx *ctx = new irs_ctx();
    memset(ctx, 0, sizeof(*ctx));
    ctx->dims.n_item = N;
    ctx->dims.d = d;
    ctx->dims.max_k = k;
    ctx->dims.max_rows = M;
    ctx->shard = irs_shard{0, 1, 0, N};
    ctx->n_local = N;
    int dp = 16;
    while (dp < d) dp <<= 1;
    ctx->d_pad = dp;
    ctx->KS = dp / 16;
    ctx->n_tiles = (int)((N + 31) / 32);
    ctx->max_rows = M;
    ctx->m_pad_max = (M + 31) & ~31;
    const int mp = ctx->m_pad_max;
    // random catalog and rows
    float *W, *b, *x;
    CK(hipMalloc(&W, (size_t)N * d * 4));
    CK(hipMalloc(&b, (size_t)N * 4));
    CK(hipMalloc(&x, (size_t)M * d * 4));
    {
        std::vector<float> h((size_t)N * d);
        unsigned long long z = 88172645463325252ull;
        auto rnd = [&]() { z ^= z << 13; z ^= z >> 7; z ^= z << 17; return (float)((z >> 40) & 0xFFFFFF) * (1.0f / 16777216.0f); };
        const float sc = 1.0f / sqrtf((float)d);
        for (auto &v : h) v = (rnd() * 2 - 1) * sc;
        CK(hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        std::vector<float> hb(N);
        for (auto &v : hb) v = (rnd() + rnd() + rnd() + rnd() - 2.0f) * 0.17f;
        CK(hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
        std::vector<float> hx((size_t)M * d);
        for (auto &v : hx) v = (rnd() + rnd() + rnd() + rnd() - 2.0f) * 1.73f; // ~N(0,1)
        CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    }
    ctx->proj_w = W;
    ctx->proj_b = b;
    CK(hipMalloc(&ctx->wp, (size_t)ctx->n_tiles * ctx->KS * 1024));
    CK(hipMalloc(&ctx->bias_pad, (size_t)ctx->n_tiles * 32 * 4 + 1024));
    CK(hipMalloc(&ctx->wnorm_max, 256));
    CK(hipMalloc(&ctx->xb, (size_t)mp * dp * 2));
    CK(hipMalloc(&ctx->eps, mp * 4));
    CK(hipMalloc(&ctx->thr, mp * 4));
    CK(hipMalloc(&ctx->gm, (size_t)IRS_MAX_GROUPS * mp * 4));
    CK(hipMalloc(&ctx->cand_cnt, (size_t)mp * IRS_CAND_BUCKETS * 4));
    CK(hipMalloc(&ctx->cand, (size_t)mp * IRS_CAND_CAP * 8));
    CK(hipMalloc(&ctx->ref_tmp, mp * 4));
    CK(hipMalloc(&ctx->step_ctr, 256));
    CK(hipMemset(ctx->step_ctr, 0, 256));
    float *val;
    int64_t *ids;
    int32_t *status;
    CK(hipMalloc(&val, (size_t)M * k * 4));
    CK(hipMalloc(&ids, (size_t)M * k * 8));
    CK(hipMalloc(&status, M * 4));
    hipStream_t s = 0;
    if (irs_launch_pack_w(ctx, s)) { printf("pack: %s\n", ctx->err); return 1; }
    // whole pipeline, timed
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<float> tt;
    for (int i = 0; i < reps + 2; ++i) {
        CK(hipEventRecord(e0, s));
        if (irs_launch_topk(ctx, x, M, k, IRS_SWEEP_BF16, val, ids, status, s)) { printf("topk: %s\n", ctx->err); return 1; }
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (i >= 2) tt.push_back(ms * 1e3f);
    }
    const double flops = 2.0 * d * (double)M * (double)N;
    printf("N=%lld d=%d M=%d: irs_score_topk %.1f us\n", (long long)N, d, M, median(tt));
    std::vector<int32_t> hs(M);
    CK(hipMemcpy(hs.data(), status, M * 4, hipMemcpyDeviceToHost));
    int fb = 0;
    for (int v : hs) fb += v & 1;
    // emitted candidates of the production EMIT (cand_cnt left by the last call)
    auto count_emitted = [&]() {
        std::vector<unsigned int> hc((size_t)mp * IRS_CAND_BUCKETS);
        CK(hipMemcpy(hc.data(), ctx->cand_cnt, hc.size() * 4, hipMemcpyDeviceToHost));
        unsigned long long t = 0;
        for (auto v : hc) t += v;
        return t;
    };
    {   // thresholds of the production pipeline per 128-row slice, against the previous kernels' (same sampled tiles)
        std::vector<float> t0(mp), t1(mp);
        CK(hipMemcpy(t0.data(), ctx->thr, mp * 4, hipMemcpyDeviceToHost));
        ctx->sweep_variant = 4;
        if (irs_launch_topk(ctx, x, M, k, IRS_SWEEP_BF16, val, ids, status, s)) { printf("topk: %s\n", ctx->err); return 1; }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(t1.data(), ctx->thr, mp * 4, hipMemcpyDeviceToHost));
        ctx->sweep_variant = 1;
        std::vector<float> t2(mp);
        if (irs_launch_topk(ctx, x, M, k, IRS_SWEEP_BF16, val, ids, status, s)) { printf("topk: %s\n", ctx->err); return 1; }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(t2.data(), ctx->thr, mp * 4, hipMemcpyDeviceToHost));
        ctx->sweep_variant = 0;
        int bad = 0, bad2 = 0;
        for (int i = 0; i < M; ++i) bad += t0[i] != t1[i], bad2 += t2[i] != t1[i];
        printf("  thresholds differing from the 32x32x16 ring's: %d of %d rows (32x32x16 ring vs the round-1 kernels: %d)\n", bad, M, bad2);
        if (bad)
            for (int q = 0; q < M; q += 128) {
                int b = 0;
                for (int i = q; i < q + 128 && i < M; ++i) b += t0[i] != t1[i];
                printf("    rows %4d..%4d: %3d differ, e.g. %g vs %g\n", q, q + 127, b, t0[q], t1[q]);
            }
        if (irs_launch_topk(ctx, x, M, k, IRS_SWEEP_BF16, val, ids, status, s)) { printf("topk: %s\n", ctx->err); return 1; }
        CK(hipDeviceSynchronize());
    }
    const unsigned long long base_emit = count_emitted();
    printf("  fallback rows %d, emitted %.1f per row\n", fb, (double)base_emit / M);
    // ---- single kernels with the production arguments
    SweepArgs a;
    sweep_common(ctx, a, x, M);
    a.thr = ctx->thr;
    a.cnt = ctx->cand_cnt;
    a.cand = ctx->cand;
    a.gm = ctx->gm;
    const int nt = ctx->n_tiles;
    auto time_it = [&](const char *name, auto &&fn, bool emits) {
        std::vector<float> ts;
        unsigned long long em = 0;
        for (int i = 0; i < reps + 2; ++i) {
            CK(hipMemsetAsync(ctx->cand_cnt, 0, (size_t)mp * IRS_CAND_BUCKETS * 4, s));
            CK(hipEventRecord(e0, s));
            fn();
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (i >= 2) ts.push_back(ms * 1e3f);
            if (i == 0 && emits) em = count_emitted();
        }
        const float us = median(ts);
        printf("  %-44s %8.1f us  %6.1f TF/s (%4.1f %% of 2.5 PF)%s\n", name, us, flops / us / 1e6, flops / us / 1e6 / 25.0,
               emits ? (em == base_emit ? "  emit ok" : "  EMIT MISMATCH") : "");
        if (emits && em != base_emit) printf("     emitted %llu vs %llu\n", em, base_emit);
    };
    {
        const int UBh = ub_bf16(ctx->KS);
        const int nub = (a.UT + UBh - 1) / UBh;
        SweepArgs e = a;
        sweep_decompose(e, 0, nt, nub, 0);
        for (int rep = 0; rep < 3; ++rep) {
            ctx->sweep_variant = 0;
            time_it("production EMIT (ring, 16x16x32)", [&]() { SweepArgs e2 = e; launch_sweep_bf16<MODE_EMIT>(ctx, e2, s); }, true);
            ctx->sweep_variant = 4;
            time_it("round-2 EMIT (ring, 32x32x16)", [&]() { SweepArgs e2 = e; launch_sweep_bf16<MODE_EMIT>(ctx, e2, s); }, true);
        }
        ctx->sweep_variant = 0;
    }
    if (ctx->KS == 8) stamps_ring16<8, 8, 4, 4>(ctx, a, nt, s);
    if (ctx->KS == 16) stamps_ring16<16, 4, 4, 4>(ctx, a, nt, s);
#define RING16(KS_, RT16_, NW_, NSLOT_, MODE_, R10_, EM_)                                                              \
    time_it((EM_) ? "ring16 RT16=" #RT16_ " NW=" #NW_ " slots=" #NSLOT_ " rounds/10=" #R10_ " EMIT"                   \
                  : "ring16 RT16=" #RT16_ " NW=" #NW_ " slots=" #NSLOT_ " rounds/10=" #R10_ " PRE",                   \
            [&]() { launch_ring16_lab<KS_, RT16_, NW_, NSLOT_, MODE_>(a, nt, R10_, s); }, EM_)
    for (int rep = 0; rep < 2; ++rep) {
        if (ctx->KS == 8) {
            RING16(8, 8, 4, 4, MODE_EMIT, 40, true);
            RING16(8, 4, 4, 4, MODE_EMIT, 40, true);
            RING16(8, 8, 4, 4, MODE_PRE, 10, false);
        } else if (ctx->KS == 16) {
            RING16(16, 4, 4, 3, MODE_EMIT, 40, true);
            RING16(16, 4, 4, 4, MODE_EMIT, 40, true);
            RING16(16, 4, 4, 4, MODE_PRE, 10, false);
        }
    }
#define RING(KS_, RT_, TPS_, NW_, WPS_, NSLOT_, MODE_, R10_, EM_)                                                          \
    time_it((EM_) ? "ring RT=" #RT_ " TPS=" #TPS_ " NW=" #NW_ " wps=" #WPS_ " slots=" #NSLOT_ " rounds/10=" #R10_ " EMIT" \
                  : "ring RT=" #RT_ " TPS=" #TPS_ " NW=" #NW_ " wps=" #WPS_ " slots=" #NSLOT_ " rounds/10=" #R10_ " PRE", \
            [&]() { launch_ring_lab<KS_, RT_, TPS_, NW_, WPS_, NSLOT_, MODE_>(a, nt, R10_, s); }, EM_)
    for (int rep = 0; rep < 2; ++rep) { // twice, interleaved: the box's clocks drift within a process
    if (ctx->KS == 8) {
        RING(8, 4, 1, 4, 2, 4, MODE_EMIT, 40, true);
        RING(8, 4, 1, 4, 2, 4, MODE_PRE, 10, false);
    } else if (ctx->KS == 16) {
        RING(16, 2, 1, 4, 2, 4, MODE_EMIT, 40, true);
        RING(16, 2, 1, 4, 2, 4, MODE_PRE, 10, false);
    }
    }
    return 0;
}
