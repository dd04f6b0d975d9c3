/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (influentialrs_amd/) never does.
 *
 * CPU restatement of the reference's score-all-items contraction and of the
 * selections made on it.  Reference call sites (/root/reference):
 *   project Linear          model/influentialRS.py:83, applied :214
 *                           (model/uRS.py:45, applied :68)
 *   topk(100) of one row    model/influentialRS.py:421
 *   sort desc + filter      model/influentialRS.py:375-389
 *   (Log)Softmax over N     model/influentialRS.py:418, model/evaluator.py:195
 *
 * The arithmetic itself lives in third-party PyTorch (requirements.txt:9 pins
 * pytorch=1.12.0; this image has 2.10.0), whose accumulation order inside
 * nn.Linear is unspecified.  This restatement FIXES the order so that results
 * are reproducible bit for bit on any machine:
 *
 *   e[j] = fmaf(x[d-1], W[j][d-1], ... fmaf(x[1], W[j][1], fmaf(x[0], W[j][0], b[j])))
 *
 * i.e. a k-ascending float32 fused-multiply-add chain seeded with the bias.
 * Ordering of items is the strict total order (score descending, id ascending);
 * torch's own tie order is unspecified.
 *
 * Pinned (tests/test_oracle_golden.py) against golden vectors produced by
 * importing the unmodified reference in the build container
 * (tests/golden/make_golden.py): logits agree to ~1e-6, top-100 ids identical
 * wherever the recorded adjacent margin exceeds the fp32 reordering noise.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* e[j] for one row x[d] over items [0, N). */
void orc_score_chain(const float *x, const float *W, const float *b, int64_t N, int d, float *out) {
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < N; ++j) {
        const float *w = W + j * (int64_t)d;
        float acc = b ? b[j] : 0.0f;
        for (int k = 0; k < d; ++k) acc = fmaf(x[k], w[k], acc);
        out[j] = acc;
    }
}

/* e[j] at selected 0-based items ids[g]. */
void orc_score_gather(const float *x, const float *W, const float *b, int d, const int64_t *ids, int g, float *out) {
    for (int i = 0; i < g; ++i) {
        int64_t j = ids[i];
        const float *w = W + j * (int64_t)d;
        float acc = b ? b[j] : 0.0f;
        for (int k = 0; k < d; ++k) acc = fmaf(x[k], w[k], acc);
        out[i] = acc;
    }
}

/* map float bits to an unsigned key whose unsigned order == float order
 * (-0.0 is folded onto +0.0; the device helper irs_fkey() is the same map). */
static inline uint32_t fkey(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if (u == 0x80000000u) u = 0u; /* -0 folded onto +0 */
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

typedef struct { uint32_t key; int64_t id; } ent_t;

static inline int before(const ent_t *a, const ent_t *b) { /* a ranks before b */
    return a->key > b->key || (a->key == b->key && a->id < b->id);
}

static void sift_down(ent_t *h, int n, int i) { /* min-heap on rank: root = worst kept */
    for (;;) {
        int l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && before(&h[m], &h[l])) m = l;
        if (r < n && before(&h[m], &h[r])) m = r;
        if (m == i) return;
        ent_t t = h[i]; h[i] = h[m]; h[m] = t;
        i = m;
    }
}

static int cmp_rank(const void *pa, const void *pb) {
    const ent_t *a = (const ent_t *)pa, *b = (const ent_t *)pb;
    if (before(a, b)) return -1;
    if (before(b, a)) return 1;
    return 0;
}

/* top-k of scores[0..N) by (score desc, id asc).  id_base is added to the
 * 0-based position (shard offset).  Writes min(k, N) entries; returns count.
 * NaN scores are not expected on this path. */
int orc_topk(const float *scores, int64_t N, int k, int64_t id_base, float *out_val, int64_t *out_idx) {
    if (k > N) k = (int)N;
    if (k <= 0) return 0;
    ent_t *h = (ent_t *)malloc(sizeof(ent_t) * (size_t)k);
    int n = 0;
    for (int64_t j = 0; j < N; ++j) {
        ent_t e = { fkey(scores[j]), j };
        if (n < k) {
            h[n++] = e;
            if (n == k) for (int i = k / 2 - 1; i >= 0; --i) sift_down(h, k, i);
        } else if (before(&e, &h[0])) {
            h[0] = e;
            sift_down(h, k, 0);
        }
    }
    qsort(h, (size_t)n, sizeof(ent_t), cmp_rank);
    for (int i = 0; i < n; ++i) {
        out_val[i] = scores[h[i].id];
        out_idx[i] = h[i].id + id_base;
    }
    free(h);
    return n;
}

/* rank of `label` (0-based item) among items not in hist (0-based ids, label
 * itself is never excluded by this routine): 1 + #{j notin hist : j ranks before label}.
 * Mirrors sort-desc + history filter + nonzero() of influentialRS.py:375-389
 * under the fixed total order. */
int64_t orc_rank(const float *scores, int64_t N, int64_t label, const int64_t *hist, int nh) {
    ent_t lab = { fkey(scores[label]), label };
    int64_t cnt = 0;
    for (int64_t j = 0; j < N; ++j) {
        ent_t e = { fkey(scores[j]), j };
        if (before(&e, &lab)) ++cnt;
    }
    for (int i = 0; i < nh; ++i) {
        int64_t j = hist[i];
        if (j < 0 || j >= N || j == label) continue;
        int dup = 0;
        for (int q = 0; q < i; ++q) if (hist[q] == j) { dup = 1; break; }
        if (dup) continue;
        ent_t e = { fkey(scores[j]), j };
        if (before(&e, &lab)) --cnt;
    }
    return cnt + 1;
}

/* max and sum(exp(s - max)) of a row, double accumulation of float32 terms
 * (the reference's softmax/log_softmax are torch's; tolerance 1e-6 relative). */
void orc_max_sumexp(const float *scores, int64_t N, float *out_max, double *out_sum) {
    float m = -INFINITY;
    for (int64_t j = 0; j < N; ++j) if (scores[j] > m) m = scores[j];
    double s = 0.0;
    for (int64_t j = 0; j < N; ++j) s += exp((double)scores[j] - (double)m);
    *out_max = m;
    *out_sum = s;
}

/* round-to-nearest-even float32 -> bfloat16 bits (inputs finite). */
uint16_t orc_bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

void orc_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
