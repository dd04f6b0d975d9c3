// Internal declarations shared by the HIP translation units of libirs_hip.so.
// gfx950 (CDNA4) only: wave = 64 lanes, MFMA 32x32 tiles, 160 KiB LDS per CU.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/irs_hip.h"

// Lab switches.  The kernels carry measurement hooks (phase stamps, fragment dumps, "run without the MFMAs / DMA / barrier"
// variants) that tools/*_lab.hip turn on by including a kernel source with the switch defined.  Several of them change what
// a kernel COMPUTES, so a stray -D must never produce a library: every switch is legal only together with IRS_LAB, which
// influentialrs_amd/build.py never defines (and tests/test_cabi.py checks that it does not).
#if !defined(IRS_LAB) && (defined(X6_DUMP) || defined(X6_STAMP) || defined(X6_NO_SPLIT) || defined(X6_STAGGER) ||            \
                          defined(X6_NO_MFMA) || defined(X6_NO_READS) || defined(X6_NO_DMA) || defined(X6_NO_BARRIER) ||       \
                          defined(X6_NW) || defined(ATTN_STAMP) || defined(ATTNP_STAMP) ||               \
                          defined(IRS_SMALL_TIMING) || defined(IRS_DIRECT_TIMING) || defined(SWEEP_LAB) || defined(X6D_STAMP) ||              \
                          defined(X6_NO_QKV_STORE) || defined(X6_NSLOT2) || defined(X6_RESID_LATE) || defined(X6_RING4) ||      \
                          defined(X6_SPLIT_ACC4) || defined(SEQ_EXP))
#error "a lab switch (X6_* / ATTN* / IRS_*_TIMING) is defined without IRS_LAB: the product library must be built without them"
#endif

#include <atomic>
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) and occupancy are PER DEVICE: the "done once" flags of the launchers are a
// bit per device ordinal (setting the attribute twice from two host threads is harmless, so a plain atomic mask suffices)
#define IRS_MAX_DEVICES 32
static inline int irs_cur_dev() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= IRS_MAX_DEVICES) d = 0;
    return d;
}
#define IRS_ONCE_PER_DEVICE(body_)                                                                                        \
    do {                                                                                                                  \
        static std::atomic<unsigned> once_mask_{0};                                                                       \
        const unsigned once_bit_ = 1u << irs_cur_dev();                                                                   \
        if (!(once_mask_.load(std::memory_order_acquire) & once_bit_)) {                                                  \
            body_;                                                                                                        \
            once_mask_.fetch_or(once_bit_, std::memory_order_release);                                                    \
        }                                                                                                                 \
    } while (0)

#define IRS_MAX_LAYERS 16
#define IRS_CAND_BUCKETS 64 // candidate lists per row (bucket = item tile mod 64)
#define IRS_CAND_SLOTS 64   // entries per bucket
#define IRS_CAND_CAP (IRS_CAND_BUCKETS * IRS_CAND_SLOTS) // emitted candidates kept per row by the sweep
#define IRS_REFINE_CAP 1024 // candidates exactly re-scored per row
#define IRS_MAX_GROUPS 2048 // pre-pass group maxima per row (upper bound: the pre-pass is decomposed to stay below it)
#define IRS_MAX_PATH 64     // beam-search path length bound
#define IRS_LSE_SLOTS_RING 4096 // (max, sum) partial slots of the <= 32-row ring log-sum-exp sweep: 2048 waves x 2 lane halves
#define IRS_COOP_FALLBACK_MIN_ITEMS 262144 // shards from this size up redo flagged rows cooperatively (score.hip: k_exh_strips)
#define IRS_EXH_SCRATCH_KEYS (64 * 32768)  // EXH_FB_MAX x EXH_KEYS_PER_ROW keys of scratch for it

struct irs_layer_w {
    const float *sa_in_w, *sa_in_b, *sa_out_w, *sa_out_b;
    const float *ca_in_w, *ca_in_b, *ca_out_w, *ca_out_b;
    const float *l1_w, *l1_b, *l2_w, *l2_b;
    const float *n1_w, *n1_b, *n2_w, *n2_b, *n3_w, *n3_b;
};

struct irs_prof_ev {
    hipEvent_t a, b;
};

struct irs_ctx {
    irs_dims dims;
    irs_shard shard;
    int64_t n_local; // items in this shard
    int d_pad;       // d rounded up to 16/32/64/128/256
    int KS;          // d_pad / 16 (bf16 MFMA k-steps)
    int n_tiles;     // ceil(n_local / 32)
    int max_seqs;    // sequences per decode call
    int max_rows;    // scored rows per call
    int m_pad_max;   // max_rows rounded up to 32

    // bound weights (device)
    const float *item_emb, *user_emb, *pe, *um_w, *um_b, *proj_w, *proj_b;
    irs_layer_w layer[IRS_MAX_LAYERS];

    // derived weights (device, in the caller's arena)
    uint4 *wp;        // bf16 fragments [n_tiles][KS][64] x 16 B
    float *bias_pad;  // [n_tiles*32], -inf beyond n_local
    float *c_l;       // [n_layers][d] cross-attention constants
    float *wnorm_max; // [3] max_j max(||W_j||, ||bf16(W_j)||), max_j ||W_j - bf16(W_j)||, max_j |b_j| (k_prep_x)
    float *w_frag16;  // fragment-packed layer weights of the 16-token latency kernel (d = 128, F = 256), or null
    uint4 *w_x6;      // split-bf16 step streams of the fused layer kernel k_block_x6 ([n_layers - 1] x 768 KB), or null
    int use_x6;       // decoder GEMMs of the throughput path on split-bf16 MFMAs (IRS_DECODER_GEMM=x6|f32)
    bool h3_ok;       // finalisation's float16 range bound holds (else IRS_GEMM_H3 runs as IRS_GEMM_X6 and V stays float32)
    float h3_bound;   // the largest operand magnitude the bound weights allow (irs_h3_operand_bound)
    int use_seq;      // sequence-resident decoder for the d = 128 throughput shape: 0 off, 1 on, 2 auto (default: from 1024 sequences up)
    bool seq_last;    // the last irs_decode took it
    int use_attn_h3;  // throughput attention on split-float16 MFMAs over K / V planes written by the layer kernel (default on; IRS_ATTN_GEMM=f32 off)
    int lse_no_ring;  // IRS_LSE_RING=0: the register-fragment log-sum-exp kernel at <= 32 rows too (A/B measurements, tests)
    bool finalized;
    bool proj_stale;  // a training entry point ran since irs_finalize_weights: wp / wnorm_max may lag project.*

    // workspace (device, caller-owned)
    char *ws;
    size_t ws_bytes;
    // decoder activations
    float *act_x, *act_y, *act_qkv, *act_ao, *act_h, *act_ru;
    float *act_qkv_b1; // second q | k | v buffer of the single-sequence fused-attention path (<= 256 rows)
    float *act_xf, *act_yf; // fragment-major copies of x / y (residual inputs of the LN-fused GEMMs)
    // packed (pad-free) decode plan
    int32_t *tok_row;  // [max_seqs * L] packed index -> b*L + t
    int32_t *seq_cnt, *seq_off, *seq_qrow; // [max_seqs]
    int32_t *seq_padq; // [max_seqs] index within the packed sequence of the one pad token it may hold (pos), or -1
    int32_t *m_dev;    // [1] number of packed rows
    // plan of the sequence-resident layer kernel (decoder.hip: k_plan_seq; null unless the shape supports it)
    int32_t *tile_seq, *tile_idx;   // [16 max_seqs] grid half tile -> sequence (-1: none), its block index
    int32_t *seq_row0, *qrow_tile;  // [max_seqs] first K / V image row in its workgroup; tile-order row of the consumed token
    int32_t *n_wg_dev;              // [1] workgroups in use
    // scoring
    uint4 *xb;          // packed bf16 rows [m_pad/32][KS][64] x 16 B
    float *eps;         // [m_pad]
    float *thr;         // [m_pad]
    float *gm;          // [m_pad / 4][n_groups <= IRS_MAX_GROUPS][4]
    unsigned int *cand_cnt; // [m_pad][IRS_CAND_BUCKETS]
    unsigned long long *cand; // [m_pad][IRS_CAND_CAP]
    float *lse_part;    // [lse_slots][m_pad][2]
    int lse_slots;
    float *ref_tmp;     // [m_pad]
    int thr_valid, thr_M, thr_k, thr_age, carry_period; // carried emission thresholds (irs_launch_topk)
    unsigned int *fb_count;        // [1] rows recorded for the cooperative exhaustive fallback
    int32_t *fb_list;              // [max_rows]
    unsigned long long *exh_keys;  // [EXH_FB_MAX][strips][k] per-strip lists (null on small shards)
    // path generation scratch
    float *xrows;       // [max_rows][d]
    float *top_val;     // [max_rows][max_k]
    int64_t *top_ids;   // [max_rows][max_k]
    int32_t *row_status;// [max_rows]
    int32_t *step_ctr;  // [2] current step, next step (hipGraph loops)
    int32_t *step_pair; // set around a decode launched by the merged small-batch path loop: the plan kernel copies
                        // step_pair[1] over step_pair[0]
    int32_t *pos_tmp;   // [max_seqs]
    // beam-search state, ping-pong [2]
    int64_t *bm_seq[2];  // [max_seqs][L]
    int32_t *bm_hep[2];  // [max_seqs]
    double *bm_cum[2];   // [max_seqs]
    float *bm_paths[2];  // [max_seqs][IRS_MAX_PATH]
    int64_t *bm_user;    // [max_seqs]
    float *lse_max, *lse_sum; // [max_rows]
    // item-sharded loops (comm.hip)
    float *x_local;           // [max_seqs][d] this rank's decoded rows (the all-gather's send buffer)
    uint64_t *keys_send, *keys_recv; // [max_rows][max_k] packed per-shard lists
    float *lse_gmax;          // [max_rows] all-reduced row maxima
    hipGraphExec_t sh_graph;  // captured sharded step (greedy: one step; beam: two)
    int sh_kind, sh_B, sh_W, sh_P, sh_k, sh_sweep, sh_sample, sh_sample_k, sh_nograph;
    int sh_overlap;           // irs_set_sharded_overlap: the greedy sharded loop runs two user micro-batches per step, collectives on sh_side
    hipStream_t sh_side;      // (created on first use)
    hipEvent_t sh_ev[8];
    uint64_t sh_seed;
    void *sh_comm, *sh_ptr[5];
    hipGraphExec_t beam_graph;
    int beam_B, beam_W, beam_k, beam_sweep, beam_P;
    void *beam_status; // the status buffer baked into the captured beam steps

    // graph cache for irs_generate_paths
    hipGraphExec_t graph_exec;
    int graph_B, graph_P, graph_k, graph_sweep, graph_sample, graph_sample_k;
    void *graph_seq, *graph_user, *graph_hep, *graph_paths, *graph_status;
    uint64_t graph_seed;

    int sweep_variant; // 0 = production kernels; 1 = the previous compute-bound bf16 sweep (development A/B only)

    // profiling
    int prof_family;
    irs_prof_ev *prof_ev;
    int prof_n, prof_cap;
    double prof_flops, prof_bytes;

    char err[512];
};

#define IRS_CHECK_HIP(ctx, expr)                                                                         \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess) {                                                                          \
            snprintf((ctx)->err, sizeof((ctx)->err), "%s:%d %s -> %s", __FILE__, __LINE__, #expr,        \
                     hipGetErrorString(_e));                                                             \
            return IRS_E_HIP;                                                                            \
        }                                                                                                \
    } while (0)

#define IRS_FAIL(ctx, code, ...)                                  \
    do {                                                          \
        snprintf((ctx)->err, sizeof((ctx)->err), __VA_ARGS__);    \
        return (code);                                            \
    } while (0)

// profiling bracket helpers (no-ops unless the family is enabled)
void irs_prof_begin(irs_ctx *ctx, int family, hipStream_t s);
void irs_prof_end(irs_ctx *ctx, int family, hipStream_t s, double flops, double bytes);

// ---- decoder.hip ----
int irs_launch_pif(irs_ctx *ctx, const int64_t *user, int B, float *r_u, hipStream_t s);
int irs_launch_decode(irs_ctx *ctx, const int64_t *seq, const int64_t *user, int B, float *x_out, const int32_t *pos,
                      float *xrows, float *r_u_out, hipStream_t s);
int irs_launch_cross_const(irs_ctx *ctx, hipStream_t s);
size_t irs_small_frag_floats(const irs_ctx *ctx);
int irs_launch_pack_small(irs_ctx *ctx, hipStream_t s);
size_t irs_x6_bytes(const irs_ctx *ctx);
int irs_launch_pack_x6(irs_ctx *ctx, hipStream_t s);
int irs_launch_h3_range(irs_ctx *ctx, float *stats, hipStream_t s);
float irs_h3_operand_bound(const irs_ctx *ctx, const float *stats);

// ---- score.hip ----
int irs_launch_pack_w(irs_ctx *ctx, hipStream_t s);
struct irs_path_args;
bool irs_topk_is_direct(const irs_ctx *ctx, int M, int k);
int irs_launch_topk(irs_ctx *ctx, const float *xrows, int M, int k, int sweep, float *val, int64_t *ids0,
                    int32_t *status, hipStream_t s, const irs_path_args *path = nullptr, float *lse_max = nullptr,
                    float *lse_sum = nullptr, int carry = 0);
int irs_launch_gather(irs_ctx *ctx, const float *xrows, int M, const int64_t *ids0, int g, float *out, hipStream_t s);
int irs_launch_count_before(irs_ctx *ctx, const float *xrows, int M, const float *ref_score, const int64_t *ref_id0,
                            const int64_t *excl, int n_excl, int64_t *count, hipStream_t s);
int irs_launch_dense(irs_ctx *ctx, const float *xrows, int M, float *out, int64_t ld, hipStream_t s);
int irs_launch_lse(irs_ctx *ctx, const float *xrows, int M, float *out_max, float *out_sum, hipStream_t s);
int irs_launch_refresh_bias(irs_ctx *ctx, hipStream_t s);
int irs_launch_lse_combine(irs_ctx *ctx, const float *mx, const float *sm, float *lse, int M, hipStream_t s);
int irs_launch_ce_reduce(irs_ctx *ctx, const float *lse, const float *lab_score, const int64_t *labels0, int M, double *out,
                         hipStream_t s);
int irs_launch_ce_grad(irs_ctx *ctx, const float *xrows, const int64_t *labels0, const float *lse, int M, float scale,
                       float *out, int64_t ld, hipStream_t s);

// ---- path.hip ----
int irs_launch_merge(irs_ctx *ctx, const float *val_in, const int64_t *ids_in, int W, int M, int k, float *val,
                     int64_t *ids0, hipStream_t s);
int irs_launch_merge_keys(irs_ctx *ctx, const uint64_t *keys_in, int W, int M, int k, float *val, int64_t *ids0, hipStream_t s);
int irs_launch_pack_topk(irs_ctx *ctx, const float *val, const int64_t *ids0, int64_t n, uint64_t *keys, hipStream_t s);
int irs_launch_path_step(irs_ctx *ctx, int64_t *seq, int32_t *hep, int B, const float *val, const int64_t *ids0, int k,
                         int step, const int32_t *step_ptr, float *paths, int path_ld, int sample, int sample_k,
                         uint64_t seed, int32_t *status, hipStream_t s, int32_t *step_next = nullptr);
int irs_launch_inc(irs_ctx *ctx, int32_t *ctr, hipStream_t s);
int irs_launch_beam_init(irs_ctx *ctx, const int64_t *seq0, const int64_t *user0, const int32_t *hep0, int B, int W,
                         int P, int64_t *seq, int64_t *user, int32_t *hep, double *cum, float *paths, hipStream_t s);
int irs_launch_beam_step(irs_ctx *ctx, const int64_t *seq_in, const int32_t *hep_in, const double *cum_in,
                         const float *paths_in, const float *val, const int64_t *ids0, const float *lse_max,
                         const float *lse_sum, int B, int W, int k, int step, const int32_t *step_ptr, int P,
                         int64_t *seq_out, int32_t *hep_out, double *cum_out, float *paths_out, int32_t *status,
                         hipStream_t s);

int irs_launch_build_eval_batch(irs_ctx *ctx, const int64_t *items, const int64_t *offsets, int B, int raw_len, int gap_len,
                                const int64_t *targets_in, const int64_t *pool, int64_t n_pool, uint64_t seed, int64_t *seq,
                                int64_t *target, int64_t *label, int64_t *raw, int32_t *raw_n, int32_t *status,
                                hipStream_t s);

// ---- small device helpers ----
#ifdef __HIPCC__
// float -> unsigned key whose unsigned order equals float order; -0 folded onto +0
// (identical to fkey() in oracle/oracle_score.c).
__device__ __forceinline__ unsigned int irs_fkey(float f) {
    unsigned int u = __float_as_uint(f);
    if (u == 0x80000000u) u = 0u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float irs_unkey(unsigned int k) {
    unsigned int u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(u);
}
// exact score: k-ascending float32 fma chain seeded with the bias.  The chain itself is serial; the LOADS of the item
// row are not: they are requested eight float4 at a time ahead of the fmas that consume them (with one load per
// loop trip a 128-float row cost 32 dependent memory round trips -- most of k_refine's time).
__device__ __forceinline__ float irs_chain(const float *__restrict__ x, const float *__restrict__ w, float b, int d) {
    float acc = b;
    if ((d & 3) == 0 && ((((uintptr_t)w) & 15) == 0)) {
        const float4 *w4 = reinterpret_cast<const float4 *>(w);
        const int n4 = d >> 2;
        int k = 0;
        for (; k + 8 <= n4; k += 8) {
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = w4[k + j];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc = __fmaf_rn(x[4 * (k + j) + 0], v[j].x, acc);
                acc = __fmaf_rn(x[4 * (k + j) + 1], v[j].y, acc);
                acc = __fmaf_rn(x[4 * (k + j) + 2], v[j].z, acc);
                acc = __fmaf_rn(x[4 * (k + j) + 3], v[j].w, acc);
            }
        }
        for (; k < n4; ++k) {
            float4 v = w4[k];
            acc = __fmaf_rn(x[4 * k + 0], v.x, acc);
            acc = __fmaf_rn(x[4 * k + 1], v.y, acc);
            acc = __fmaf_rn(x[4 * k + 2], v.z, acc);
            acc = __fmaf_rn(x[4 * k + 3], v.w, acc);
        }
    } else {
        for (int k = 0; k < d; ++k) acc = __fmaf_rn(x[k], w[k], acc);
    }
    return acc;
}

// ---- one row of the persuasion-path step (reference model/influentialRS.py:419-450), executed by one wave:
// window filter of the ranked candidates, greedy / sampled choice, path record, window grow / shift.
struct irs_path_args {
    int64_t *seq;
    int32_t *hep;
    int L;
    float *paths;
    int path_ld, sample, sample_k;
    unsigned long long seed;
    int32_t *status;
    const int32_t *step_ptr;
    int step_arg;
    int32_t *step_next;
    int enabled;
};
__device__ __forceinline__ void irs_path_step_row(const irs_path_args &p_, int row, int lane, const float *__restrict__ val,
                                                  const int64_t *__restrict__ ids0, int k) {
    const int L = p_.L, path_ld = p_.path_ld, sample = p_.sample;
    int sample_k = p_.sample_k;
    const unsigned long long seed = p_.seed;
    int64_t *seq = p_.seq;
    int32_t *hep = p_.hep, *status = p_.status, *step_next = p_.step_next;
    float *paths = p_.paths;
    const int step = p_.step_ptr ? p_.step_ptr[0] : p_.step_arg;
    // merged small-batch loop: publish the next step index in a second word (nobody reads it during this kernel;
    // the next step's first kernel copies it over step_ptr[0]) instead of a k_inc launch
    if (step_next && row == 0 && lane == 0) step_next[0] = step + 1;
    int64_t *w = seq + (size_t)row * L;
    const int he = hep[row];
    const int wl = he + 1; // window = seq[row, 0 .. he]
    // window into registers (L <= 256 -> 4 per lane)
    int64_t wv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int p = lane + 64 * i;
        wv[i] = (p < wl && p < L) ? w[p] : (int64_t)-1;
    }
    int64_t chosen = 0;
    int found = 0;
    float surv_val[IRS_MAX_SAMPLE_K];
    int64_t surv_id[IRS_MAX_SAMPLE_K];
    if (sample_k > IRS_MAX_SAMPLE_K) sample_k = IRS_MAX_SAMPLE_K; // the entry points reject larger values
    const int want = sample ? sample_k : 1;
    for (int c = 0; c < k && found < want; ++c) {
        int64_t id0 = ids0[(size_t)row * k + c];
        if (id0 < 0) break;
        int64_t item = id0 + 1;
        bool hit = false;
#pragma unroll
        for (int i = 0; i < 4; ++i) hit |= (wv[i] == item);
        if (!__any(hit)) {
            surv_val[found] = val[(size_t)row * k + c];
            surv_id[found] = item;
            ++found;
        }
    }
    if (found == 0) {
        if (lane == 0) status[row] |= IRS_ROW_NO_CANDIDATE;
        chosen = 0;
    } else if (!sample) {
        chosen = surv_id[0];
    } else {
        // multinomial over the first `found` survivors with weights exp(val) (softmax's
        // normaliser cancels, influentialRS.py:431-434).  Counter RNG: splitmix64(seed, row, step).
        unsigned long long z = seed + 0x9E3779B97F4A7C15ull * ((unsigned long long)row * 1315423911ull + (unsigned long long)step + 1ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
        float u = (float)((z >> 40) & 0xFFFFFF) * (1.0f / 16777216.0f);
        float mx = surv_val[0], tot = 0.f, p[IRS_MAX_SAMPLE_K];
        for (int i = 0; i < found; ++i) {
            p[i] = __expf(surv_val[i] - mx);
            tot += p[i];
        }
        float acc = 0.f;
        chosen = surv_id[found - 1];
        for (int i = 0; i < found; ++i) {
            acc += p[i] / tot;
            if (u < acc) {
                chosen = surv_id[i];
                break;
            }
        }
    }
    if (lane == 0 && step < path_ld) paths[(size_t)row * path_ld + step] = (float)chosen;
    if (found == 0) return;
    if (he < L - 2) { // room before the target: grow (influentialRS.py:438-441)
        if (lane == 0) {
            w[he + 1] = chosen;
            hep[row] = he + 1;
        }
    } else { // shift left by one, keep the target last (influentialRS.py:442-450)
        int64_t nv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int p = lane + 64 * i;
            nv[i] = (p + 1 <= L - 2) ? w[p + 1] : 0;
        }
        // all loads of this wave are complete before the stores (single wave, in-order memory ops per lane;
        // cross-lane overlap p <-> p+1 needs the explicit fence below)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_s_waitcnt(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int p = lane + 64 * i;
            if (p < L - 2) w[p] = nv[i];
        }
        if (lane == 0) w[L - 2] = chosen;
    }
}

// ---- cross-lane all-reductions without the LDS crossbar.  hipcc lowers every __shfl_xor to ds_bpermute_b32 (an
// LDS-pipe round trip, awaited at once when the next step depends on it); the latency kernels run chains of them.
// Here: DPP for the steps inside a 16-lane row (quad_perm for xor 1 / 2, row_ror:8 for xor 8; row_half_mirror reaches
// the other quad, so level 4 needs the quads to agree already -- true after levels 1 and 2, or when the data is
// uniform per quad), gfx950's v_permlane16_swap / v_permlane32_swap for the steps across rows.  STEPS says which of
// the six butterfly levels run (bit i = level 2^i).  Every lane of a reduced group ends with the same bits.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __uint_as_float((unsigned int)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xF, 0xF, true));
}
struct OpSum {
    __device__ __forceinline__ float operator()(float a, float b) const { return a + b; }
};
struct OpMax {
    __device__ __forceinline__ float operator()(float a, float b) const { return fmaxf(a, b); }
};
template <int STEPS, typename OP>
__device__ __forceinline__ float lanes_reduce(float v, OP op) {
    if (STEPS & 1) v = op(v, dpp_f32<0xB1>(v));  // quad_perm [1,0,3,2]
    if (STEPS & 2) v = op(v, dpp_f32<0x4E>(v));  // quad_perm [2,3,0,1]
    if (STEPS & 4) v = op(v, dpp_f32<0x141>(v)); // row_half_mirror
    if (STEPS & 8) v = op(v, dpp_f32<0x128>(v)); // row_ror:8 = lane ^ 8 inside the row, no agreement needed
    if (STEPS & 16) {
        const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = op(__uint_as_float(a[0]), __uint_as_float(a[1]));
    }
    if (STEPS & 32) {
        const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = op(__uint_as_float(b[0]), __uint_as_float(b[1]));
    }
    return v;
}
template <int STEPS>
__device__ __forceinline__ float lanes_sum(float v) { return lanes_reduce<STEPS>(v, OpSum()); }
template <int STEPS>
__device__ __forceinline__ float lanes_max(float v) { return lanes_reduce<STEPS>(v, OpMax()); }


#endif
