// Persuasion-path search step and the cross-shard top-k merge, gfx950.
//
// Replaces the per-row Python body of IRSNN.get_seq_in_batch
// (/root/reference model/influentialRS.py:419-450): window filter of the top-100,
// greedy / top-sample_k choice, path record, window grow / shift.  Runs entirely
// on the device so that one step can be captured into a hipGraph
// (no .item() host round trips; the reference pays B x 20 of them per batch).
#include "irs_internal.h"

// One wave per row (the row body lives in irs_internal.h: the one-launch small-shard top-k runs it in its tail).
__global__ void __launch_bounds__(256) k_path_step(int64_t *__restrict__ seq, int32_t *__restrict__ hep, int B, int L,
                                                   const float *__restrict__ val, const int64_t *__restrict__ ids0,
                                                   int k, int step_arg, const int32_t *__restrict__ step_ptr,
                                                   float *__restrict__ paths, int path_ld, int sample, int sample_k,
                                                   unsigned long long seed, int32_t *__restrict__ status,
                                                   int32_t *__restrict__ step_next) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= B) return;
    const irs_path_args pa{seq, hep, L, paths, path_ld, sample, sample_k, seed, status, step_ptr, step_arg, step_next, 1};
    irs_path_step_row(pa, row, lane, val, ids0, k);
}

__global__ void k_inc(int32_t *ctr) {
    if (threadIdx.x == 0 && blockIdx.x == 0) ctr[0] += 1;
}

// merge W lists of k entries per row: one workgroup per row, bitonic sort of <= 2048 keys.
// PACKED: the lists arrive as the exchange step's 64-bit keys (irs_hip.h: irs_pack_topk).
__device__ __forceinline__ unsigned long long irs_topk_key(float v, int64_t id) {
    return id >= 0 ? (((unsigned long long)irs_fkey(v) << 32) | (0xFFFFFFFFu - (unsigned int)id)) : 0ull;
}

__global__ void __launch_bounds__(256) k_pack_topk(const float *__restrict__ val, const int64_t *__restrict__ ids, int64_t n,
                                                   unsigned long long *__restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) keys[i] = irs_topk_key(val[i], ids[i]);
}

template <bool PACKED>
__global__ void __launch_bounds__(256) k_merge(const float *__restrict__ val_in, const int64_t *__restrict__ ids_in,
                                               const unsigned long long *__restrict__ keys_in, int W, int M, int k,
                                               float *__restrict__ val, int64_t *__restrict__ ids) {
    __shared__ unsigned long long keys[2048];
    const int row = blockIdx.x, tid = threadIdx.x;
    const int n = W * k;
    int n2 = 2;
    while (n2 < n) n2 <<= 1;
    for (int i = tid; i < n2; i += 256) {
        unsigned long long key = 0ull;
        if (i < n) {
            const int w = i / k, c = i % k;
            const size_t at = ((size_t)w * M + row) * k + c;
            key = PACKED ? keys_in[at] : irs_topk_key(val_in[at], ids_in[at]);
        }
        keys[i] = key;
    }
    for (int size = 2; size <= n2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int i = tid; i < n2 / 2; i += 256) {
                int lo = 2 * i - (i & (stride - 1));
                int hi = lo + stride;
                bool desc = ((lo & size) == 0);
                unsigned long long a = keys[lo], b = keys[hi];
                if ((a < b) == desc) {
                    keys[lo] = b;
                    keys[hi] = a;
                }
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < k; i += 256) {
        unsigned long long kk = (i < n2) ? keys[i] : 0ull;
        if (kk != 0ull) {
            val[(size_t)row * k + i] = irs_unkey((unsigned int)(kk >> 32));
            ids[(size_t)row * k + i] = (int64_t)(0xFFFFFFFFu - (unsigned int)kk);
        } else {
            val[(size_t)row * k + i] = -INFINITY;
            ids[(size_t)row * k + i] = -1;
        }
    }
}

// ------------------------------------------------------------------ beam search step
// BUILD-DEFINED extension (the reference has no beam search; BASELINE config 5).
// Per user: W beams, each a window + cumulative log-probability.  One step:
//   for every live beam j: survivors = its top-k candidates (descending) not present in its
//   window; the first W survivors become candidates with score cum[j] + log p(item | beam j),
//   log p = e - max - log(sumexp) (row-wise log-softmax over the whole catalog);
//   the best W candidates by (score desc, parent beam asc, rank within parent asc) become the
//   new beams: window grown / shifted exactly like the greedy step, path extended.
// W == 1 reduces to the greedy step of k_path_step bit for bit (the single live beam's first
// survivor wins whatever its score; lse_* may then be null).
// One workgroup (one wave per beam, at most 16 waves) per user; state is ping-ponged (in -> out).
// A beam's candidates are taken 64 at a time into registers (one per lane) before the survivor scan, so the
// scan is a chain of v_readlane + ballot, not of dependent global loads; the W best of the W*W candidates are ordered
// by rank counting (no barriers) instead of a full sort.
#define BEAM_MAXW 32
__global__ void __launch_bounds__(1024) k_beam_step(const int64_t *__restrict__ seq_in, const int32_t *__restrict__ hep_in,
                                                    const double *__restrict__ cum_in, const float *__restrict__ paths_in,
                                                    const float *__restrict__ val, const int64_t *__restrict__ ids0,
                                                    const float *__restrict__ lse_max, const float *__restrict__ lse_sum,
                                                    int W, int L, int k, int step_arg, const int32_t *__restrict__ step_ptr,
                                                    int P, int64_t *__restrict__ seq_out, int32_t *__restrict__ hep_out,
                                                    double *__restrict__ cum_out, float *__restrict__ paths_out,
                                                    int32_t *__restrict__ status) {
    __shared__ double c_score[BEAM_MAXW * BEAM_MAXW];
    __shared__ int64_t c_item[BEAM_MAXW * BEAM_MAXW];
    __shared__ int c_order[BEAM_MAXW];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwave = nthr >> 6;
    const int step = step_ptr ? step_ptr[0] : step_arg;
    const int WW = W * W;
    for (int i = tid; i < WW; i += nthr) {
        c_score[i] = -INFINITY;
        c_item[i] = 0;
    }
    if (tid < BEAM_MAXW) c_order[tid] = WW; // "no candidate of this rank"
    __syncthreads();
    // phase 1: survivors of every live beam (one wave per beam)
    for (int j = wave; j < W; j += nwave) {
        const int row = b * W + j;
        const double cj = cum_in[row];
        int found = 0;
        if (cj > -INFINITY) {
            const int64_t *w = seq_in + (size_t)row * L;
            const int wl = hep_in[row] + 1;
            int64_t wv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int p = lane + 64 * i;
                wv[i] = (p < wl && p < L) ? w[p] : (int64_t)-1;
            }
            double norm = 0.0;
            if (lse_max) norm = (double)lse_max[row] + log((double)lse_sum[row]);
            bool more = true;
            for (int c0 = 0; c0 < k && found < W && more; c0 += 64) { // 64 candidates per round, one per lane
                const int cl = c0 + lane;
                const int64_t cid = cl < k ? ids0[(size_t)row * k + cl] : (int64_t)-1;
                const float cv = cl < k ? val[(size_t)row * k + cl] : 0.f;
                for (int c = 0; c < 64 && c0 + c < k && found < W; ++c) {
                    const int64_t id0 = __shfl(cid, c, 64);
                    if (id0 < 0) { // end of the list (fewer than k items on this shard / excluded)
                        more = false;
                        break;
                    }
                    const int64_t item = id0 + 1;
                    bool hit = false;
#pragma unroll
                    for (int q = 0; q < 4; ++q) hit |= (wv[q] == item);
                    if (!__any(hit)) {
                        const float v = __shfl(cv, c, 64);
                        if (lane == 0) {
                            c_score[j * W + found] = cj + ((double)v - norm);
                            c_item[j * W + found] = item;
                        }
                        ++found;
                    }
                }
            }
            if (found == 0 && lane == 0) atomicOr(&status[b], IRS_ROW_NO_CANDIDATE);
        }
    }
    __syncthreads();
    // phase 2: rank of every candidate by (score desc, index asc), index = parent * W + rank in parent; ranks < W survive
    for (int x = tid; x < WW; x += nthr) {
        const double sx = c_score[x];
        if (sx > -INFINITY) {
            int rank = 0;
            for (int y = 0; y < WW; ++y) {
                const double sy = c_score[y];
                rank += (sy > sx || (sy == sx && y < x)) ? 1 : 0;
            }
            if (rank < W) c_order[rank] = x;
        }
    }
    __syncthreads();
    // phase 3: materialise the new beams (one wave per new beam)
    for (int t = wave; t < W; t += nwave) {
        const int ci = c_order[t];
        const double sc = ci < WW ? c_score[ci] : -INFINITY;
        const int orow = b * W + t;
        int64_t *wo = seq_out + (size_t)orow * L;
        float *po = paths_out + (size_t)orow * P;
        if (!(sc > -INFINITY)) { // dead beam: keep a well-formed (copied) window, never selected again
            const int64_t *wi = seq_in + (size_t)(b * W) * L;
            for (int p = lane; p < L; p += 64) wo[p] = wi[p];
            for (int p = lane; p < P; p += 64) po[p] = 0.f;
            if (lane == 0) {
                cum_out[orow] = -INFINITY;
                hep_out[orow] = hep_in[b * W];
            }
            continue;
        }
        const int parent = ci / W;
        const int prow = b * W + parent;
        const int64_t item = c_item[ci];
        const int64_t *wi = seq_in + (size_t)prow * L;
        const float *pi = paths_in + (size_t)prow * P;
        const int he = hep_in[prow];
        if (he < L - 2) { // grow
            for (int p = lane; p < L; p += 64) wo[p] = (p == he + 1) ? item : wi[p];
            if (lane == 0) hep_out[orow] = he + 1;
        } else { // shift, target stays last
            for (int p = lane; p < L; p += 64) wo[p] = (p < L - 2) ? wi[p + 1] : (p == L - 2 ? item : wi[L - 1]);
            if (lane == 0) hep_out[orow] = he;
        }
        for (int p = lane; p < P; p += 64) po[p] = (p < step) ? pi[p] : (p == step ? (float)item : 0.f);
        if (lane == 0) cum_out[orow] = sc;
    }
}

// beam state initialisation: beam 0 = the input window with score 0, others dead copies
__global__ void k_beam_init(const int64_t *__restrict__ seq0, const int64_t *__restrict__ user0,
                            const int32_t *__restrict__ hep0, int B, int W, int L, int P, int64_t *__restrict__ seq,
                            int64_t *__restrict__ user, int32_t *__restrict__ hep, double *__restrict__ cum,
                            float *__restrict__ paths) {
    const int row = blockIdx.x; // b * W + j
    const int b = row / W, j = row % W;
    for (int p = threadIdx.x; p < L; p += blockDim.x) seq[(size_t)row * L + p] = seq0[(size_t)b * L + p];
    for (int p = threadIdx.x; p < P; p += blockDim.x) paths[(size_t)row * P + p] = 0.f;
    if (threadIdx.x == 0) {
        user[row] = user0 ? user0[b] : 0;
        hep[row] = hep0[b];
        cum[row] = (j == 0) ? 0.0 : -INFINITY;
    }
}

int irs_launch_beam_init(irs_ctx *ctx, const int64_t *seq0, const int64_t *user0, const int32_t *hep0, int B, int W,
                         int P, int64_t *seq, int64_t *user, int32_t *hep, double *cum, float *paths, hipStream_t s) {
    hipLaunchKernelGGL(k_beam_init, dim3(B * W), dim3(64), 0, s, seq0, user0, hep0, B, W, ctx->dims.max_len, P, seq,
                       user, hep, cum, paths);
    IRS_CHECK_HIP(ctx, hipGetLastError());
    return IRS_OK;
}

int irs_launch_beam_step(irs_ctx *ctx, const int64_t *seq_in, const int32_t *hep_in, const double *cum_in,
                         const float *paths_in, const float *val, const int64_t *ids0, const float *lse_max,
                         const float *lse_sum, int B, int W, int k, int step, const int32_t *step_ptr, int P,
                         int64_t *seq_out, int32_t *hep_out, double *cum_out, float *paths_out, int32_t *status,
                         hipStream_t s) {
    if (W < 1 || W > BEAM_MAXW) IRS_FAIL(ctx, IRS_E_UNSUPPORTED, "beam width %d outside [1, %d]", W, BEAM_MAXW);
    if (ctx->dims.max_len > 256) IRS_FAIL(ctx, IRS_E_UNSUPPORTED, "beam step: window length %d > 256", ctx->dims.max_len);
    const int waves = W < 4 ? 4 : (W > 16 ? 16 : W);
    hipLaunchKernelGGL(k_beam_step, dim3(B), dim3(64 * waves), 0, s, seq_in, hep_in, cum_in, paths_in, val, ids0, lse_max,
                       lse_sum, W, ctx->dims.max_len, k, step, step_ptr, P, seq_out, hep_out, cum_out, paths_out, status);
    IRS_CHECK_HIP(ctx, hipGetLastError());
    return IRS_OK;
}

int irs_launch_merge(irs_ctx *ctx, const float *val_in, const int64_t *ids_in, int W, int M, int k, float *val,
                     int64_t *ids0, hipStream_t s) {
    if (W * k > 2048) IRS_FAIL(ctx, IRS_E_UNSUPPORTED, "merge of %d x %d entries exceeds 2048", W, k);
    hipLaunchKernelGGL(k_merge<false>, dim3(M), dim3(256), 0, s, val_in, ids_in, (const unsigned long long *)nullptr, W, M, k, val, ids0);
    IRS_CHECK_HIP(ctx, hipGetLastError());
    return IRS_OK;
}

int irs_launch_merge_keys(irs_ctx *ctx, const uint64_t *keys_in, int W, int M, int k, float *val, int64_t *ids0, hipStream_t s) {
    if (W * k > 2048) IRS_FAIL(ctx, IRS_E_UNSUPPORTED, "merge of %d x %d entries exceeds 2048", W, k);
    hipLaunchKernelGGL(k_merge<true>, dim3(M), dim3(256), 0, s, (const float *)nullptr, (const int64_t *)nullptr,
                       reinterpret_cast<const unsigned long long *>(keys_in), W, M, k, val, ids0);
    IRS_CHECK_HIP(ctx, hipGetLastError());
    return IRS_OK;
}

int irs_launch_pack_topk(irs_ctx *ctx, const float *val, const int64_t *ids0, int64_t n, uint64_t *keys, hipStream_t s) {
    hipLaunchKernelGGL(k_pack_topk, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, val, ids0, n,
                       reinterpret_cast<unsigned long long *>(keys));
    IRS_CHECK_HIP(ctx, hipGetLastError());
    return IRS_OK;
}

int irs_launch_path_step(irs_ctx *ctx, int64_t *seq, int32_t *hep, int B, const float *val, const int64_t *ids0, int k,
                         int step, const int32_t *step_ptr, float *paths, int path_ld, int sample, int sample_k,
                         uint64_t seed, int32_t *status, hipStream_t s, int32_t *step_next) {
    hipLaunchKernelGGL(k_path_step, dim3((B + 3) / 4), dim3(256), 0, s, seq, hep, B, ctx->dims.max_len, val, ids0, k,
                       step, step_ptr, paths, path_ld, sample, sample_k, (unsigned long long)seed, status, step_next);
    IRS_CHECK_HIP(ctx, hipGetLastError());
    return IRS_OK;
}

int irs_launch_inc(irs_ctx *ctx, int32_t *ctr, hipStream_t s) {
    hipLaunchKernelGGL(k_inc, dim3(1), dim3(64), 0, s, ctr);
    IRS_CHECK_HIP(ctx, hipGetLastError());
    return IRS_OK;
}

// ------------------------------------------------------------------ evaluation batch on the device
// Replaces the per-user Python of DataProvider.get_random_evaluate_data (data_provider.py:398-449: history =
// all but the last event, label = the last event, target = a random item absent from the last raw_len
// history items) and DataLoaderEvalIRS._collate_fn (:591-617: pre-padded window, gap zeros, target last).
// One wave per user.  The target is drawn by rejection from a counter RNG (splitmix64 of (seed, user,
// attempt)) over [1, n_item] or over an explicit candidate pool (the reference's `popular_item` set):
// distributional parity with random.sample, exact parity of everything else.
__global__ void __launch_bounds__(64) k_build_eval_batch(const int64_t *__restrict__ items, const int64_t *__restrict__ offsets,
                                                         int B, int L, int raw_len, int gap_len, int64_t n_item,
                                                         const int64_t *__restrict__ targets_in, const int64_t *__restrict__ pool,
                                                         int64_t n_pool, unsigned long long seed, int64_t *__restrict__ seq,
                                                         int64_t *__restrict__ target, int64_t *__restrict__ label,
                                                         int64_t *__restrict__ raw, int32_t *__restrict__ raw_n,
                                                         int32_t *__restrict__ status) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int64_t lo = offsets[b], hi = offsets[b + 1];
    const int64_t n_hist = hi - lo - 1;                       // history = all but the last event
    const int rn = (int)(n_hist < raw_len ? (n_hist < 0 ? 0 : n_hist) : raw_len);
    const int64_t *rw = items + lo + (n_hist - rn);           // the raw window, oldest first
    int64_t tgt = 0;
    if (targets_in) {
        tgt = targets_in[b];
    } else {
        const int64_t space = pool ? n_pool : n_item;
        bool found = false;
        for (int attempt = 0; attempt < 256 && !found; ++attempt) {
            unsigned long long z = seed + 0x9E3779B97F4A7C15ull * ((unsigned long long)b * 2654435761ull + (unsigned long long)attempt + 1ull);
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            z = z ^ (z >> 31);
            const unsigned long long pick = __umul64hi(z, (unsigned long long)space); // uniform in [0, space)
            const int64_t cand = pool ? pool[pick] : (int64_t)pick + 1;
            bool hit = false;
            for (int i = lane; i < rn; i += 64) hit |= (rw[i] == cand);
            if (!__any(hit)) {
                tgt = cand;
                found = true;
            }
        }
        if (!found && lane == 0 && status) status[b] |= IRS_ROW_NO_CANDIDATE;
    }
    const int l_history = L - gap_len - 1;
    const int nh = rn < l_history ? rn : l_history;           // seq[-l_history:] of the raw window
    const int start = L - nh - gap_len - 1;
    int64_t *row = seq + (int64_t)b * L;
    for (int t = lane; t < L; t += 64) {
        int64_t v = 0;
        if (t >= start && t < start + nh) v = rw[rn - nh + (t - start)];
        if (t == L - 1) v = tgt;
        row[t] = v;
    }
    if (raw)
        for (int i = lane; i < raw_len; i += 64) raw[(int64_t)b * raw_len + i] = (i >= raw_len - rn) ? rw[i - (raw_len - rn)] : 0;
    if (lane == 0) {
        target[b] = tgt;
        label[b] = (hi > lo) ? items[hi - 1] : 0;
        if (raw_n) raw_n[b] = rn;
    }
}

int irs_launch_build_eval_batch(irs_ctx *ctx, const int64_t *items, const int64_t *offsets, int B, int raw_len, int gap_len,
                                const int64_t *targets_in, const int64_t *pool, int64_t n_pool, uint64_t seed, int64_t *seq,
                                int64_t *target, int64_t *label, int64_t *raw, int32_t *raw_n, int32_t *status,
                                hipStream_t s) {
    const int L = ctx->dims.max_len;
    if (L - gap_len - 1 < 1) IRS_FAIL(ctx, IRS_E_UNSUPPORTED, "gap_len %d leaves no history slot in a window of %d", gap_len, L);
    hipLaunchKernelGGL(k_build_eval_batch, dim3(B), dim3(64), 0, s, items, offsets, B, L, raw_len, gap_len,
                       (int64_t)ctx->dims.n_item, targets_in, pool, n_pool, (unsigned long long)seed, seq, target, label, raw,
                       raw_n, status);
    IRS_CHECK_HIP(ctx, hipGetLastError());
    return IRS_OK;
}
