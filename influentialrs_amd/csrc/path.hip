// Persuasion-path search step and the cross-shard top-k merge, gfx950.
//
// Replaces the per-row Python body of IRSNN.get_seq_in_batch
// (/root/reference model/influentialRS.py:419-450): window filter of the top-100,
// greedy / top-sample_k choice, path record, window grow / shift.  Runs entirely
// on the device so that one step can be captured into a hipGraph
// (no .item() host round trips; the reference pays B x 20 of them per batch).
#include "irs_internal.h"

// One wave per row.
__global__ void __launch_bounds__(256) k_path_step(int64_t *__restrict__ seq, int32_t *__restrict__ hep, int B, int L,
                                                   const float *__restrict__ val, const int64_t *__restrict__ ids0,
                                                   int k, int step_arg, const int32_t *__restrict__ step_ptr,
                                                   float *__restrict__ paths, int path_ld, int sample, int sample_k,
                                                   unsigned long long seed, int32_t *__restrict__ status) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= B) return;
    const int step = step_ptr ? step_ptr[0] : step_arg;
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
    float surv_val[8];
    int64_t surv_id[8];
    if (sample_k > 8) sample_k = 8;
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
        float mx = surv_val[0], tot = 0.f, p[8];
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

__global__ void k_inc(int32_t *ctr) {
    if (threadIdx.x == 0 && blockIdx.x == 0) ctr[0] += 1;
}

// merge W lists of k entries per row: one workgroup per row, bitonic sort of <= 2048 keys
__global__ void __launch_bounds__(256) k_merge(const float *__restrict__ val_in, const int64_t *__restrict__ ids_in,
                                               int W, int M, int k, float *__restrict__ val, int64_t *__restrict__ ids) {
    __shared__ unsigned long long keys[2048];
    const int row = blockIdx.x, tid = threadIdx.x;
    const int n = W * k;
    int n2 = 2;
    while (n2 < n) n2 <<= 1;
    for (int i = tid; i < n2; i += 256) {
        unsigned long long key = 0ull;
        if (i < n) {
            int w = i / k, c = i % k;
            int64_t id = ids_in[((size_t)w * M + row) * k + c];
            float v = val_in[((size_t)w * M + row) * k + c];
            if (id >= 0) key = ((unsigned long long)irs_fkey(v) << 32) | (0xFFFFFFFFu - (unsigned int)id);
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

int irs_launch_merge(irs_ctx *ctx, const float *val_in, const int64_t *ids_in, int W, int M, int k, float *val,
                     int64_t *ids0, hipStream_t s) {
    if (W * k > 2048) IRS_FAIL(ctx, IRS_E_UNSUPPORTED, "merge of %d x %d entries exceeds 2048", W, k);
    hipLaunchKernelGGL(k_merge, dim3(M), dim3(256), 0, s, val_in, ids_in, W, M, k, val, ids0);
    IRS_CHECK_HIP(ctx, hipGetLastError());
    return IRS_OK;
}

int irs_launch_path_step(irs_ctx *ctx, int64_t *seq, int32_t *hep, int B, const float *val, const int64_t *ids0, int k,
                         int step, const int32_t *step_ptr, float *paths, int path_ld, int sample, int sample_k,
                         uint64_t seed, int32_t *status, hipStream_t s) {
    hipLaunchKernelGGL(k_path_step, dim3((B + 3) / 4), dim3(256), 0, s, seq, hep, B, ctx->dims.max_len, val, ids0, k,
                       step, step_ptr, paths, path_ld, sample, sample_k, (unsigned long long)seed, status);
    IRS_CHECK_HIP(ctx, hipGetLastError());
    return IRS_OK;
}

int irs_launch_inc(irs_ctx *ctx, int32_t *ctr, hipStream_t s) {
    hipLaunchKernelGGL(k_inc, dim3(1), dim3(64), 0, s, ctr);
    IRS_CHECK_HIP(ctx, hipGetLastError());
    return IRS_OK;
}
