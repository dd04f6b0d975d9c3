// Decoder of InfluentialNet / SampleNet on gfx950, float32 end to end.
//
// Replaces (reference, /root/reference):
//   embedding * sqrt(d) + PE          model/influentialRS.py:174-175, model/uRS.py:55
//   r_u = Linear(user_emb)            model/influentialRS.py:180
//   mask (as called) + key padding    model/influentialRS.py:120-155,171,183-186; uRS.py:47-53
//   6 x nn.TransformerDecoderLayer    model/influentialRS.py:67-74,189-193 (post-norm, relu, eps 1e-5)
//   cross-attention over zero memory  model/influentialRS.py:172-173  -> constant c_l
//
// Kernels: k_embed, k_pif, k_linear (fp32 MFMA 32x32x2, 128x128 tiles, fused
// bias / relu / residual), k_attn (per (sequence, head) with the mask computed
// in registers, never materialised), k_ln (LN1 [+ c_l, LN2] fused), k_gather_rows.
// fp32 MFMA is an exact k-ordered fma chain (no TF32-like path on gfx950), so
// decoder rows agree with the fp32 reference to ~1e-6.
#include "irs_internal.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;

// ------------------------------------------------------------------ embed
__global__ void __launch_bounds__(256) k_embed(const int64_t *__restrict__ seq, const float *__restrict__ E,
                                               const float *__restrict__ pe, float *__restrict__ x, int rows, int L,
                                               int d, float sqrtd, int64_t n_item) {
    int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (row >= rows) return;
    int64_t id = seq[row];
    if (id < 0) id = 0;
    if (id > n_item) id = n_item;
    int t = row % L;
    const float *e = E + id * (int64_t)d;
    const float *p = pe + (int64_t)t * d;
    float *o = x + (int64_t)row * d;
    for (int c = lane; c < d; c += 64) o[c] = __fadd_rn(__fmul_rn(e[c], sqrtd), p[c]);
}

// ------------------------------------------------------------------ r_u
__global__ void k_pif(const int64_t *__restrict__ user, const float *__restrict__ U, const float *__restrict__ w,
                      const float *__restrict__ b, float *__restrict__ r_u, int B, int ud, int64_t n_user) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    int64_t u = user[i];
    if (u < 0) u = 0;
    if (u >= n_user) u = n_user - 1;
    const float *e = U + u * (int64_t)ud;
    float acc = 0.f;
    for (int c = 0; c < ud; ++c) acc = __fmaf_rn(e[c], w[c], acc);
    r_u[i] = acc + b[0];
}

__global__ void k_fill(float *p, float v, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// c[l][i] = sum_j Wo[i][j] * bv[j] + bo[i]
__global__ void k_cross_const(const float *__restrict__ Wo, const float *__restrict__ b_in, const float *__restrict__ bo,
                              float *__restrict__ c, int d) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d) return;
    const float *bv = b_in + 2 * d;
    float acc = 0.f;
    for (int j = 0; j < d; ++j) acc = __fmaf_rn(Wo[(int64_t)i * d + j], bv[j], acc);
    c[i] = acc + bo[i];
}

// ------------------------------------------------------------------ packed (pad-free) decode plan
// Pad tokens are masked as keys and their rows are never consumed, yet they are ~half of an ml-1m-shaped
// batch (pre-padded windows, data_provider.py:591-617).  When only row pos[b] of every sequence is wanted,
// the decoder runs on the PACKED valid tokens: tok_row[m'] = b*L + t for every t with seq[b,t] != 0 (plus
// t = pos[b] itself), in (b, t) order; cnt[b] / off[b] delimit sequence b; qrow[b] is the packed index of
// (b, pos[b]); m_dev[0] = total.  Causality is order-preserving, so every kernel just works on shorter
// sequences.
// Three small kernels: count (one wave per sequence, ballots), scan (one workgroup), fill (one wave per sequence).
__device__ __forceinline__ bool plan_valid(const int64_t *__restrict__ sq, int t, int L, int p) {
    return t < L && (sq[t] != 0 || t == p);
}

__global__ void __launch_bounds__(256) k_plan_count(const int64_t *__restrict__ seq, const int32_t *__restrict__ pos, int B,
                                                    int L, int32_t *__restrict__ cnt, int32_t *__restrict__ tile_seq) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    if (tile_seq && lane < 16) tile_seq[16 * b + lane] = -1; // (the sequence-resident plan's table: 16 half tiles per possible workgroup)
    int p = pos[b];
    p = p < 0 ? 0 : (p >= L ? L - 1 : p);
    const int64_t *sq = seq + (int64_t)b * L;
    int n = 0;
    for (int t0 = 0; t0 < L; t0 += 64) n += __popcll(__ballot(plan_valid(sq, t0 + lane, L, p)));
    if (lane == 0) cnt[b] = n;
}

__global__ void __launch_bounds__(1024) k_plan_scan(const int32_t *__restrict__ cnt, int B, int32_t *__restrict__ off,
                                                    int32_t *__restrict__ m_dev) {
    __shared__ int s_scan[1024];
    __shared__ int s_base;
    const int tid = threadIdx.x;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int b0 = 0; b0 < B; b0 += 1024) {
        const int b = b0 + tid;
        const int n = (b < B) ? cnt[b] : 0;
        s_scan[tid] = n;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) { // inclusive Hillis-Steele scan
            int v = (tid >= o) ? s_scan[tid - o] : 0;
            __syncthreads();
            s_scan[tid] += v;
            __syncthreads();
        }
        const int base = s_base;
        if (b < B) off[b] = base + s_scan[tid] - n;
        __syncthreads();
        if (tid == 1023) s_base = base + s_scan[1023];
        __syncthreads();
    }
    if (tid == 0) m_dev[0] = s_base;
}

__global__ void __launch_bounds__(256) k_plan_fill(const int64_t *__restrict__ seq, const int32_t *__restrict__ pos, int B,
                                                   int L, const int32_t *__restrict__ off, int32_t *__restrict__ qrow,
                                                   int32_t *__restrict__ tok_row, int32_t *__restrict__ padq) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    int p = pos[b];
    p = p < 0 ? 0 : (p >= L ? L - 1 : p);
    const int64_t *sq = seq + (int64_t)b * L;
    int o = off[b];
    for (int t0 = 0; t0 < L; t0 += 64) {
        const int t = t0 + lane;
        const bool v = plan_valid(sq, t, L, p);
        const unsigned long long m = __ballot(v);
        if (v) {
            const int idx = o + __popcll(m & ((1ull << lane) - 1ull));
            tok_row[idx] = b * L + t;
            if (t == p) {
                qrow[b] = idx;
                padq[b] = (sq[t] == 0) ? idx - off[b] : -1; // the only pad a packed sequence can hold
            }
        }
        o += __popcll(m);
    }
}

// ------------------------------------------------------------------ plan of the sequence-resident layer kernel (round 5)
// k_block_x6<.., SEQ> wants every sequence inside ONE workgroup of 8 wave tiles = 16 half tiles of 16 tokens.  A sequence of nb
// 16-token blocks takes nb CONSECUTIVE half tiles of its workgroup, [slot, slot + nb) (the first version gave every sequence
// whole tiles: 0.84 lane efficiency on the bench's windows; in blocks: 0.915, tools/seq_pack_sim.py).  Any set of sequences with
// <= 16 blocks fits, so the packing is one-dimensional, best fit, largest first, done class by class (nb = 16 .. 1) with the
// open workgroups kept as pools per free-block count f: the items of a class fill the pools' free blocks smallest f first (a
// workgroup with f free blocks takes floor(f / nb) of them), the rest opens new workgroups.  Workgroups are created as
// contiguous id ranges and move between pools a prefix at a time, so a pool is a short list of ranges and item i of a class
// finds its (workgroup, first half tile) by arithmetic: every thread places its own items, one thread does the O(pools)
// bookkeeping between classes (a one-thread walk over 4096 sequences took 1.7 ms).
// Which block goes to which of the sequence's half tiles depends on (slot parity, nb) alone (seq_half_of_block): the two halves of
// a wave tile get MIRRORED blocks (i, nb - 1 - i) -- causal attention costs qb + 1 key tiles for block qb, so every wave of a
// sequence gets nb + 1 of them -- and the one or two halves left over at an odd start / end get the middle block(s) and share
// their tile with the neighbouring sequence's.  (The first version laid a workgroup out as "all pairs, then the odd blocks":
// member lists per workgroup in global memory, a returning atomic per sequence and a pass of dependent loads over them -- 80 us
// per plan, most of it those round trips.)
//   tile_seq[2 t + h] / tile_qb[2 t + h]: sequence (-1: none, preset by k_plan_count) and block index in lanes 16 h .. 16 h + 15 of
//   grid tile t = 8 wg + wave; seq_row0[b]: first row of the sequence in its workgroup's K / V images (16 x slot); qrow_tile[b]:
//   tile-order row of the consumed token; n_wg[0]: workgroups in use.
#define SEQ_WG_TILES 8
#define SEQ_WG_BLOCKS 16
#define SEQ_AUTO_MIN_SEQS 384
#define SEQ_RMAX 48
// half tile (0 .. nb - 1, relative to the sequence's first) of block blk of a sequence of nb blocks starting at an even / odd half
__host__ __device__ __forceinline__ int seq_half_of_block(bool slot_odd, int nb, int blk) {
    const int mir = nb - 1 - blk;
    if (!slot_odd) {
        if ((nb & 1) && blk == mir) return nb - 1;       // the middle block: the last half, alone
        return blk < mir ? 2 * blk : 2 * mir + 1;
    }
    if (nb & 1) {
        if (blk == mir) return 0;                         // the middle block: the first half (shares its tile with the neighbour)
        return blk < mir ? 1 + 2 * blk : 2 + 2 * mir;
    }
    if (blk == nb / 2 - 1) return 0;                      // even nb at an odd start: the two middle blocks are the single halves
    if (blk == nb / 2) return nb - 1;
    return blk < mir ? 1 + 2 * blk : 2 + 2 * mir;
}
#define SEQ_PLAN_PER_THREAD 8 // x 1024: the sequences the plan kernel's LDS tables hold (the launch takes up to 8192 sequences)
__global__ void __launch_bounds__(1024) k_plan_seq(const int32_t *__restrict__ cnt, const int32_t *__restrict__ off,
                                                   const int32_t *__restrict__ qrow, int B, int32_t *__restrict__ tile_seq,
                                                   int32_t *__restrict__ tile_qb, int32_t *__restrict__ seq_row0,
                                                   int32_t *__restrict__ qrow_tile, int32_t *__restrict__ n_wg, int tiles_cap) {
    constexpr int C = SEQ_WG_BLOCKS, PT = SEQ_PLAN_PER_THREAD;
    __shared__ int s_hist[C + 1], s_start[C + 2];
    __shared__ int p_n[C], p_s[C][SEQ_RMAX], p_l[C][SEQ_RMAX]; // pool f = 1 .. 15: ranges of workgroup ids with f free blocks
    __shared__ int c_cum[C + 1], c_q[C], c_exist, c_newbase, s_nwg;
    // the sequences sorted by block count (largest first) and each one's consumed token, in LDS: the class loop below loads nothing
    // from memory (the first version re-read a sorted index array and three per-sequence values there: a round trip per class)
    __shared__ unsigned short s_order[1024 * PT], s_pt[1024 * PT];
    const int tid = threadIdx.x;
    if (tid <= C) s_hist[tid] = 0;
    if (tid < C) p_n[tid] = 0;
    if (tid == 0) s_nwg = 0;
    __syncthreads();
    int my_T[PT], my_rk[PT];
#pragma unroll
    for (int j = 0; j < PT; ++j) {
        const int b = tid + 1024 * j;
        my_T[j] = 0, my_rk[j] = 0;
        if (b < B) {
            int nb = (cnt[b] + 15) >> 4;
            nb = nb < 1 ? 1 : (nb > C ? C : nb);
            my_T[j] = nb;
            int pt = qrow[b] - off[b];
            s_pt[b] = (unsigned short)(pt < 0 ? 0 : (pt > 65535 ? 65535 : pt));
        }
    }
#pragma unroll
    for (int j = 0; j < PT; ++j)
        if (my_T[j]) my_rk[j] = atomicAdd(&s_hist[my_T[j]], 1); // rank inside the class: the order of arrival (any order packs equally well)
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int T = C; T >= 1; --T) { // largest first
            s_start[T] = run;
            run += s_hist[T];
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PT; ++j)
        if (my_T[j]) s_order[s_start[my_T[j]] + my_rk[j]] = (unsigned short)(tid + 1024 * j);
    __syncthreads();
    auto pool_bins = [&](int f) {
        int n = 0;
        for (int r = 0; r < p_n[f]; ++r) n += p_l[f][r];
        return n;
    };
    auto append = [&](int f, int start, int len) { // (thread 0) `len` workgroups from `start` on now have f free blocks
        if (f < 1 || len < 1) return;
        if (p_n[f] > 0 && p_s[f][p_n[f] - 1] + p_l[f][p_n[f] - 1] == start) p_l[f][p_n[f] - 1] += len;
        else if (p_n[f] < SEQ_RMAX) p_s[f][p_n[f]] = start, p_l[f][p_n[f]] = len, ++p_n[f];
        // (a full list drops the range: those workgroups stay as filled as they are -- packing quality, never correctness)
    };
    auto take_prefix = [&](int f, int m, int dest) { // (thread 0) the first m workgroups of pool f move to pool dest
        for (int r = 0; r < p_n[f] && m > 0; ++r) {
            const int t = m < p_l[f][r] ? m : p_l[f][r];
            append(dest, p_s[f][r], t);
            p_s[f][r] += t, p_l[f][r] -= t, m -= t;
        }
        int w = 0;
        for (int r = 0; r < p_n[f]; ++r)
            if (p_l[f][r] > 0) p_s[f][w] = p_s[f][r], p_l[f][w] = p_l[f][r], ++w;
        p_n[f] = w;
    };
    for (int T = C; T >= 1; --T) { // largest first
        const int n = s_hist[T], qnew = C / T;
        if (n == 0) continue; // (uniform: s_hist is shared)
        if (tid == 0) {
            int tot = 0;
            for (int f = T; f < C; ++f) {
                c_cum[f] = tot, c_q[f] = f / T;
                tot += pool_bins(f) * c_q[f];
            }
            c_cum[C] = tot;
            c_exist = n < tot ? n : tot;
            c_newbase = s_nwg;
            s_nwg += (n - c_exist + qnew - 1) / qnew;
        }
        __syncthreads();
        const int exist = c_exist, base = s_start[T];
        for (int i = tid; i < n; i += 1024) {
            const int b = s_order[base + i];
            int w, slot;
            if (i < exist) {
                int f = T;
                while (f < C - 1 && i >= c_cum[f + 1]) ++f;
                const int s_ = i - c_cum[f];
                int bl = s_ / c_q[f];
                const int k = s_ % c_q[f];
                int r = 0;
                while (r < p_n[f] - 1 && bl >= p_l[f][r]) bl -= p_l[f][r], ++r;
                w = p_s[f][r] + bl;
                slot = (C - f) + k * T;
            } else {
                const int j_ = i - exist;
                w = c_newbase + j_ / qnew;
                slot = (j_ % qnew) * T;
            }
            if (w * SEQ_WG_TILES >= tiles_cap) continue; // (cannot happen: at most one workgroup per sequence)
            // the sequence's T blocks onto the half tiles [slot, slot + T) of workgroup w
            const int h0 = 2 * SEQ_WG_TILES * w + slot;
            for (int blk = 0; blk < T; ++blk) {
                const int h = h0 + seq_half_of_block(slot & 1, T, blk);
                tile_seq[h] = b, tile_qb[h] = blk;
            }
            seq_row0[b] = 16 * slot;
            const int pt = s_pt[b];
            int pb = pt >> 4; // the consumed token: block pb
            pb = pb < 0 ? 0 : (pb >= T ? T - 1 : pb);
            qrow_tile[b] = 16 * (h0 + seq_half_of_block(slot & 1, T, pb)) + (pt & 15);
        }
        __syncthreads();
        if (tid == 0) {
            int left = exist;
            for (int f = T; f < C && left > 0; ++f) {
                const int q = c_q[f], cap = pool_bins(f) * q;
                const int used = left < cap ? left : cap;
                left -= used;
                const int full = used / q, part = used % q;
                take_prefix(f, full, f - T * q);
                if (part) take_prefix(f, 1, f - T * part);
            }
            const int nnew = n - exist, fullnew = nnew / qnew, partnew = nnew % qnew;
            append(C - T * qnew, c_newbase, fullnew);
            if (partnew) append(C - T * partnew, c_newbase + fullnew, 1);
        }
        __syncthreads();
    }
    if (tid == 0) n_wg[0] = s_nwg;
}

// Few sequences (the latency path): count, scan and fill in ONE workgroup of 16 waves, together with the
// personalised impressionability factor r_u (k_pif) and, inside a hipGraph path loop, the hand-over of the step
// counter -- five launches of ~4.5 us each become one.
__global__ void __launch_bounds__(1024) k_plan_small(const int64_t *__restrict__ seq, const int32_t *__restrict__ pos, int B, int L,
                                                     int32_t *__restrict__ cnt, int32_t *__restrict__ off,
                                                     int32_t *__restrict__ qrow, int32_t *__restrict__ tok_row,
                                                     int32_t *__restrict__ padq, int32_t *__restrict__ m_dev,
                                                     const int64_t *__restrict__ user, const float *__restrict__ U,
                                                     const float *__restrict__ uw, const float *__restrict__ ub,
                                                     float *__restrict__ r_u, int ud, int64_t n_user,
                                                     int32_t *__restrict__ step_pair) {
    __shared__ int s_cnt[64], s_off[64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int b = wave; b < B; b += 16) {
        int p = pos[b];
        p = p < 0 ? 0 : (p >= L ? L - 1 : p);
        const int64_t *sq = seq + (int64_t)b * L;
        int n = 0;
        for (int t0 = 0; t0 < L; t0 += 64) n += __popcll(__ballot(plan_valid(sq, t0 + lane, L, p)));
        if (lane == 0) s_cnt[b] = n;
    }
    if (tid < B) { // r_u = user_mask_layer(user_embedder(user)) (influentialRS.py:180), 0 without the user factor
        float acc = 0.f;
        if (U) {
            int64_t u = user[tid];
            if (u < 0) u = 0;
            if (u >= n_user) u = n_user - 1;
            const float *e = U + u * (int64_t)ud;
            for (int c = 0; c < ud; ++c) acc = __fmaf_rn(e[c], uw[c], acc);
            acc += ub[0];
        }
        r_u[tid] = acc;
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int b = 0; b < B; ++b) {
            s_off[b] = run;
            run += s_cnt[b];
        }
        m_dev[0] = run;
        if (step_pair) step_pair[0] = step_pair[1];
    }
    __syncthreads();
    for (int b = wave; b < B; b += 16) {
        int p = pos[b];
        p = p < 0 ? 0 : (p >= L ? L - 1 : p);
        const int64_t *sq = seq + (int64_t)b * L;
        int o = s_off[b];
        if (lane == 0) {
            cnt[b] = s_cnt[b];
            off[b] = o;
        }
        const int o0 = o;
        for (int t0 = 0; t0 < L; t0 += 64) {
            const int t = t0 + lane;
            const bool v = plan_valid(sq, t, L, p);
            const unsigned long long m = __ballot(v);
            if (v) {
                const int idx = o + __popcll(m & ((1ull << lane) - 1ull));
                tok_row[idx] = b * L + t;
                if (t == p) {
                    qrow[b] = idx;
                    padq[b] = (sq[t] == 0) ? idx - o0 : -1;
                }
            }
            o += __popcll(m);
        }
    }
}

__global__ void __launch_bounds__(256) k_embed_packed(const int64_t *__restrict__ seq, const float *__restrict__ E,
                                                      const float *__restrict__ pe, float *__restrict__ x,
                                                      const int32_t *__restrict__ tok_row, const int32_t *__restrict__ m_dev,
                                                      int L, int d, float sqrtd, int64_t n_item) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= m_dev[0]) return;
    const int orig = tok_row[row];
    int64_t id = seq[orig];
    if (id < 0) id = 0;
    if (id > n_item) id = n_item;
    const float *e = E + id * (int64_t)d;
    const float *p = pe + (int64_t)(orig % L) * d;
    float *o = x + (int64_t)row * d;
    for (int c = lane; c < d; c += 64) o[c] = __fadd_rn(__fmul_rn(e[c], sqrtd), p[c]);
}

__global__ void k_gather_rows_idx(const float *__restrict__ x, const int32_t *__restrict__ rowidx, float *__restrict__ out,
                                  int d) {
    const float *src = x + (int64_t)rowidx[blockIdx.x] * d;
    for (int c = threadIdx.x; c < d; c += blockDim.x) out[(int64_t)blockIdx.x * d + c] = src[c];
}

// Fragment-major embed (see frag_index below): one wave per 32-token tile; per (column group) every lane
// reads 16 bytes of its token's item row and of its position row and the wave stores one contiguous KiB.
// Rows beyond the packed / total count and columns beyond d are written as zeros.
__global__ void __launch_bounds__(256) k_embed_frag(const int64_t *__restrict__ seq, const float *__restrict__ E,
                                                    const float *__restrict__ pe, float *__restrict__ xf,
                                                    const int32_t *__restrict__ tok_row, const int32_t *__restrict__ m_dev,
                                                    int rows, int L, int d, float sqrtd, int64_t n_item) {
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, li = lane & 31, lk = lane >> 5;
    const int M = m_dev ? min(rows, m_dev[0]) : rows;
    if (tile * 32 >= M) return;
    const int row = tile * 32 + li;
    const bool live = row < M;
    const int orig = live ? (tok_row ? tok_row[row] : row) : 0;
    int64_t id = live ? seq[orig] : 0;
    if (id < 0) id = 0;
    if (id > n_item) id = n_item;
    const float *e = E + id * (int64_t)d;
    const float *p = pe + (int64_t)(orig % L) * d;
    float4 *o = reinterpret_cast<float4 *>(xf) + (size_t)tile * 16 * 64 + lane;
#pragma unroll 4
    for (int c = 0; c < 16; ++c) { // c = tn * 4 + g
        const int n = (c >> 2) * 32 + (c & 3) * 8 + 4 * lk;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live && n < d) { // d is a multiple of 4 on this path
            v.x = __fadd_rn(__fmul_rn(e[n + 0], sqrtd), p[n + 0]);
            v.y = __fadd_rn(__fmul_rn(e[n + 1], sqrtd), p[n + 1]);
            v.z = __fadd_rn(__fmul_rn(e[n + 2], sqrtd), p[n + 2]);
            v.w = __fadd_rn(__fmul_rn(e[n + 3], sqrtd), p[n + 3]);
        }
        o[c * 64] = v;
    }
}

// out[b, :] = act[rowidx[b], :] read from the fragment-major layout
// (nt = column tiles of 32 per token tile: 4 for every width up to 128 -- the image is padded to 128 columns --, d / 32 beyond)
__global__ void k_gather_rows_frag(const float *__restrict__ xf, const int32_t *__restrict__ rowidx,
                                   float *__restrict__ out, int d, int nt) {
    const int row = rowidx[blockIdx.x];
    for (int c = threadIdx.x; c < d; c += blockDim.x)
        out[(int64_t)blockIdx.x * d + c] =
            xf[(((size_t)(row >> 5) * nt + (c >> 5)) * 4 + ((c >> 3) & 3)) * 256 + ((c >> 2) & 1) * 128 + (row & 31) * 4 + (c & 3)];
}

// ------------------------------------------------------------------ linear
// Y[M,N] = epilogue(X[M,K] . W[N,K]^T + bias[N]);  X, W, Y row-major fp32.
// 256 threads = 4 waves (2x2); wave tile 64x64 = 2x2 v_mfma_f32_32x32x2_f32 tiles; block tile
// 128x128, K staged in slabs of 32 through LDS (double buffered, rows padded to 36 floats so
// that staging stores are 16-byte and fragment reads stay cheap next to 64-cycle MFMAs).
// The accumulator tile is staged through the same LDS for a coalesced float4 epilogue:
//   EPI_BIAS       y = acc + b (relu optional)
//   EPI_RES_LN     z = acc + b + R;  y = LN(z; g1, b1);  if (c) y = LN(y + c; g2, b2)
//                  (needs the whole row in the tile: N <= 128, grid.x == 1) -- this is
//                  out-proj + LN1 + (x + c_l) + LN2, and FFN2 + LN3, of the post-norm layer.
#define LIN_BM 128
#define LIN_BN 128

struct LinArgs {
    const float *X, *W, *bias, *R;
    float *Y;
    int M, N, K;
    int relu;
    const float *g1, *b1, *c, *g2, *b2; // fused LayerNorm parameters
    const float *Rf; // residual in fragment-major layout (may be null -> row-major R is read, strided)
    float *Yf;       // optional fragment-major copy of the output (next LN-GEMM's residual)
    const int32_t *m_dev; // optional device-side row count (packed decode): rows >= *m_dev do not exist
    int slots;            // workgroups of this kernel the device holds at once (k_linear's work-unit split)
    const float *Xf;      // X operand in fragment-major layout (K <= 128); X is then unused
};

// LDS slab image: [128 rows][32 floats], 16-byte chunk c of row r stored at chunk c ^ ((r >> 1) & 7).
// A wave's ds_read_b128 of (row = lane&31 (+const), chunk 2q + (lane>>5)) is then bank-conflict free,
// and the staging stores are whole 16-byte chunks.  MFMA k-slots: for the 8 k of group q, half 0
// supplies k = 8q + t and half 1 supplies k = 8q + 4 + t in step t (A and B agree, so any
// assignment of k to slots is a valid contraction).
// FULL = aligned operands and K a multiple of the slab depth: unconditional 16-byte loads.  Rows beyond
// `rows` are clamped to the last row (their products land in output rows / columns that are never stored),
// so the loader has no branches and no per-load waits.  !FULL = fully guarded generic path.
template <bool FULL, int BK>
__device__ __forceinline__ void lin_load_tile(const float *__restrict__ P, int rows, int K, int r0, int k0, int tid,
                                              float4 (&v)[BK / 8]) {
    constexpr int CH = BK / 4; // 16-byte chunks per row
#pragma unroll
    for (int i = 0; i < BK / 8; ++i) {
        const int idx = tid + i * 256;
        const int r = idx / CH, c = (idx % CH) * 4;
        if (FULL) {
            const int gr = min(r0 + r, rows - 1);
            const float *p = P + (int64_t)gr * K + k0 + c;
            float4 t;
            t.x = p[0], t.y = p[1], t.z = p[2], t.w = p[3];
            v[i] = t;
        } else {
            const int gr = r0 + r, gk = k0 + c;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gr < rows) {
                const float *p = P + (int64_t)gr * K + gk;
                if (gk < K) t.x = p[0];
                if (gk + 1 < K) t.y = p[1];
                if (gk + 2 < K) t.z = p[2];
                if (gk + 3 < K) t.w = p[3];
            }
            v[i] = t;
        }
    }
}

// swizzle of the 16-byte chunk index: 128-byte rows (BK=32) use (row>>1)&7, 64-byte rows (BK=16) use (row>>2)&3;
// both make a wave's ds_read_b128 of (row = lane&31, chunk 2q + (lane>>5)) bank-conflict free.
template <int BK>
__device__ __forceinline__ int lin_swz(int r) { return BK == 32 ? ((r >> 1) & 7) : ((r >> 2) & 3); }

template <int BK>
__device__ __forceinline__ void lin_store_tile(float *S, int tid, const float4 (&v)[BK / 8]) {
    constexpr int CH = BK / 4;
#pragma unroll
    for (int i = 0; i < BK / 8; ++i) {
        int idx = tid + i * 256;
        int r = idx / CH, c = idx % CH;
        *reinterpret_cast<float4 *>(S + r * BK + ((c ^ lin_swz<BK>(r)) << 2)) = v[i];
    }
}

// Fragment-major activations ("xf" layout): tokens in tiles of 32, 128 (zero-padded) columns per token,
//   float4 index = ((tile * 4 + tn) * 4 + g) * 64 + lk * 32 + li   holds   act[tile*32 + li][tn*32 + 8g + 4lk + 0..3].
// It is the register image of the transposed MFMA C layout (a lane owns a token), so the LN-fused GEMMs write
// it and read their residual from it with contiguous 1 KiB wave accesses, and a [32 tokens][8 k] piece of a
// k-slab is one contiguous KiB too: the GEMM loaders below read it as the X operand, which is why no
// row-major copy of x / y exists between the layers.
__device__ __forceinline__ size_t frag_index(int token, int n) {
    return (((size_t)(token >> 5) * 4 + (n >> 5)) * 4 + ((n >> 3) & 3)) * 256 + ((n >> 2) & 1) * 128 + (token & 31) * 4 + (n & 3);
}

template <int BK>
__device__ __forceinline__ void lin_load_tile_frag(const float *__restrict__ Pf, int m0, int k0, int tid,
                                                   float4 (&v)[BK / 8]) {
#pragma unroll
    for (int i = 0; i < BK / 8; ++i) {
        const int idx = tid + i * 256;
        const int t = idx / (8 * BK), j = idx % (8 * BK); // 32-token tile of the 128-row block, float4 within its slab
        const int kk = k0 + 8 * (j >> 6);
        const float *p = Pf + ((((size_t)((m0 >> 5) + t) * 4 + (kk >> 5)) * 4 + ((kk >> 3) & 3)) * 64 + (j & 63)) * 4;
        float4 q;
        q.x = p[0], q.y = p[1], q.z = p[2], q.w = p[3];
        v[i] = q;
    }
}

template <int BK>
__device__ __forceinline__ void lin_store_tile_frag(float *S, int tid, const float4 (&v)[BK / 8]) {
#pragma unroll
    for (int i = 0; i < BK / 8; ++i) {
        const int idx = tid + i * 256;
        const int t = idx / (8 * BK), j = idx % (8 * BK);
        const int r = t * 32 + (j & 31), c = 2 * (j >> 6) + ((j >> 5) & 1);
        *reinterpret_cast<float4 *>(S + r * BK + ((c ^ lin_swz<BK>(r)) << 2)) = v[i];
    }
}

__device__ __forceinline__ float wave_sum(float v) { return lanes_sum<63>(v); }

// One workgroup owns 128 rows and walks every 128-column block of the output: the (column block, k-slab)
// steps form ONE software pipeline, so the global-load latency of a block's first slab and the store tail of
// the previous block hide behind MFMA work instead of costing a prologue/epilogue bubble per output tile,
// and the X rows are re-read from the workgroup's own L2 (same XCD) rather than by workgroups on other XCDs.
// XF = the X operand comes from the fragment-major layout (a.Xf) instead of row-major a.X.
template <bool FULL, int BK, bool XF>
__global__ void __launch_bounds__(256, BK == 16 ? 4 : 2) k_linear(LinArgs a) {
    // staging: [stage][operand][128][BK] floats (BK=16: 32 KB, <=128 VGPRs -> 4 workgroups per CU)
    __shared__ __attribute__((aligned(16))) float sm[2 * 2 * LIN_BM * BK];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, lk = lane >> 5;
    const int M = a.m_dev ? min(a.M, a.m_dev[0]) : a.M, N = a.N, K = a.K;
    // work units: the first `full` workgroups (whole rounds of the device's resident slots) own a 128-row block
    // and all of its column blocks; the remaining row blocks are cut into one-column-block units, so the last,
    // partially filled round costs 1/ntn of a full one
    const int ntn = (N + LIN_BN - 1) / LIN_BN, ntm = (M + LIN_BM - 1) / LIN_BM;
    const int full = (ntm / a.slots) * a.slots;
    int mt = blockIdx.x, nb = 0, ne = ntn;
    if ((int)blockIdx.x >= full) {
        const int u = blockIdx.x - full;
        mt = full + u / ntn, nb = u % ntn, ne = nb + 1;
    }
    if (mt >= ntm) return;
    const int m0 = mt * LIN_BM;
    auto Xs = [&](int st) { return sm + (st * 2 + 0) * LIN_BM * BK; };
    auto Ws = [&](int st) { return sm + (st * 2 + 1) * LIN_BM * BK; };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 xv[BK / 8], wv[BK / 8];
    if (XF) lin_load_tile_frag<BK>(a.Xf, m0, 0, tid, xv);
    else lin_load_tile<FULL, BK>(a.X, M, K, m0, 0, tid, xv);
    lin_load_tile<FULL, BK>(a.W, N, K, nb * LIN_BN, 0, tid, wv);
    if (XF) lin_store_tile_frag<BK>(Xs(0), tid, xv);
    else lin_store_tile<BK>(Xs(0), tid, xv);
    lin_store_tile<BK>(Ws(0), tid, wv);
    __syncthreads();
    const int nkt = (K + BK - 1) / BK;
    const int nsteps = (ne - nb) * nkt;
    const int sw = lin_swz<BK>(li);
    int cur = 0, kt = 0, n0 = nb * LIN_BN;
    for (int s = 0; s < nsteps; ++s) {
        int kt1 = kt + 1, n1 = n0;
        if (kt1 == nkt) kt1 = 0, n1 = n0 + LIN_BN;
        const bool more = s + 1 < nsteps;
        if (more) {
            if (XF) lin_load_tile_frag<BK>(a.Xf, m0, kt1 * BK, tid, xv);
            else lin_load_tile<FULL, BK>(a.X, M, K, m0, kt1 * BK, tid, xv);
            lin_load_tile<FULL, BK>(a.W, N, K, n1, kt1 * BK, tid, wv);
        }
        const float *xa = Xs(cur) + (wr * 64 + li) * BK;
        const float *wb = Ws(cur) + (wc * 64 + li) * BK;
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            const int off = ((2 * q + lk) ^ sw) << 2;
            const float4 a0 = *reinterpret_cast<const float4 *>(xa + off);
            const float4 a1 = *reinterpret_cast<const float4 *>(xa + 32 * BK + off);
            const float4 b0 = *reinterpret_cast<const float4 *>(wb + off);
            const float4 b1 = *reinterpret_cast<const float4 *>(wb + 32 * BK + off);
#define LIN_STEP(E)                                                                       \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.E, b0.E, acc[0][0], 0, 0, 0);     \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.E, b1.E, acc[0][1], 0, 0, 0);     \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.E, b0.E, acc[1][0], 0, 0, 0);     \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.E, b1.E, acc[1][1], 0, 0, 0);
            LIN_STEP(x) LIN_STEP(y) LIN_STEP(z) LIN_STEP(w)
#undef LIN_STEP
        }
        if (kt == nkt - 1) {
            // epilogue straight from registers: C/D layout col = lane&31, row = (reg&3)+8*(reg>>2)+4*(lane>>5);
            // one store instruction covers two 128-byte row segments.  The row base is made opaque so the
            // 64 output addresses are formed here, not hoisted out of the pipeline loop into live registers.
            int mrow = m0 + wr * 64 + 4 * lk;
            asm volatile("" : "+v"(mrow));
            const bool interior = (m0 + LIN_BM <= M) && (n0 + LIN_BN <= N);
            if (interior && a.R == nullptr) { // workgroup-uniform: no per-element masks, no loads -> 64 back-to-back stores
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) {
                    const int n = n0 + wc * 64 + tn * 32 + li;
                    const float bv = a.bias ? a.bias[n] : 0.f;
                    float *yp = a.Y + (int64_t)mrow * N + n;
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int dm = tm * 32 + (r & 3) + 8 * (r >> 2);
                            float v = acc[tm][tn][r] + bv;
                            v = a.relu ? fmaxf(v, 0.f) : v;
                            yp[(int64_t)dm * N] = v;
                            acc[tm][tn][r] = 0.f;
                        }
                }
            } else if (interior) { // residual: 8 loads in flight, then 8 stores (not a load-wait-store chain per element)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) {
                    const int n = n0 + wc * 64 + tn * 32 + li;
                    const float bv = a.bias ? a.bias[n] : 0.f;
                    const int64_t col = (int64_t)mrow * N + n;
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                        for (int r0 = 0; r0 < 16; r0 += 8) {
                            float rv[8];
#pragma unroll
                            for (int r = 0; r < 8; ++r) rv[r] = a.R[col + (int64_t)(tm * 32 + ((r0 + r) & 3) + 8 * ((r0 + r) >> 2)) * N];
#pragma unroll
                            for (int r = 0; r < 8; ++r) {
                                float v = acc[tm][tn][r0 + r] + bv;
                                v = a.relu ? fmaxf(v, 0.f) : v;
                                a.Y[col + (int64_t)(tm * 32 + ((r0 + r) & 3) + 8 * ((r0 + r) >> 2)) * N] = v + rv[r];
                                acc[tm][tn][r0 + r] = 0.f;
                            }
                        }
                }
            } else {
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) {
                    const int n = n0 + wc * 64 + tn * 32 + li;
                    const bool nin = n < N;
                    const float bv = (nin && a.bias) ? a.bias[n] : 0.f;
                    const int64_t col = (int64_t)mrow * N + n;
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int dm = tm * 32 + (r & 3) + 8 * (r >> 2);
                            if (nin && mrow + dm < M) {
                                float v = acc[tm][tn][r] + bv;
                                if (a.relu) v = fmaxf(v, 0.f);
                                if (a.R) v += a.R[col + (int64_t)dm * N];
                                a.Y[col + (int64_t)dm * N] = v;
                            }
                            acc[tm][tn][r] = 0.f;
                        }
                }
            }
        }
        if (more) {
            if (XF) lin_store_tile_frag<BK>(Xs(cur ^ 1), tid, xv);
            else lin_store_tile<BK>(Xs(cur ^ 1), tid, xv);
            lin_store_tile<BK>(Ws(cur ^ 1), tid, wv);
        }
        __syncthreads();
        cur ^= 1;
        kt = kt1, n0 = n1;
    }
}

// ------------------------------------------------------------------ linear + residual + LayerNorm (N <= 128)
// y = LN(X W^T + b + R; g1, b1); if (c) y = LN(y + c; g2, b2)      -- out-proj + LN1 (+c_l, LN2), FFN2 + LN3.
// Transposed MFMA orientation: D[n][m] = W_tile[n][k] . X^T[k][m], so a lane owns ONE token (column
// lane&31) and a wave owns all 128 output columns of its 32 tokens (4 accumulators = 64 values per
// lane, the other 64 in lane^32).  LayerNorm is then register-local: 64-term sums plus one xor-32
// shuffle per statistic -- no LDS staging, no barrier, no per-row shuffle chains.
// Workgroup = 4 waves = 128 tokens; W (all N rows) and X slabs of 32 k stream through the same
// swizzled double-buffered LDS image as k_linear.
// FULL = N == 128, K a multiple of BK, every operand 16-byte aligned (host-checked): no guards anywhere.
// NT = 32-column tiles per token: 4 (N <= 128) or 8 (N <= 256: a lane then holds 128 of its token's values, the
// W slab is two 128-row tiles; two waves per SIMD).  The fragment-major operands (Rf, Yf, Xf) exist for NT = 4 only.
template <int BK, bool FULL, bool XF, int NT = 4>
__global__ void __launch_bounds__(256, (BK == 16 && FULL && NT == 4) ? 3 : 2) k_linear_ln(LinArgs a) {
    constexpr int NC = 32 * NT; // padded row width
    static_assert(NT == 4 || (NT == 8 && !XF), "fragment-major X is a 128-column layout");
    __shared__ __attribute__((aligned(16))) float sm[2 * (LIN_BM + NC) * BK + 6 * NC];
    float *vecs = sm + 2 * (LIN_BM + NC) * BK; // bias, g1, b1, c, g2, b2 (zero padded to NC)
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int m0 = blockIdx.x * LIN_BM;
    const int M = a.m_dev ? min(a.M, a.m_dev[0]) : a.M, N = a.N, K = a.K;
    if (m0 >= M) return;
    auto Xs = [&](int st) { return sm + st * (LIN_BM + NC) * BK; };
    auto Ws = [&](int st) { return sm + st * (LIN_BM + NC) * BK + LIN_BM * BK; };
    if (tid < NC) {
        const bool in = tid < N;
        vecs[0 * NC + tid] = (in && a.bias) ? a.bias[tid] : 0.f;
        vecs[1 * NC + tid] = in ? a.g1[tid] : 0.f;
        vecs[2 * NC + tid] = in ? a.b1[tid] : 0.f;
        vecs[3 * NC + tid] = (in && a.c) ? a.c[tid] : 0.f;
        vecs[4 * NC + tid] = (in && a.c) ? a.g2[tid] : 0.f;
        vecs[5 * NC + tid] = (in && a.c) ? a.b2[tid] : 0.f;
    }
    const int mt = m0 + wave * 32 + li; // this lane's token

    // accumulators start from the residual of this lane's token (C layout: register 4g+e of tile tn is
    // column tn*32 + 8g + 4lk + e); bias is added in the epilogue from LDS.  A fragment-major residual
    // Rf[((token/32 * 4 + tn) * 4 + g) * 64 + lane] (float4) makes every wave load one contiguous KiB;
    // the row-major fallback reads 16-byte pieces of 64 different rows per instruction.
    f32x16 acc[NT];
    const int mtile = (m0 >> 5) + wave;
    if (NT == 4 && a.Rf) { // 16 unconditional, perfectly coalesced loads
#pragma unroll
        for (int tn = 0; tn < NT; ++tn)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 t = reinterpret_cast<const float4 *>(a.Rf)[((size_t)(mtile * 4 + tn) * 4 + g) * 64 + lane];
                acc[tn][4 * g + 0] = t.x;
                acc[tn][4 * g + 1] = t.y;
                acc[tn][4 * g + 2] = t.z;
                acc[tn][4 * g + 3] = t.w;
            }
    } else if (FULL) { // full rows: unconditional loads from a clamped row (rows >= M are never stored)
        const float *rrow = a.R + (int64_t)min(mt, M - 1) * N;
#pragma unroll
        for (int tn = 0; tn < NT; ++tn)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 t = *reinterpret_cast<const float4 *>(rrow + tn * 32 + 8 * g + 4 * lk);
                acc[tn][4 * g + 0] = t.x;
                acc[tn][4 * g + 1] = t.y;
                acc[tn][4 * g + 2] = t.z;
                acc[tn][4 * g + 3] = t.w;
            }
    } else {
#pragma unroll
        for (int tn = 0; tn < NT; ++tn)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = tn * 32 + 8 * g + 4 * lk;
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if (mt < M && n < N) {
                    const float *p = a.R + (int64_t)mt * N + n;
                    t.x = p[0];
                    if (n + 1 < N) t.y = p[1];
                    if (n + 2 < N) t.z = p[2];
                    if (n + 3 < N) t.w = p[3];
                }
                acc[tn][4 * g + 0] = t.x;
                acc[tn][4 * g + 1] = t.y;
                acc[tn][4 * g + 2] = t.z;
                acc[tn][4 * g + 3] = t.w;
            }
    }

    float4 xv[BK / 8], wv[BK / 8], wv2[BK / 8];
    if (XF) lin_load_tile_frag<BK>(a.Xf, m0, 0, tid, xv);
    else lin_load_tile<FULL, BK>(a.X, M, K, m0, 0, tid, xv);
    lin_load_tile<FULL, BK>(a.W, N, K, 0, 0, tid, wv);
    if (NT == 8) lin_load_tile<FULL, BK>(a.W, N, K, LIN_BM, 0, tid, wv2);
    if (XF) lin_store_tile_frag<BK>(Xs(0), tid, xv);
    else lin_store_tile<BK>(Xs(0), tid, xv);
    lin_store_tile<BK>(Ws(0), tid, wv);
    if (NT == 8) lin_store_tile<BK>(Ws(0) + LIN_BM * BK, tid, wv2);
    __syncthreads();
    const int nkt = (K + BK - 1) / BK;
    const int sw = lin_swz<BK>(li);
    int cur = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) {
            if (XF) lin_load_tile_frag<BK>(a.Xf, m0, (kt + 1) * BK, tid, xv);
            else lin_load_tile<FULL, BK>(a.X, M, K, m0, (kt + 1) * BK, tid, xv);
            lin_load_tile<FULL, BK>(a.W, N, K, 0, (kt + 1) * BK, tid, wv);
            if (NT == 8) lin_load_tile<FULL, BK>(a.W, N, K, LIN_BM, (kt + 1) * BK, tid, wv2);
        }
        const float *xb = Xs(cur) + (wave * 32 + li) * BK;
        const float *wa = Ws(cur) + li * BK;
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            const int off = ((2 * q + lk) ^ sw) << 2;
            const float4 x4 = *reinterpret_cast<const float4 *>(xb + off);
            float4 wq[NT];
#pragma unroll
            for (int tn = 0; tn < NT; ++tn) wq[tn] = *reinterpret_cast<const float4 *>(wa + tn * 32 * BK + off);
#define LN_STEP(E)                                                                                        \
    _Pragma("unroll") for (int tn = 0; tn < NT; ++tn)                                                     \
        acc[tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[tn].E, x4.E, acc[tn], 0, 0, 0);
            LN_STEP(x) LN_STEP(y) LN_STEP(z) LN_STEP(w)
#undef LN_STEP
        }
        if (kt + 1 < nkt) {
            if (XF) lin_store_tile_frag<BK>(Xs(cur ^ 1), tid, xv);
            else lin_store_tile<BK>(Xs(cur ^ 1), tid, xv);
            lin_store_tile<BK>(Ws(cur ^ 1), tid, wv);
            if (NT == 8) lin_store_tile<BK>(Ws(cur ^ 1) + LIN_BM * BK, tid, wv2);
        }
        __syncthreads();
        cur ^= 1;
    }
    // ---- register-local LayerNorm of this lane's token (half of its N values here, half in lane^32)
    const float invn = 1.0f / (float)N;
    float sum = 0.f;
#pragma unroll
    for (int tn = 0; tn < NT; ++tn)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = tn * 32 + 8 * g + 4 * lk;
            const float4 bb = *reinterpret_cast<const float4 *>(vecs + n);
            float z0 = acc[tn][4 * g + 0] + bb.x;
            float z1 = acc[tn][4 * g + 1] + bb.y;
            float z2 = acc[tn][4 * g + 2] + bb.z;
            float z3 = acc[tn][4 * g + 3] + bb.w;
            if (n >= N) z0 = 0.f;
            if (n + 1 >= N) z1 = 0.f;
            if (n + 2 >= N) z2 = 0.f;
            if (n + 3 >= N) z3 = 0.f;
            acc[tn][4 * g + 0] = z0;
            acc[tn][4 * g + 1] = z1;
            acc[tn][4 * g + 2] = z2;
            acc[tn][4 * g + 3] = z3;
            sum += (z0 + z1) + (z2 + z3);
        }
    auto stats = [&](float s1, float &mu, float &rstd) {
        mu = lanes_sum<32>(s1) * invn;
        float q = 0.f;
#pragma unroll
        for (int tn = 0; tn < NT; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = tn * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                const float dlt = (n < N) ? acc[tn][r] - mu : 0.f;
                q += dlt * dlt;
            }
        rstd = 1.0f / sqrtf(lanes_sum<32>(q) * invn + 1e-5f);
    };
    float mu, rstd;
    stats(sum, mu, rstd);
    const bool two = a.c != nullptr;
    sum = 0.f;
#pragma unroll
    for (int tn = 0; tn < NT; ++tn)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = tn * 32 + 8 * g + 4 * lk;
            const float4 gg = *reinterpret_cast<const float4 *>(vecs + 1 * NC + n);
            const float4 be = *reinterpret_cast<const float4 *>(vecs + 2 * NC + n);
            const float4 cc = *reinterpret_cast<const float4 *>(vecs + 3 * NC + n);
            float y0 = (acc[tn][4 * g + 0] - mu) * rstd * gg.x + be.x + cc.x;
            float y1 = (acc[tn][4 * g + 1] - mu) * rstd * gg.y + be.y + cc.y;
            float y2 = (acc[tn][4 * g + 2] - mu) * rstd * gg.z + be.z + cc.z;
            float y3 = (acc[tn][4 * g + 3] - mu) * rstd * gg.w + be.w + cc.w;
            if (n >= N) y0 = 0.f;
            if (n + 1 >= N) y1 = 0.f;
            if (n + 2 >= N) y2 = 0.f;
            if (n + 3 >= N) y3 = 0.f;
            acc[tn][4 * g + 0] = y0;
            acc[tn][4 * g + 1] = y1;
            acc[tn][4 * g + 2] = y2;
            acc[tn][4 * g + 3] = y3;
            sum += (y0 + y1) + (y2 + y3);
        }
    if (two) {
        stats(sum, mu, rstd);
#pragma unroll
        for (int tn = 0; tn < NT; ++tn)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = tn * 32 + 8 * g + 4 * lk;
                const float4 gg = *reinterpret_cast<const float4 *>(vecs + 4 * NC + n);
                const float4 be = *reinterpret_cast<const float4 *>(vecs + 5 * NC + n);
                acc[tn][4 * g + 0] = (acc[tn][4 * g + 0] - mu) * rstd * gg.x + be.x;
                acc[tn][4 * g + 1] = (acc[tn][4 * g + 1] - mu) * rstd * gg.y + be.y;
                acc[tn][4 * g + 2] = (acc[tn][4 * g + 2] - mu) * rstd * gg.z + be.z;
                acc[tn][4 * g + 3] = (acc[tn][4 * g + 3] - mu) * rstd * gg.w + be.w;
            }
    }
    if (NT == 4 && a.Yf) { // fragment-major copy for the next LN-GEMM's residual (rows/cols beyond M/N are zeros)
#pragma unroll
        for (int tn = 0; tn < NT; ++tn)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                reinterpret_cast<float4 *>(a.Yf)[((size_t)(mtile * 4 + tn) * 4 + g) * 64 + lane] =
                    make_float4(acc[tn][4 * g], acc[tn][4 * g + 1], acc[tn][4 * g + 2], acc[tn][4 * g + 3]);
    }
    if (FULL && a.Y && mt < M) {
        float *Yr = a.Y + (int64_t)mt * N;
#pragma unroll
        for (int tn = 0; tn < NT; ++tn)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4 *>(Yr + tn * 32 + 8 * g + 4 * lk) =
                    make_float4(acc[tn][4 * g], acc[tn][4 * g + 1], acc[tn][4 * g + 2], acc[tn][4 * g + 3]);
    } else if (!FULL && a.Y && mt < M) {
        float *Yr = a.Y + (int64_t)mt * N;
#pragma unroll
        for (int tn = 0; tn < NT; ++tn)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = tn * 32 + 8 * g + 4 * lk;
                if (n >= N) continue;
                if (false)
                    *reinterpret_cast<float4 *>(Yr + n) = make_float4(acc[tn][4 * g], acc[tn][4 * g + 1], acc[tn][4 * g + 2], acc[tn][4 * g + 3]);
                else {
                    Yr[n] = acc[tn][4 * g];
                    if (n + 1 < N) Yr[n + 1] = acc[tn][4 * g + 1];
                    if (n + 2 < N) Yr[n + 2] = acc[tn][4 * g + 2];
                    if (n + 3 < N) Yr[n + 3] = acc[tn][4 * g + 3];
                }
            }
    }
}

// ------------------------------------------------------------------ fused post-attention block (d = 128, F = 256)
//   y  = LN2(LN1(x + ao W_o^T + b_o) + c_l)            (OUT: self-attention out-projection + norm1, constant
//                                                        cross-attention + norm2)
//   x' = LN3(y + relu(y W1^T + b1) W2^T + b2)           nn.TransformerDecoderLayer._ff_block + norm3
//   qkv' = x' W_in'^T + b_in'                            (QKV: the NEXT layer's in-projection)
// One kernel, activations chained through registers.  In the transposed MFMA orientation D[n][token] a wave owns
// 32 tokens and a lane one token, and the C layout of an accumulator tile is a valid B-operand layout (register
// 4g+e of tile tn <-> k-pair {32tn+8g+e, 32tn+8g+4+e}).  So: the out-projection accumulates onto the residual x
// (fragment-major, contiguous KiB per wave) with the attention output as B operand straight from its
// fragment-major image; LayerNorm is register-local; the y tile (4 accumulator tiles) is GEMM 1's B operand; the
// hidden activation h^T (8 tiles) is GEMM 2's B operand and GEMM 2 accumulates onto y in place (the residual);
// after LN3 the x' tile is the B operand of the next layer's QKV projection (2 passes of 6 output tiles in the
// registers h occupied).  y, h and x' never travel through memory; HBM traffic per token is the attention output
// and the residual in (1 KiB), x' and qkv' out (2 KiB).  Workgroup = 4 waves = 128 tokens; all weight k-slabs
// of 16 stream through one double-buffered LDS image (8 + 8 + 16 + 16 pipeline steps).
struct BlockArgs {
    // OUT phase (null Wo: y is read from Yf instead)
    const float *Af;  // attention output, fragment-major
    const float *Rf;  // residual x, fragment-major
    const float *Wo, *bo, *g1, *b1n, *c, *g2, *b2n;
    const float *Yf;  // y, fragment-major (only when the OUT phase is not fused)
    const float *W1, *b1, *W2, *b2, *g, *b;
    float *Xf; // fragment-major output (may be null)
    float *Y;  // row-major output [M][128] (may be null)
    int M;
    const int32_t *m_dev;
    const float *Win, *bin; // QKV tail: [384][128], [384]
    float *QKV;             // row-major [M][384]
    int qkv_n0, qkv_nt1;    // QKV tail: first output column and tiles of the second pass: (0, 6) = all of q | k | v;
                            // (128, 2) = k | v only (the last layer's queries are needed for one row per sequence)
};

template <bool OUT, bool QKV>
__global__ void __launch_bounds__(256, 2) k_block(BlockArgs a) {
    constexpr int D = 128, F = 256, BK = 16;
    constexpr int V_B1 = 0, V_B2 = F, V_G = F + D, V_B = F + 2 * D, V_BIN = F + 3 * D, V_O = F + 3 * D + (QKV ? 3 * D : 0);
    __shared__ __attribute__((aligned(16))) float sm[2 * F * BK + V_O + (OUT ? 6 * D : 0)];
    float *vecs = sm + 2 * F * BK; // b1[256], b2, g, b [, b_in[384]] [, b_o, g1, b1n, c, g2, b2n]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int m0 = blockIdx.x * 128;
    const int M = a.m_dev ? min(a.M, a.m_dev[0]) : a.M;
    if (m0 >= M) return;
    vecs[V_B1 + tid] = a.b1[tid];
    if (tid < D) {
        vecs[V_B2 + tid] = a.b2[tid];
        vecs[V_G + tid] = a.g[tid];
        vecs[V_B + tid] = a.b[tid];
    }
    if (QKV) {
        vecs[V_BIN + tid] = a.bin[tid];
        if (tid < D) vecs[V_BIN + 256 + tid] = a.bin[256 + tid];
    }
    if (OUT && tid < D) {
        vecs[V_O + 0 * D + tid] = a.bo ? a.bo[tid] : 0.f;
        vecs[V_O + 1 * D + tid] = a.g1[tid];
        vecs[V_O + 2 * D + tid] = a.b1n[tid];
        vecs[V_O + 3 * D + tid] = a.c ? a.c[tid] : 0.f;
        vecs[V_O + 4 * D + tid] = a.c ? a.g2[tid] : 0.f;
        vecs[V_O + 5 * D + tid] = a.c ? a.b2n[tid] : 0.f;
    }
    const int mtile = (m0 >> 5) + wave;
    const int mt = m0 + wave * 32 + li; // this lane's token
    const size_t fbase = (size_t)mtile * 16 * 64 + lane; // + (tn*4+g)*64: this lane's float4 of a fragment-major image

    // slab loaders, step t: [-8, 0) W_o[:, 16(t+8) ..] (128 rows) | [0, 8) W1 (256 rows) | [8, 24) W2[:, 16(t-8) ..]
    // (128 rows) | [24, 40) W_in[192p .. 192p+191][16s ..], p = (t-24)/8, s = (t-24)%8 (192 rows)
    float4 wv[4];
    auto load_slab = [&](int t) {
        if (OUT && t < 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = tid + i * 256;
                const float *p = a.Wo + (size_t)(idx >> 2) * D + (t + 8) * BK + (idx & 3) * 4;
                wv[i] = make_float4(p[0], p[1], p[2], p[3]);
            }
        } else if (QKV && t >= 24) {
            const int pp = (t - 24) >> 3, ss = (t - 24) & 7;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int idx = tid + i * 256;
                const int wrow = min(a.qkv_n0 + 192 * pp + (idx >> 2), 3 * D - 1); // rows past the end: unused tiles
                const float *p = a.Win + (size_t)wrow * D + ss * BK + (idx & 3) * 4;
                wv[i] = make_float4(p[0], p[1], p[2], p[3]);
            }
        } else if (t < 8) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = tid + i * 256;
                const float *p = a.W1 + (size_t)(idx >> 2) * D + t * BK + (idx & 3) * 4;
                wv[i] = make_float4(p[0], p[1], p[2], p[3]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = tid + i * 256;
                const float *p = a.W2 + (size_t)(idx >> 2) * F + (t - 8) * BK + (idx & 3) * 4;
                wv[i] = make_float4(p[0], p[1], p[2], p[3]);
            }
        }
    };
    auto store_slab = [&](int t, float *S) {
        const int n = t < 0 ? 2 : (t < 8 ? 4 : (t < 24 ? 2 : 3));
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < n) {
                const int idx = tid + i * 256;
                const int r = idx >> 2, c = idx & 3;
                *reinterpret_cast<float4 *>(S + r * BK + ((c ^ lin_swz<BK>(r)) << 2)) = wv[i];
            }
    };
    const int sw = lin_swz<BK>(li);
    const float invn = 1.0f / (float)D;
    int cur = 0;
    f32x16 acc[4]; // x + out-projection -> y -> y + FFN -> x' (this lane's token, 64 of its 128 columns)

    float4 yv[2], yn[2]; // B fragments streamed from a fragment-major image (attention output / y)
    if (OUT) {
        // ---- accumulators start from the residual x; out-projection with the attention output as B operand
        const float4 *rfrag = reinterpret_cast<const float4 *>(a.Rf) + fbase;
        const float4 *afrag = reinterpret_cast<const float4 *>(a.Af) + fbase;
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 t4 = rfrag[(tn * 4 + g) * 64];
                acc[tn][4 * g + 0] = t4.x, acc[tn][4 * g + 1] = t4.y, acc[tn][4 * g + 2] = t4.z, acc[tn][4 * g + 3] = t4.w;
            }
        load_slab(-8);
        yv[0] = afrag[0];
        yv[1] = afrag[64];
        store_slab(-8, sm);
        __syncthreads();
#pragma unroll
        for (int t = -8; t < 0; ++t) {
            load_slab(t + 1); // t + 1 == 0 is W1's first slab
            if (t + 1 < 0) {
                yn[0] = afrag[(2 * (t + 9)) * 64];
                yn[1] = afrag[(2 * (t + 9) + 1) * 64];
            }
            const float *wa = sm + cur * F * BK + li * BK;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int off = ((2 * q + lk) ^ sw) << 2;
                const float4 x4 = yv[q];
                float4 w[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) w[i] = *reinterpret_cast<const float4 *>(wa + i * 32 * BK + off);
#define OUT_STEP(E)                                                                    \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0].E, x4.E, acc[0], 0, 0, 0);      \
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[1].E, x4.E, acc[1], 0, 0, 0);      \
    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2].E, x4.E, acc[2], 0, 0, 0);      \
    acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[3].E, x4.E, acc[3], 0, 0, 0);
                OUT_STEP(x) OUT_STEP(y) OUT_STEP(z) OUT_STEP(w)
#undef OUT_STEP
            }
            store_slab(t + 1, sm + (cur ^ 1) * F * BK);
            __syncthreads();
            cur ^= 1;
            yv[0] = yn[0];
            yv[1] = yn[1];
        }
        // ---- + b_o, LN1 (+ c, LN2): register-local (64 of the 128 values here, 64 in lane^32)
        auto layer_norm = [&](int vb, int vg, int vbeta, int vadd) {
            float sum = 0.f;
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (vb >= 0) {
                        const float4 bb = *reinterpret_cast<const float4 *>(vecs + vb + tn * 32 + 8 * g + 4 * lk);
                        acc[tn][4 * g + 0] += bb.x, acc[tn][4 * g + 1] += bb.y, acc[tn][4 * g + 2] += bb.z, acc[tn][4 * g + 3] += bb.w;
                    }
                    sum += (acc[tn][4 * g + 0] + acc[tn][4 * g + 1]) + (acc[tn][4 * g + 2] + acc[tn][4 * g + 3]);
                }
            const float mu = lanes_sum<32>(sum) * invn;
            float qs = 0.f;
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float dlt = acc[tn][r] - mu;
                    qs += dlt * dlt;
                }
            const float rstd = 1.0f / sqrtf(lanes_sum<32>(qs) * invn + 1e-5f);
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = tn * 32 + 8 * g + 4 * lk;
                    const float4 gg = *reinterpret_cast<const float4 *>(vecs + vg + n);
                    const float4 be = *reinterpret_cast<const float4 *>(vecs + vbeta + n);
                    float4 ad = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (vadd >= 0) ad = *reinterpret_cast<const float4 *>(vecs + vadd + n);
                    acc[tn][4 * g + 0] = (acc[tn][4 * g + 0] - mu) * rstd * gg.x + be.x + ad.x;
                    acc[tn][4 * g + 1] = (acc[tn][4 * g + 1] - mu) * rstd * gg.y + be.y + ad.y;
                    acc[tn][4 * g + 2] = (acc[tn][4 * g + 2] - mu) * rstd * gg.z + be.z + ad.z;
                    acc[tn][4 * g + 3] = (acc[tn][4 * g + 3] - mu) * rstd * gg.w + be.w + ad.w;
                }
        };
        layer_norm(V_O + 0 * D, V_O + 1 * D, V_O + 2 * D, V_O + 3 * D);
        if (a.c) layer_norm(-1, V_O + 4 * D, V_O + 5 * D, -1);
    } else {
        load_slab(0);
        const float4 *yfrag = reinterpret_cast<const float4 *>(a.Yf) + fbase;
        yv[0] = yfrag[0];
        yv[1] = yfrag[64];
        store_slab(0, sm);
        __syncthreads();
    }

    f32x16 h[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) h[i][r] = 0.f;
    // ---- GEMM 1: h^T = W1 y^T (B operand: the y tile in registers, or y fragments streamed from Yf)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        load_slab(t + 1);
        if (!OUT && t + 1 < 8) {
            const float4 *yfrag = reinterpret_cast<const float4 *>(a.Yf) + fbase;
            yn[0] = yfrag[(2 * (t + 1)) * 64];
            yn[1] = yfrag[(2 * (t + 1) + 1) * 64];
        }
        const float *wa = sm + cur * F * BK + li * BK;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int off = ((2 * q + lk) ^ sw) << 2;
            const int tn = t >> 1, g = 2 * (t & 1) + q;
            const float4 x4 = OUT ? make_float4(acc[tn][4 * g + 0], acc[tn][4 * g + 1], acc[tn][4 * g + 2], acc[tn][4 * g + 3]) : yv[q];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float4 w[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) w[i] = *reinterpret_cast<const float4 *>(wa + (half * 4 + i) * 32 * BK + off);
#define FFN_STEP(E)                                                                                         \
    h[half * 4 + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0].E, x4.E, h[half * 4 + 0], 0, 0, 0);         \
    h[half * 4 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[1].E, x4.E, h[half * 4 + 1], 0, 0, 0);         \
    h[half * 4 + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2].E, x4.E, h[half * 4 + 2], 0, 0, 0);         \
    h[half * 4 + 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[3].E, x4.E, h[half * 4 + 3], 0, 0, 0);
                FFN_STEP(x) FFN_STEP(y) FFN_STEP(z) FFN_STEP(w)
#undef FFN_STEP
            }
        }
        store_slab(t + 1, sm + (cur ^ 1) * F * BK);
        __syncthreads();
        cur ^= 1;
        if (!OUT) {
            yv[0] = yn[0];
            yv[1] = yn[1];
        }
    }
    // ---- h = relu(h + b1) in place
#pragma unroll
    for (int ft = 0; ft < 8; ++ft)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bb = *reinterpret_cast<const float4 *>(vecs + V_B1 + ft * 32 + 8 * g + 4 * lk);
            h[ft][4 * g + 0] = fmaxf(h[ft][4 * g + 0] + bb.x, 0.f);
            h[ft][4 * g + 1] = fmaxf(h[ft][4 * g + 1] + bb.y, 0.f);
            h[ft][4 * g + 2] = fmaxf(h[ft][4 * g + 2] + bb.z, 0.f);
            h[ft][4 * g + 3] = fmaxf(h[ft][4 * g + 3] + bb.w, 0.f);
        }
    // ---- GEMM 2 accumulates onto the residual y: already in the accumulators (OUT) or read from Yf
    if (!OUT) {
        const float4 *yfrag = reinterpret_cast<const float4 *>(a.Yf) + fbase;
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 t4 = yfrag[(tn * 4 + g) * 64];
                acc[tn][4 * g + 0] = t4.x, acc[tn][4 * g + 1] = t4.y, acc[tn][4 * g + 2] = t4.z, acc[tn][4 * g + 3] = t4.w;
            }
    }
    // ---- GEMM 2: out^T += W2 h^T ; slab s covers f = 16s .. 16s+15 = tile ft = s/2, groups g = 2(s&1) + q
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
        if (s2 + 1 < 16 || QKV) load_slab(8 + s2 + 1);
        const float *wa = sm + cur * F * BK + li * BK;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int off = ((2 * q + lk) ^ sw) << 2;
            const int ft = s2 >> 1, g = 2 * (s2 & 1) + q;
            float4 w[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = *reinterpret_cast<const float4 *>(wa + i * 32 * BK + off);
#define FFN_STEP2(E, R)                                                                                \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0].E, h[ft][4 * g + R], acc[0], 0, 0, 0);          \
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[1].E, h[ft][4 * g + R], acc[1], 0, 0, 0);          \
    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2].E, h[ft][4 * g + R], acc[2], 0, 0, 0);          \
    acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[3].E, h[ft][4 * g + R], acc[3], 0, 0, 0);
            FFN_STEP2(x, 0) FFN_STEP2(y, 1) FFN_STEP2(z, 2) FFN_STEP2(w, 3)
#undef FFN_STEP2
        }
        if (s2 + 1 < 16 || QKV) store_slab(8 + s2 + 1, sm + (cur ^ 1) * F * BK);
        __syncthreads();
        cur ^= 1;
    }
    // ---- + b2, register-local LayerNorm (64 of the 128 values here, 64 in lane^32), stores
    float sum = 0.f;
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bb = *reinterpret_cast<const float4 *>(vecs + V_B2 + tn * 32 + 8 * g + 4 * lk);
            acc[tn][4 * g + 0] += bb.x;
            acc[tn][4 * g + 1] += bb.y;
            acc[tn][4 * g + 2] += bb.z;
            acc[tn][4 * g + 3] += bb.w;
            sum += (acc[tn][4 * g + 0] + acc[tn][4 * g + 1]) + (acc[tn][4 * g + 2] + acc[tn][4 * g + 3]);
        }
    const float mu = lanes_sum<32>(sum) * invn;
    float qs = 0.f;
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float dlt = acc[tn][r] - mu;
            qs += dlt * dlt;
        }
    const float rstd = 1.0f / sqrtf(lanes_sum<32>(qs) * invn + 1e-5f);
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = tn * 32 + 8 * g + 4 * lk;
            const float4 gg = *reinterpret_cast<const float4 *>(vecs + V_G + n);
            const float4 be = *reinterpret_cast<const float4 *>(vecs + V_B + n);
            const float4 o = make_float4((acc[tn][4 * g + 0] - mu) * rstd * gg.x + be.x, (acc[tn][4 * g + 1] - mu) * rstd * gg.y + be.y,
                                         (acc[tn][4 * g + 2] - mu) * rstd * gg.z + be.z, (acc[tn][4 * g + 3] - mu) * rstd * gg.w + be.w);
            if (a.Xf) reinterpret_cast<float4 *>(a.Xf)[((size_t)(mtile * 4 + tn) * 4 + g) * 64 + lane] = o;
            if (a.Y && mt < M) *reinterpret_cast<float4 *>(a.Y + (int64_t)mt * D + n) = o;
            if (QKV) acc[tn][4 * g + 0] = o.x, acc[tn][4 * g + 1] = o.y, acc[tn][4 * g + 2] = o.z, acc[tn][4 * g + 3] = o.w;
        }
    if (!QKV) return;
    // ---- QKV tail: qkv^T = W_in x^T, 2 passes x (6, qkv_nt1) output tiles from column qkv_n0, k-slab s <-> x tile
    //      tn = s/2, groups g = 2(s&1) + q
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
        const int nt = pp == 0 ? 6 : a.qkv_nt1; // workgroup-uniform
        f32x16 qa[6];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) qa[i][r] = 0.f;
#pragma unroll
        for (int ss = 0; ss < 8; ++ss) {
            const int t = 24 + pp * 8 + ss;
            if (t + 1 < 40) load_slab(t + 1);
            const float *wa = sm + cur * F * BK + li * BK;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int off = ((2 * q + lk) ^ sw) << 2;
                const int tn = ss >> 1, g = 2 * (ss & 1) + q;
                float4 w[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) w[i] = *reinterpret_cast<const float4 *>(wa + i * 32 * BK + off);
#define QKV_STEP(E, R)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 6; ++i) if (i < 2 || i < nt)                                          \
        qa[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[i].E, acc[tn][4 * g + R], qa[i], 0, 0, 0);
                QKV_STEP(x, 0) QKV_STEP(y, 1) QKV_STEP(z, 2) QKV_STEP(w, 3)
#undef QKV_STEP
            }
            if (t + 1 < 40) store_slab(t + 1, sm + (cur ^ 1) * F * BK);
            __syncthreads();
            cur ^= 1;
        }
        if (mt < M) {
            const int c0 = a.qkv_n0 + pp * 192;
            float *qrow = a.QKV + (int64_t)mt * (3 * D) + c0 + 4 * lk;
#pragma unroll
            for (int i = 0; i < 6; ++i)
                if (i < nt) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 bb = *reinterpret_cast<const float4 *>(vecs + V_BIN + c0 + i * 32 + 8 * g + 4 * lk);
                        *reinterpret_cast<float4 *>(qrow + i * 32 + 8 * g) =
                            make_float4(qa[i][4 * g + 0] + bb.x, qa[i][4 * g + 1] + bb.y, qa[i][4 * g + 2] + bb.z, qa[i][4 * g + 3] + bb.w);
                    }
                }
        }
    }
}

// ------------------------------------------------------------------ fused post-attention block on split-bf16 MFMAs
// k_block<true, true> with every GEMM on v_mfma_f32_32x32x16_bf16: a float32 value is the exact sum of three bf16
// values (hi + mid + lo: 8 + 8 + 8 significant bits), so the six products hh, hm, mh, hl, lh, mm reproduce a float32
// product to ~2^-24 relative (the three dropped ones are below 2^-32) at 6/16 of the fp32-MFMA cost.  Round 2 tried
// this on k_linear's tile and pipeline and found it no faster (a 16-k slab is 768 MFMA cycles: one slab of register
// prefetch and a barrier per slab no longer cover the load latency) -- this kernel is built around it instead:
//   * the weights are split ONCE at finalisation (k_pack_x6) into the exact A-fragment order the kernel consumes, one
//     24 KiB block per pipeline step (4 output tiles x 2 k-steps of 16 x 3 planes x 1 KiB), so a step's weights are 24
//     contiguous KiB pieces that go HBM/L2 -> LDS by LDS-DMA (no VGPR staging, no ds_write) into a ring of 3 slots
//     with ONE raw s_barrier per step placed in the middle of the step's MFMAs (the sweep kernels' protocol);
//   * a pipeline step is 32 k (one accumulator tile of the producing GEMM) x 128 output rows = 48 MFMAs = 1536 cycles;
//     32 steps per 128-token workgroup: out-projection 4, feed-forward 16 (per hidden tile of 32 units one FFN-1 step
//     over the whole K and one FFN-2 step), the next layer's QKV 12 (three passes of 128 columns; a k | v-only tail skips the first);
//   * activations still chain through registers in the transposed orientation (a lane owns a token): an accumulator
//     tile IS a B operand -- registers 8s .. 8s+7 of a tile, converted pairwise to bf16, are the B fragment of k-step s,
//     the k order inside a step being 16s + 8(j>>2) + 4h + (j&3) (cdna_hip_programming.md section 3), which is the
//     order k_pack_x6 packs the weights in; the split of a tile into its three planes costs ~5.5 vector instructions
//     per value, once per tile and pipeline step (1.8 per MFMA);
//   * fragment reads are inline-asm ds_read_b128 three fragments ahead through a ring of four registers sets with
//     hand-counted lgkmcnt (a compiler-visible LDS read behind an LDS-DMA issue gets a vmcnt(0) in front).
// Accuracy: the bf16 matrix pipe accumulates without round-to-nearest; round 2 measured the decoder rows' error at
// ~1.5-3x the fp32-MFMA kernels' (still 1e-5-scale against the 1e-3 relative bar of the north star).
typedef __attribute__((ext_vector_type(8))) __bf16 x6_bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int x6_u32x4;
#define X6_STEP_B 24576
#define X6_NSTEP 32
#define X6_LAYER_BYTES (X6_NSTEP * X6_STEP_B)
// NT = d / 32 accumulator tiles per token (4: d = 128; 8: d = 256, round 4), HT = NT / 4 "halves": a step is always 24 KB =
// 8 groups x 3 planes, so at NT = 8 a k tile's eight output tiles take two four-tile steps and a one-tile product over
// K = 256 takes two one-tile steps (k tiles 0-3, 4-7).  Steps per layer: NT HT (out-projection) + 8 (2 HT) (feed-forward:
// per hidden tile HT FFN-1 steps, then HT FFN-2 steps) + 3 NT HT (q | k | v): 32 at NT = 4, 96 at NT = 8.
__host__ __device__ constexpr int x6_npre(int NT) { return NT * (NT / 4) + 16 * (NT / 4); } // steps in front of the q | k | v region
__host__ __device__ constexpr int x6_nstep(int NT) { return x6_npre(NT) + 3 * NT * (NT / 4); }
// NPL = planes per float32 operand.  3 (round 3): bf16 planes h + m + l = the value exactly, six plane products per float32
// product.  2 (round 4, "h3"): FLOAT16 planes h = f16(x), l = f16(x - h) -- 22 of the 24 significand bits (f16 carries 11)
// while l is a normal float16, 2^-25 absolute below (see X6_H3_WSHIFT for what that means for the weights) --,
// three plane products (hh, hl, lh; what is dropped is 2^-22 relative: the float32 accumulation's own rounding class) on
// v_mfma_f32_32x32x16_f16, same shapes and lane maps: HALF the matrix instructions, 2/3 of the weight stream, a cheaper split
// (v_cvt_pk_f16_f32: 3 vector instructions per value instead of 5.5).  Float16's range is the price: |operand| < 65504 (weights
// are O(0.1), activations behind a LayerNorm O(10), embedded tokens |E| sqrt(d) + 1); tiny values fall into f16 subnormals,
// whose ABSOLUTE spacing 2^-24 is what matters for a sum of products.  A step is 8 groups x NPL planes = 8 NPL KB.
__host__ __device__ constexpr int x6_step_b(int NPL) { return 8 * NPL * 1024; }
// Round 5: the float16-plane WEIGHT operands are packed times 2^8.  Weights are O(0.05): their low plane l = f16(w - f16(w)) is
// a float16 SUBNORMAL (|l| <= 2^-16 < 2^-14), whose absolute spacing 2^-24 left only ~19-20 significand bits on typical weights
// ("22 bits" held for |x| >= 2^-3 only).  Scaled by a power of two -- exact, commutes with every rounding -- the absolute floor
// of a weight's two-plane value is 2^-33 instead of 2^-25; the accumulators then hold 2^8 times the product and every epilogue
// folds the 2^-8 into the multiply-add that adds its bias (one fma instead of one add: no extra instruction).  Activations
// (LayerNorm outputs, hidden units, attention outputs: O(1)) keep their scale: 2^-25 absolute on an O(1) operand is 2^-25
// relative to the sum it enters.  Bf16 planes (IRS_GEMM_X6) carry float32's exponent range and are not scaled.
#define X6_H3_WSHIFT 8
__host__ __device__ constexpr float x6_wscale(int NPL) { return NPL == 2 ? (float)(1 << X6_H3_WSHIFT) : 1.0f; }
__host__ __device__ constexpr size_t x6_layer_bytes(int NT, int NPL = 3) { return (size_t)x6_nstep(NT) * x6_step_b(NPL); }
// ring slots (X6_NSLOT2 for float16 planes: their steps are 16 KB, so four fit twice per CU -- two steps of DMA in flight
// behind the one being multiplied)
#ifndef X6_NSLOT2
#define X6_NSLOT2 4
#endif
__host__ __device__ constexpr int x6_nslot(int NPL) { return NPL == 2 ? X6_NSLOT2 : 3; }
__host__ __device__ constexpr int x6_lds_bytes(int NT, int NPL = 3) { return x6_nslot(NPL) * x6_step_b(NPL) + (256 + 12 * 32 * NT) * 4; }
typedef __attribute__((ext_vector_type(8))) _Float16 x6_f16x8;
#ifndef X6_RESID_LATE
#define X6_RESID_LATE 1
#endif
#ifndef X6_RING4
#define X6_RING4 4 // fragment register sets of k_block_x6 at d = 128, float16 planes (lab: 8 measured no faster, 16 more registers)
#endif
#ifndef X6_SPLIT_ACC4
#define X6_SPLIT_ACC4 0 // one-tile steps at d = 128: 1 = h.h products on their own accumulator instead of a second read of W_h (lab: no faster, 16 more registers)
#endif
#ifndef X6_NW
#define X6_NW 4 // waves per workgroup of k_block_x6 (tools/x6_lab measures both)
#endif

// one layer's weights -> the kernel's step stream.  Thread = one 16-byte fragment piece (8 bf16 of one lane).
template <int NT = 4, int NPL = 3>
__global__ void __launch_bounds__(256) k_pack_x6(const float *__restrict__ Wo, const float *__restrict__ W1,
                                                 const float *__restrict__ W2, const float *__restrict__ Win,
                                                 uint4 *__restrict__ out) {
    constexpr int D = 32 * NT, HT = NT / 4, NPRE = x6_npre(NT), NOUT = NT * HT, NP = 8 * NPL;
    const int gid = blockIdx.x * 256 + threadIdx.x; // < steps * NP pieces * 64 lanes
    if (gid >= x6_nstep(NT) * NP * 64) return;
    const int lane = gid & 63, piece = (gid >> 6) % NP, step = gid / (NP * 64);
    // piece = 3 g + plane in the kernel's consumption order.  "Four-tile" steps (out-projection, FFN-2): group g =
    // (k-step s = g >> 2, output tile nt = g & 3) of one 32-wide k tile and one output half oh (tiles 4 oh .. 4 oh + 3);
    // "one-tile" steps (FFN-1, QKV): ONE output tile over 128 of the K = d columns (k half kh), group g = (k tile 4 kh +
    // (g >> 1), k-step s = g & 1).
    const int p = piece % NPL, g = piece / NPL;
    const int r = lane & 31, hh = lane >> 5;
    const float *W;
    int ld, n, kb, s;
    if (step < NOUT) { // out-projection: k tile step / HT, output half step % HT
        const int t = step / HT, oh = step % HT;
        W = Wo, ld = D, s = g >> 2, n = 32 * (4 * oh + (g & 3)) + r, kb = 32 * t;
    } else if (step < NPRE) {
        const int ft = (step - NOUT) / (2 * HT), q = (step - NOUT) % (2 * HT);
        if (q < HT) W = W1, ld = D, s = g & 1, n = 32 * ft + r, kb = 32 * (4 * q + (g >> 1));            // FFN-1, hidden tile ft, k half q
        else W = W2, ld = 256, s = g >> 2, n = 32 * (4 * (q - HT) + (g & 3)) + r, kb = 32 * ft;          // FFN-2, k tile ft, output half q - HT
    } else { // q | k | v output tile (step - NPRE) / HT, k half (step - NPRE) % HT (one-tile)
        const int qt = (step - NPRE) / HT, kh = (step - NPRE) % HT;
        W = Win, ld = D, s = g & 1, n = 32 * qt + r, kb = 32 * (4 * kh + (g >> 1));
    }
    unsigned int hw[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = kb + 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3);
        float v = W ? W[(size_t)n * ld + k] : 0.f;
        unsigned int bits = 0;
        if constexpr (NPL == 3) {
#pragma unroll
            for (int q = 0; q <= p; ++q) { // plane q = bf16(v - the planes before it), round to nearest even
                unsigned int u = __float_as_uint(v);
                u += 0x7FFFu + ((u >> 16) & 1u);
                bits = u >> 16;
                v -= __uint_as_float(bits << 16);
            }
        } else { // float16 planes of 2^8 v: h = f16(v), l = f16(v - h), round to nearest even (the conversion instruction's mode)
            v *= x6_wscale(2);
            _Float16 hv = (_Float16)v;
            if (p == 1) hv = (_Float16)(v - (float)hv);
            bits = (unsigned int)__builtin_bit_cast(unsigned short, hv);
        }
        hw[j] = bits;
    }
    out[gid] = make_uint4(hw[0] | (hw[1] << 16), hw[2] | (hw[3] << 16), hw[4] | (hw[5] << 16), hw[6] | (hw[7] << 16));
}

struct BlockX6Args {
    const float *Af, *Rf;  // attention output and residual x, fragment-major
    const uint4 *Wx;       // this layer's step stream (k_pack_x6)
    const float *bo, *g1, *b1n, *c, *g2, *b2n, *b1, *b2, *g, *b, *bin;
    float *Xf;             // fragment-major x' (may be null)
    float *QKV;            // row-major [M][384]
    int M;
    const int32_t *m_dev;
    int qkv_pass0;         // (documentation only: the kernel's QP0 template parameter decides)
    // EMBED instantiation (layer 0's input: x = item_emb[seq] * sqrt(d) + pe[pos], then layer 0's q | k | v)
    const int64_t *seq;
    const float *E, *pe;
    const int32_t *tok_row;
    int L;
    float sqrtd;
    int64_t n_item;
    // != 0 (with the q tail, QP0 = 0): the V section of a q | k | v row is written as float16 PLANE PAIRS -- per
    // (token, head) [32 f16 h | 32 f16 l], h = f16(v), l = f16(v - h), the same 128 bytes as 32 floats -- the operand format
    // of k_attn16h; the q and k sections stay float32.  (The k | v-only tail that feeds the rows-only last layer keeps float32:
    // k_attn_row32 reads it.)
    int kv_planes;
    // SEQ instantiation (round 5, k_block_x6<.., SEQ = true>: the sequence-resident layer kernel).  A workgroup of eight waves owns
    // WHOLE sequences (tile t of the grid = 32 consecutive tokens of sequence tile_seq[t], its tile_idx[t]-th; k_plan_seq), computes
    // this layer's q | k | v from x itself, keeps K / V of one head at a time in LDS, runs the attention of its own queries and
    // goes on with the layer body: no q | k | v row ever reaches HBM.  Rf = x (fragment-major, TILE order), Af = scratch for the
    // attention output in the same order (written and read back by the same wave), Xf = x'.
    // ALL of layers 0 .. n_lay - 1 in ONE launch (stage B): x stays in the wave's registers from layer to layer, the weight ring
    // runs through the layer boundaries (one empty step per layer keeps the slot numbering: 33 = 0 mod 3 steps), the parameter
    // vectors of the next layer are re-staged into LDS at the boundary from a packed copy.  The last of them also writes the
    // k | v rows the rows-only last layer reads.
    const uint4 *Wbase;    // float16-plane stream of layer 0; layer l's stream = Wbase + l * wstride (x6_stream)
    long long wstride;     // in uint4
    int n_lay, nl_total;   // layers run here (n_layers - 1); streams in the arena (layer l's q | k | v sits in stream l - 1, layer 0's in stream nl_total - 1)
    const float *vecpack;  // [n_lay + 1][X6_SEQ_VECS]: every parameter vector of a layer in the kernel's LDS order (k_pack_seqvec)
    // Behind layer n_lay - 1 the kernel runs the FRONT of layer n_lay too (the model's last layer, of which the handlers consume one
    // row per sequence): q | k | v of every token, K / V images, and the attention of the ONE 16-query block that holds the consumed
    // token seq_qrow[b] -- its output row is gathered from the attention tiles by the host side; no k | v row reaches HBM.
    const int32_t *seq_qrow;
    const int32_t *tile_seq, *tile_qb, *seq_off, *seq_cnt, *seq_padq, *seq_row0, *n_wg_dev;
    const float *r_u;
    int mask_mode;
#ifdef X6_DUMP
    uint4 *dbg;            // (lab) the fragments workgroup 0 / wave 0 consumed, [step][fragment][lane]
#endif
#ifdef X6_STAMP
    unsigned long long *stamps; // (lab) per wave: total, DMA wait, barrier, DMA issue, steps (cycles of s_memtime)
#endif
};

// float32 accumulator tile -> the three bf16 planes of its k-step s (registers 8s .. 8s+7) as B fragments
__device__ __forceinline__ void x6_split(const f32x16 &t, int s, x6_f16x8 (&X)[2]) { // float16 planes (NPL = 2)
#ifdef X6_NO_SPLIT
    for (int p = 0; p < 2; ++p) X[p] = __builtin_bit_cast(x6_f16x8, make_float4(t[8 * s + p], t[8 * s + p + 1], t[8 * s + p + 2], t[8 * s + p + 3]));
    return;
#endif
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = t[8 * s + j];
        const _Float16 h = (_Float16)v;
        X[0][j] = h;
        X[1][j] = (_Float16)(v - (float)h);
    }
}
__device__ __forceinline__ f32x16 x6_mma(const x6_bf16x8 &w, const x6_bf16x8 &x, const f32x16 &c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 x6_mma(const x6_f16x8 &w, const x6_f16x8 &x, const f32x16 &c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(w, x, c, 0, 0, 0);
}
__device__ __forceinline__ void x6_split(const f32x16 &t, int s, x6_bf16x8 (&X)[3]) {
#ifdef X6_NO_SPLIT
#pragma unroll
    for (int p = 0; p < 3; ++p) X[p] = __builtin_bit_cast(x6_bf16x8, make_float4(t[8 * s + p], t[8 * s + p + 1], t[8 * s + p + 2], t[8 * s + p + 3]));
    return;
#endif
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = t[8 * s + j];
        const __bf16 h = (__bf16)v;
        const float r1 = v - (float)h;
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;
        X[0][j] = h;
        X[1][j] = m;
        X[2][j] = (__bf16)r2;
    }
}

template <int B_, int E_, class Fn>
__device__ __forceinline__ void x6_static_for(Fn &&fn) {
    if constexpr (B_ < E_) {
        fn(std::integral_constant<int, B_>{});
        x6_static_for<B_ + 1, E_>(fn);
    }
}
// (asm: a compiler-visible LDS read behind an LDS-DMA issue gets a vmcnt(0) in front; the offset is an immediate so that
// the 72 fragment addresses of the ring cost three base registers, not one register each)
template <int OFF>
__device__ __forceinline__ x6_u32x4 x6_rd(unsigned int base) {
    x6_u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base), "i"(OFF));
    return v;
}

// a 16-byte global load the compiler does not track (its wait is a counted vmcnt in inline asm: see RESID_LATE)
template <int OFF>
__device__ __forceinline__ x6_u32x4 x6_gld(const void *p) {
    x6_u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(v) : "v"(p), "i"(OFF) : "memory");
    return v;
}

// the same read with its wait inside the statement: the value is valid when the statement ends, whatever the register
// allocator does next (at NT = 8 the prologue runs at ~270 live registers and the compiler parked the four first fragments
// in AGPRs right behind their reads -- copies of registers whose data had not landed: rows wrong by ~1e-4, found by scanning
// the ISA for uses of pending read destinations)
template <int OFF>
__device__ __forceinline__ x6_u32x4 x6_rd_sync(unsigned int base) {
    x6_u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(base), "i"(OFF));
    return v;
}

// one 16-query block of the sequence-resident kernel's attention (defined behind k_attn16h, whose mathematics it shares)
__device__ __forceinline__ void seq_attn_block(const float *Ks, const char *Vp, int PL, int L, int qb, bool irn, float tgt_add, bool tgt_ok,
                                               int pq, const float *qscr, float4 *of, bool store);
// k and q of the sequence-resident attention are split into float16 planes as they are.  (Measured, profiles/r05/README.md: planes
// of 16 k and 16 q -- low planes of elements below 2^-3 are subnormal float16, 2^-25 absolute instead of 2^-22 relative -- change
// nothing: rows against the two-kernel path max 1.88e-5 / mean 3.98e-7 with the factor, 1.87e-5 / 4.0e-7 without, and four more
// vector instructions per score tile.)  Range: |k|, |q| < 65504 is the V rows' bound (irs_h3_operand_bound: st[6] covers all of W_in).
#define SEQ_KQ_SCALE 1.0f
#ifndef SEQ_ASM_DMA
#define SEQ_ASM_DMA 1
#endif
#ifndef SEQ_EXP
#define SEQ_EXP 0 // (lab, tools/seq_lab.sh: 1 = no attention compute, 2 = asynchronous tail reads: timing experiments, results wrong)
#endif
// SEQ layout of the dynamic LDS behind the ring and the parameter vectors (x6_seq_lds_bytes)
#define X6_SEQ_VECS (256 + 12 * 32 * 4 + 384)   // floats: the parameter vectors + this layer's in-projection bias
#define X6_SEQ_KIMG (3 * 16384 + X6_SEQ_VECS * 4)  // K image of ONE head: 256 token rows x 32 float32, chunk-swizzled (k_attn16h's)
#define X6_SEQ_VIMG (X6_SEQ_KIMG + 32768)        // V image: 2 float16 planes x 256 rows x 64 B
#define X6_SEQ_SCR (X6_SEQ_VIMG + 32768)         // per wave: the q tile as [32 tokens][36] float32
#define X6_SEQ_SCR_B 4608
__host__ __device__ constexpr int x6_seq_lds_bytes() { return X6_SEQ_SCR + 8 * X6_SEQ_SCR_B; }
// QP0 = 0: the tail computes q | k | v; 1: k | v only (feeding the rows-only last layer); 3 (SEQ): no tail
#ifdef X6_STAMP
#define X6_T(v_) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); v_ = t_; }
#if X6_STAMP + 0 == 2 // (phase stamps only: the per-step stamps cost the sequence-resident form 400 spilled scalar registers)
#define X6_TI(v_)
#else
#define X6_TI(v_) X6_T(v_)
#endif
#define X6_PH(i_) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); st_ph[i_] += t_ - st_prev; st_prev = t_; }
#else
#define X6_T(v_)
#define X6_TI(v_)
#define X6_PH(i_)
#endif
// NW = waves per workgroup (4 or 8), each on its own 32 tokens; the 24 1-KB pieces of a step are fetched 24 / NW per wave
// EMBED: the kernel in front of layer 0 -- the accumulator tiles are filled with the embedded tokens (k_embed_frag's
// arithmetic) instead of a layer's result, written to Xf, and only the q | k | v steps (20 .. 31 of a stream whose other
// blocks are unused) run: the same ring, the same step code.
// NT = 8 (d = 256, round 4): 8 accumulator tiles per token -- 128 registers of accumulators, 192 of cached planes -- so ONE
// wave per SIMD (512 registers per lane), one workgroup per CU; the step stream has 96 steps (x6_nstep).
template <int QP0, int NW, bool EMBED, int NT = 4, int NPL = 3, bool SEQ = false>
__global__ void __launch_bounds__(64 * NW, (NT == 8 || NW == 8) ? 1 : 2) k_block_x6(BlockX6Args a) {
    static_assert(!SEQ || (NT == 4 && NPL == 2 && NW == 8 && !EMBED && QP0 == 3), "the sequence-resident form: d = 128, float16 planes, eight waves, no q | k | v tail");
    constexpr int NP = 8 * NPL, STEP_B = x6_step_b(NPL); // pieces (fragments) and bytes of a step
    using x6_plane = typename std::conditional<NPL == 2, x6_f16x8, x6_bf16x8>::type;
    // (SEQ: three ring slots -- the K / V images and the q scratch take the rest of the CU's LDS)
    constexpr int D = 32 * NT, F = 256, NSLOT = SEQ ? 3 : x6_nslot(NPL), LEAD = NSLOT - 1, PPW = NP / NW, HT = NT / 4;
    constexpr int NFRONT = SEQ ? 3 * NT : 0; // SEQ: the q | k | v steps of THIS layer run in front (head-major: q_h, k_h, v_h)
    // the weight planes hold WS x the weights (x6_wscale): an accumulator of plane products holds WS x the product, IWS folds back
    constexpr float WS = x6_wscale(NPL), IWS = 1.0f / WS;
    // RESID_LATE (round 4, d = 128 with float16 planes): the workgroup's start no longer waits for its 128 KB of inputs.  The
    // first three steps' DMA goes out FIRST, then the attention tiles; the residual goes to its OWN tiles, requested behind the
    // second step's barrier (when two attention tiles are dead: all three sets at once spill) and added with b_o in front of
    // LayerNorm 1 -- the out-projection accumulates from zero.  The start barrier waits for step 0's pieces only: vector memory
    // operations retire in order, so "all but the newest 8 + 16" is exactly that, and the first four steps' counted waits
    // allow one tile set beside the newest DMA group; the compiler's own waits bring each attention tile in when its step
    // splits it.  (tools/x6_lab stamps: the start was 16.8 K of a wave's 100 K cycles.)
    constexpr bool RESID_LATE = X6_RESID_LATE && NPL == 2 && NT == 4 && !EMBED && NSLOT == 4 && (NW == 4 || NW == 8);
    // (SEQ keeps the residual apart the same way -- out-projection from zero, residual added with b_o -- without the counted start)
    constexpr bool RESID_SEP = RESID_LATE || SEQ;
    constexpr int NLOAD = 4 * NT; // plain global loads of one set of a wave's input tiles (attention output; residual)
    constexpr int NPRE = x6_npre(NT), NOUT = NT * HT;
    constexpr int S0 = EMBED ? NPRE : 0; // first step of the sequence
    constexpr int V_B1 = 0, V_B2 = F, V_G = F + D, V_B = F + 2 * D, V_BIN = F + 3 * D, V_O = F + 3 * D + 3 * D;
    constexpr int V_BQ = V_O + 6 * D; // (SEQ) this layer's in-projection bias
    static_assert(NT == 4 || NT == 8, "d = 128 or 256");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *vecs = reinterpret_cast<float *>(smem + NSLOT * STEP_B); // b1[256], b2, g, b, b_in[384], b_o, g1, b1n, c, g2, b2n
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lk = lane >> 5;
    unsigned long long st_p[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // (lab) phase boundaries
    X6_T(st_p[0])
    const int m0 = blockIdx.x * (32 * NW);
    const int M = SEQ ? 0x7FFFFFFF : (a.m_dev ? min(a.M, a.m_dev[0]) : a.M);
    // SEQ: this wave's tile = TWO 16-token blocks, one per half h (lanes 16 h .. 16 h + 15 of li): block sh_qb[h] of sequence
    // sh_b[h] (-1: nothing there; a wave with two empty halves still takes part in every step and barrier).  Normally the two are
    // mirror images of ONE sequence (blocks i and nb - 1 - i: causal attention costs qb + 1 key tiles for block qb, so every
    // wave of a sequence gets nb + 1 of them); the odd middle blocks of two sequences may share a tile (k_plan_seq).
    int sh_b[2] = {-1, -1}, sh_qb[2] = {0, 0}, sh_cnt[2] = {0, 0}, sh_pq[2] = {-1, -1}, sh_row0[2] = {0, 0}, sh_pb[2] = {-1, -1};
    int sh_off[2] = {0, 0};
    float sh_ru[2] = {0.f, 0.f};
    bool sh_tgt[2] = {false, false};
    if constexpr (SEQ) {
        if ((int)blockIdx.x >= a.n_wg_dev[0]) return;
        const int tg_ = blockIdx.x * NW + wave;
        const bool irn_ = a.mask_mode == IRS_MASK_IRN;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int b_ = __builtin_amdgcn_readfirstlane(a.tile_seq[2 * tg_ + h]);
            sh_b[h] = b_;
            if (b_ >= 0) {
                sh_qb[h] = __builtin_amdgcn_readfirstlane(a.tile_qb[2 * tg_ + h]);
                const int off_ = __builtin_amdgcn_readfirstlane(a.seq_off[b_]);
                sh_off[h] = off_;
                sh_cnt[h] = __builtin_amdgcn_readfirstlane(a.seq_cnt[b_]);
                sh_pq[h] = __builtin_amdgcn_readfirstlane(a.seq_padq[b_]), sh_row0[h] = __builtin_amdgcn_readfirstlane(a.seq_row0[b_]);
                sh_ru[h] = irn_ ? a.r_u[b_] : 0.f;
                sh_tgt[h] = irn_ && a.seq[(int64_t)b_ * a.L + a.L - 1] != 0;
                sh_pb[h] = (__builtin_amdgcn_readfirstlane(a.seq_qrow[b_]) - off_) >> 4;
            }
        }
    }
    // (RESID_LATE) the twelve parameter values of this thread in ONE batch of unconditional loads.  (The conditional form --
    // "b_o if present", "c_l if present", the halves of the workgroup that hold 128-wide vectors -- compiled to six dependent
    // memory round trips in front of the first DMA issue.  Stores behind the DMA issue would be better still, but any wait the
    // compiler inserts while LDS-DMA pieces are pending is vmcnt(0), whatever it waits for.)
    float pv[RESID_LATE ? 13 : 1];
    if constexpr (RESID_LATE) {
        const int td = tid & (D - 1);
        const float *pbo = a.bo ? a.bo : a.b2, *pc = a.c ? a.c : a.b2, *pg2 = a.c ? a.g2 : a.b2, *pb2n = a.c ? a.b2n : a.b2;
        pv[0] = a.b1[tid], pv[1] = a.bin[tid], pv[2] = a.bin[256 + td], pv[3] = a.b2[td], pv[4] = a.g[td], pv[5] = a.b[td];
        pv[6] = pbo[td], pv[7] = a.g1[td], pv[8] = a.b1n[td], pv[9] = pc[td], pv[10] = pg2[td], pv[11] = pb2n[td];
        pv[12] = 0.f;
    }
    if (!SEQ && m0 >= M) return;
    if constexpr (RESID_LATE) {
        static_assert(NW == 4 || NW == 8, "every thread index below 256 exists");
        // (every value "used" here, by every wave: a load still pending in the waves that skip a store below would cost a
        //  vmcnt(0) in the middle of the DMA issue, where the compiler reuses its register)
        asm volatile("" ::"v"(pv[0]), "v"(pv[1]), "v"(pv[2]), "v"(pv[3]), "v"(pv[4]), "v"(pv[5]), "v"(pv[6]), "v"(pv[7]), "v"(pv[8]),
                     "v"(pv[9]), "v"(pv[10]), "v"(pv[11]));
        if (tid < 256) {
            vecs[V_B1 + tid] = pv[0];
            vecs[V_BIN + tid] = pv[1];
        }
        if (tid < D) {
            vecs[V_BIN + 256 + tid] = pv[2];
            vecs[V_B2 + tid] = pv[3], vecs[V_G + tid] = pv[4], vecs[V_B + tid] = pv[5];
            vecs[V_O + 0 * D + tid] = a.bo ? pv[6] : 0.f;
            vecs[V_O + 1 * D + tid] = pv[7];
            vecs[V_O + 2 * D + tid] = pv[8];
            vecs[V_O + 3 * D + tid] = a.c ? pv[9] : 0.f;
            vecs[V_O + 4 * D + tid] = a.c ? pv[10] : 0.f;
            vecs[V_O + 5 * D + tid] = a.c ? pv[11] : 0.f;
        }
    } else if constexpr (NT == 4) {
        if constexpr (SEQ) {
#pragma unroll
            for (int k = 0; k < (X6_SEQ_VECS + 64 * NW - 1) / (64 * NW); ++k)
                if (tid + 64 * NW * k < X6_SEQ_VECS) vecs[tid + 64 * NW * k] = a.vecpack[tid + 64 * NW * k];
        }
        if (!SEQ && tid < 256) {
            if (!EMBED) vecs[V_B1 + tid] = a.b1[tid];
            vecs[V_BIN + tid] = a.bin[tid];
        }
        if (EMBED) {
            if (tid < D) vecs[V_BIN + 256 + tid] = a.bin[256 + tid];
        } else if (!SEQ && tid < D) {
            vecs[V_B2 + tid] = a.b2[tid];
            vecs[V_G + tid] = a.g[tid];
            vecs[V_B + tid] = a.b[tid];
            vecs[V_BIN + 256 + tid] = a.bin[256 + tid];
            vecs[V_O + 0 * D + tid] = a.bo ? a.bo[tid] : 0.f;
            vecs[V_O + 1 * D + tid] = a.g1[tid];
            vecs[V_O + 2 * D + tid] = a.b1n[tid];
            vecs[V_O + 3 * D + tid] = a.c ? a.c[tid] : 0.f;
            vecs[V_O + 4 * D + tid] = a.c ? a.g2[tid] : 0.f;
            vecs[V_O + 5 * D + tid] = a.c ? a.b2n[tid] : 0.f;
        }
    } else {
        for (int i = tid; i < 3 * D; i += 64 * NW) vecs[V_BIN + i] = a.bin[i];
        if (!EMBED) {
            for (int i = tid; i < F; i += 64 * NW) vecs[V_B1 + i] = a.b1[i];
            for (int i = tid; i < D; i += 64 * NW) {
                vecs[V_B2 + i] = a.b2[i];
                vecs[V_G + i] = a.g[i];
                vecs[V_B + i] = a.b[i];
                vecs[V_O + 0 * D + i] = a.bo ? a.bo[i] : 0.f;
                vecs[V_O + 1 * D + i] = a.g1[i];
                vecs[V_O + 2 * D + i] = a.b1n[i];
                vecs[V_O + 3 * D + i] = a.c ? a.c[i] : 0.f;
                vecs[V_O + 4 * D + i] = a.c ? a.g2[i] : 0.f;
                vecs[V_O + 5 * D + i] = a.c ? a.b2n[i] : 0.f;
            }
        }
    }
#ifdef X6_STAGGER
    if (blockIdx.x >= 256 && blockIdx.x < 512)
        for (int i = 0; i < X6_STAGGER; ++i) __builtin_amdgcn_s_sleep(127); // (lab) second resident workgroup of a CU starts ~4 us x N late
#endif
    int mtile = SEQ ? (int)blockIdx.x * NW + wave : (m0 >> 5) + wave; // (not const: SEQ launders it per layer, see layer_body)
    // (SEQ) this lane's half, its sequence's image row of its token
    const bool l_h1 = li >= 16;
    const int l_b = l_h1 ? sh_b[1] : sh_b[0];
    const int s_j = l_b >= 0 ? 16 * (l_h1 ? sh_qb[1] : sh_qb[0]) + (li & 15) : 0x3FFFFFFF;
    const int l_row0 = l_h1 ? sh_row0[1] : sh_row0[0];
    const int mt = SEQ ? 0x7FFFFFFF : m0 + wave * 32 + li; // packed row (SEQ: no row-major stores)
    size_t fbase = (size_t)mtile * (4 * NT) * 64 + lane;
    const unsigned int lds0 = (unsigned int)(size_t)(__attribute__((address_space(3))) char *)smem;
    unsigned int fr_addr = lds0 + lane * 16; // + slot * X6_STEP_B + piece * 1024
    unsigned int vecs_addr = lds0 + NSLOT * STEP_B + 16 * lk; // this lane's float4 of a 32-value tile's group g: + 32 g bytes
    constexpr int qoff = NOUT * QP0;                // the q tiles' steps (NT tiles x HT) are skipped when QP0
    constexpr int nsteps_c = NFRONT + x6_nstep(NT) - qoff; // executed steps; step i of the sequence is stream block i (+ qoff past the FFN)
    // SEQ: per layer NFRONT + NPRE = 32 steps + ONE empty step (33 = 0 mod NSLOT: the ring's slot numbering repeats from layer
    // to layer); the last layer has its k | v tail (8 steps) in place of the empty step.  `nsteps` only bounds the ring refills:
    // behind a layer that is not the last one the refills go on into the next layer's stream.
    constexpr int SEQ_LSTEPS = NFRONT + NPRE + 1;
    static_assert(!SEQ || SEQ_LSTEPS % NSLOT == 0, "slot numbering repeats per layer");
    // (the non-SEQ kernels see compile-time constants here: `nsteps` and `last` are macros over a constexpr condition)
    int ly = 0, nsteps_rt = nsteps_c;
    bool last_rt = true;
#define nsteps (SEQ ? nsteps_rt : nsteps_c)
#define last (!SEQ || last_rt)
    // DMA of sequence step i into slot i % NSLOT: this wave's pieces PPW wave .. PPW wave + PPW - 1.  The slot holds the
    // step's block in stream order, and the instruction's immediate offset moves the global source AND the LDS destination
    // (tools/dma_probe.hip), so the pieces share one address register pair and one M0: base = the middle piece, offsets
    // -(PPW / 2) .. PPW / 2 - 1 KB (13-bit signed immediates).
    // (NW < 4 -- few tokens, a workgroup per 32 or 64 of them so that more CUs take part: the wave's 12 or 24 pieces go out in
    //  groups of six, one base per group)
    constexpr int PPG = PPW <= 6 ? PPW : (PPW % 6 == 0 ? 6 : 4), NGRP = PPW / PPG;
    static_assert(PPW * NW == NP && PPG * NGRP == PPW, "a step's pieces divide over the waves and their issue groups");
    // (SEQ: layer l's out-projection / FFN / next q | k | v = stream l; ITS q | k | v = the tail region of stream l - 1, layer 0's of
    //  stream nl_total - 1.  dma_src / dma_src_q / dma_src_qn = this layer's, this layer's q | k | v, the NEXT layer's q | k | v.)
    const uint4 *dma_src = (SEQ ? a.Wbase : a.Wx) + (PPW * wave + PPG / 2) * 64 + lane;
    const uint4 *dma_src_q = SEQ ? a.Wbase + (long long)(a.nl_total - 1) * a.wstride + (PPW * wave + PPG / 2) * 64 + lane : dma_src;
    const uint4 *dma_src_qn = dma_src; // (layer l + 1's q | k | v sits in stream l)
    auto issue = [&](int i) __attribute__((always_inline)) {
        // SEQ: front step i = 3 h + c (c = 0 q, 1 k, 2 v of head h) is q | k | v tile NT c + h of the OTHER stream's tail region;
        // body step i is block i - NFRONT of this layer's stream; behind the body: the last layer's k | v tail, else the empty
        // step (nothing to fetch) and then the NEXT layer's front steps
        const int ib = i - NFRONT;
        int blk = (SEQ && i < NFRONT) ? NPRE + NT * (i % 3) + i / 3 : (ib < NPRE ? ib : ib + qoff);
        const uint4 *dsrc = (SEQ && i < NFRONT) ? dma_src_q : dma_src;
        if (SEQ && ib >= NPRE && !last) {
            if (i == SEQ_LSTEPS - 1) return; // the empty step
            const int j = i - SEQ_LSTEPS;
            blk = NPRE + NT * (j % 3) + j / 3;
            dsrc = dma_src_qn;
        }
        x6_static_for<0, NGRP>([&](auto gc) __attribute__((always_inline)) {
            constexpr int grp = decltype(gc)::value;
            const uint4 *src = dsrc + (size_t)blk * (STEP_B / 16) + grp * PPG * 64;
            char *dst = smem + (i % NSLOT) * STEP_B + (PPW * wave + grp * PPG + PPG / 2) * 1024;
            if constexpr (SEQ && SEQ_ASM_DMA) {
                // (the sequence-resident form: the DMA as inline asm.  Behind the builtin the compiler puts s_waitcnt vmcnt(0) in
                //  front of every LDS access it can see -- the image writes, the attention's reads -- i.e. a refill issued before
                //  the attention would be WAITED for there; the first version therefore left the v step's refill to the end of the
                //  head and the next head's first step then waited out the whole DMA latency.  The ring's slots are disjoint from
                //  everything the compiler reads or writes, and every consumer of a slot sits behind a counted wait + barrier.)
                const unsigned int m0v = lds0 + (unsigned int)((i % NSLOT) * STEP_B + (PPW * wave + grp * PPG + PPG / 2) * 1024);
                x6_static_for<0, PPG>([&](auto jc) __attribute__((always_inline)) {
                    constexpr int off = (decltype(jc)::value - PPG / 2) * 1024;
                    const uint4 *src_ = src; // (an asm operand does not capture)
                    const unsigned int m0_ = m0v;
                    if constexpr (decltype(jc)::value == 0)
                        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:%2" :: "v"(src_), "s"(m0_), "n"(off) : "memory", "m0");
                    else
                        asm volatile("global_load_lds_dwordx4 %0, off offset:%1" :: "v"(src_), "n"(off) : "memory");
                });
            } else {
            x6_static_for<0, PPG>([&](auto jc) __attribute__((always_inline)) {
                constexpr int off = (decltype(jc)::value - PPG / 2) * 1024;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)dst, 16, off, 0);
            });
            }
        });
    };
    // accumulators start from the residual x; the attention output tile 0 is requested with it
    f32x16 acc[NT];
    f32x16 at[NT]; // attention output tiles (B operand source of the out-projection): all requested here, so that
                   // the steps carry no plain global load (its wait would be a vmcnt(0) behind the DMA pieces)
    x6_u32x4 resq[RESID_SEP ? NT : 1][4]; // (RESID_LATE, SEQ) the residual as its 16-byte loads, added in front of LayerNorm 1
    x6_u32x4 atq[RESID_SEP ? NT : 1][4]; // (RESID_LATE, SEQ) the attention tiles as their four 16-byte loads
    if constexpr (!EMBED) {
        const float4 *rfrag = reinterpret_cast<const float4 *>(a.Rf) + fbase;
        const float4 *afrag = reinterpret_cast<const float4 *>(a.Af) + fbase;
        if constexpr (RESID_LATE) {
            // (compiler barriers: exactly the 2 x 16 tile loads lie between the DMA issue and the counted wait below)
            asm volatile("" ::: "memory");
            x6_static_for<0, LEAD>([&](auto ic) __attribute__((always_inline)) { issue(S0 + decltype(ic)::value); });
            asm volatile("" ::: "memory");
        }
        if constexpr (SEQ) { // x itself: the B operand of this layer's q | k | v steps (split into planes below)
            // = item_emb[seq] * sqrt(d) + pe[position] of this lane's token (k_embed_frag's arithmetic), gathered here: the
            // separate embedding launch wrote and this prologue re-read 0.5 GB per C2 step and took 90 us; a workgroup's gather
            // is a few microseconds of its ~500.  The tile is stored once: layer 0's residual.  Dead lanes hold zeros.
            const bool live = l_b >= 0 && s_j < (l_h1 ? sh_cnt[1] : sh_cnt[0]);
            const int orig = live ? a.tok_row[(l_h1 ? sh_off[1] : sh_off[0]) + s_j] : 0;
            int64_t id = live ? a.seq[orig] : 0;
            id = id < 0 ? 0 : (id > a.n_item ? a.n_item : id);
            const float *e_ = a.E + id * (int64_t)D + 4 * lk;
            const float *p_ = a.pe + (int64_t)(orig % a.L) * D + 4 * lk;
            float4 *xw = reinterpret_cast<float4 *>(const_cast<float *>(a.Rf)) + fbase;
#pragma unroll
            for (int tn = 0; tn < NT; ++tn)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 e4 = *reinterpret_cast<const float4 *>(e_ + tn * 32 + 8 * g), p4 = *reinterpret_cast<const float4 *>(p_ + tn * 32 + 8 * g);
                    float4 t4 = make_float4(__fadd_rn(__fmul_rn(e4.x, a.sqrtd), p4.x), __fadd_rn(__fmul_rn(e4.y, a.sqrtd), p4.y),
                                            __fadd_rn(__fmul_rn(e4.z, a.sqrtd), p4.z), __fadd_rn(__fmul_rn(e4.w, a.sqrtd), p4.w));
                    if (!live) t4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    xw[(tn * 4 + g) * 64] = t4;
                    acc[tn][4 * g + 0] = t4.x, acc[tn][4 * g + 1] = t4.y, acc[tn][4 * g + 2] = t4.z, acc[tn][4 * g + 3] = t4.w;
                }
            (void)rfrag;
        } else if constexpr (!RESID_LATE) {
#pragma unroll
            for (int tn = 0; tn < NT; ++tn)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 t4 = rfrag[(tn * 4 + g) * 64]; // (the out-projection accumulates WS x its product onto it)
                    acc[tn][4 * g + 0] = t4.x * WS, acc[tn][4 * g + 1] = t4.y * WS, acc[tn][4 * g + 2] = t4.z * WS, acc[tn][4 * g + 3] = t4.w * WS;
                }
        }
        if constexpr (RESID_LATE) {
            // inline-asm loads, tile by tile in order, waited for by COUNTED waits in front of each tile's step: a load the
            // compiler sees gets vmcnt(0) in front of its first use here (every tile and every DMA piece), whatever the order
            x6_static_for<0, NT>([&](auto tc) __attribute__((always_inline)) {
                constexpr int tn = decltype(tc)::value;
                const float4 *tp = afrag + tn * 4 * 64;
                atq[tn][0] = x6_gld<0>(tp), atq[tn][1] = x6_gld<1024>(tp), atq[tn][2] = x6_gld<2048>(tp), atq[tn][3] = x6_gld<3072>(tp);
            });
        } else if constexpr (!SEQ) { // (SEQ: the attention tiles do not exist yet)
#pragma unroll
            for (int tn = 0; tn < NT; ++tn)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 t4 = afrag[(tn * 4 + g) * 64];
                    at[tn][4 * g + 0] = t4.x, at[tn][4 * g + 1] = t4.y, at[tn][4 * g + 2] = t4.z, at[tn][4 * g + 3] = t4.w;
                }
        }
        if constexpr (RESID_LATE) {
#pragma unroll
            for (int tn = 0; tn < NT; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tn][r] = 0.f;
        }
    } else {
        // EMBED: x[token] = item_emb[seq[token]] * sqrt(d) + pe[position], k_embed_frag's arithmetic bit for bit.
        // Round 5: the gather is COALESCED.  The accumulator layout wants lane = token, i.e. 64 lanes reading 32-byte pieces of 32
        // different table rows per instruction, each 128-byte line being touched by four instructions with the whole CU's 256 KB
        // of rows in between (the kernel spent ~140 us of its 313 in front of its first matrix instruction).  Now a wave
        // reads WHOLE rows -- lanes 0-31 the 512 bytes of row 2 i, lanes 32-63 of row 2 i + 1 (D = 256: in two halves) --,
        // scales and adds the position row in that layout, and transposes through LDS: the ring's slots are idle until the
        // first DMA, a wave's 32 x 128 tile is 16 KB of them; 16-byte chunk c of row r sits at chunk c ^ (r & 31), so the row
        // writes (one row = 32 distinct chunks) and the transposed reads (16 lanes = rows r .. r + 15, chunk c ^ r: 16
        // distinct 16-byte slots of the 256-byte bank window) are both conflict-free.
        const int l31 = lane & 31, lhi = lane >> 5;
        char *scr = smem + wave * 16384;
        float4 *xo = reinterpret_cast<float4 *>(a.Xf) + fbase;
        int64_t rid[16];
        int rpos[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { // token 2 i + lhi of this wave's tile (every lane of a half reads the same words)
            const int mr = m0 + wave * 32 + 2 * i + lhi;
            const int orig = mr < M ? (a.tok_row ? a.tok_row[mr] : mr) : 0;
            int64_t id = mr < M ? a.seq[orig] : 0;
            if (id < 0) id = 0;
            if (id > a.n_item) id = a.n_item;
            rid[i] = id, rpos[i] = orig % a.L;
        }
#pragma unroll
        for (int hf = 0; hf < NT / 4; ++hf) { // 128 columns at a time
            float4 ev[16], pv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                ev[i] = *reinterpret_cast<const float4 *>(a.E + rid[i] * (int64_t)D + 128 * hf + 4 * l31);
                pv[i] = *reinterpret_cast<const float4 *>(a.pe + (int64_t)rpos[i] * D + 128 * hf + 4 * l31);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int r = 2 * i + lhi;
                const bool live = m0 + wave * 32 + r < M;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (live)
                    v = make_float4(__fadd_rn(__fmul_rn(ev[i].x, a.sqrtd), pv[i].x), __fadd_rn(__fmul_rn(ev[i].y, a.sqrtd), pv[i].y),
                                    __fadd_rn(__fmul_rn(ev[i].z, a.sqrtd), pv[i].z), __fadd_rn(__fmul_rn(ev[i].w, a.sqrtd), pv[i].w));
                *reinterpret_cast<float4 *>(scr + r * 512 + ((l31 ^ r) << 4)) = v;
            }
            // (wave-private scratch: a wave's LDS operations execute in order, no barrier)
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int tn = 4 * hf + t4, c = 8 * t4 + 2 * g + lk; // columns 32 tn + 8 g + 4 lk .. + 3 of token li
                    const float4 v = *reinterpret_cast<const float4 *>(scr + li * 512 + ((c ^ li) << 4));
                    xo[(tn * 4 + g) * 64] = v;
                    acc[tn][4 * g + 0] = v.x, acc[tn][4 * g + 1] = v.y, acc[tn][4 * g + 2] = v.z, acc[tn][4 * g + 3] = v.w;
                }
        }
        __syncthreads(); // the scratch is the ring: every wave has read its tile before the first DMA piece lands
    }
    auto load_res = [&]() __attribute__((always_inline)) {
        if constexpr (RESID_SEP) {
            const float4 *rfrag = reinterpret_cast<const float4 *>(a.Rf) + fbase;
            x6_static_for<0, NT>([&](auto tc) __attribute__((always_inline)) {
                constexpr int tn = decltype(tc)::value;
                const float4 *tp = rfrag + tn * 4 * 64;
                resq[tn][0] = x6_gld<0>(tp), resq[tn][1] = x6_gld<1024>(tp), resq[tn][2] = x6_gld<2048>(tp), resq[tn][3] = x6_gld<3072>(tp);
            });
        }
    };
    if constexpr (RESID_LATE) {
        // step S0's pieces have landed once at most the two later steps' pieces and the tile loads are in flight; the vecs stores too
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "i"(PPW * (LEAD - 1) + NLOAD) : "memory");
        __builtin_amdgcn_s_barrier();
    } else {
        issue(S0);
        // everything older than this point has landed (the compiler is free to order the plain loads above around the DMA
        // issue, so no counted wait here), the vecs stores too; the later steps go out behind the barrier
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        x6_static_for<1, LEAD>([&](auto ic) __attribute__((always_inline)) { issue(S0 + decltype(ic)::value); });
    }

    // fragment reads: the stream of a step is 8 groups x (plane 0, 1, 2) = 24 reads in consumption order (k_pack_x6); a
    // ring of four register sets, three reads ahead
    // (round 4: RING register sets, RA = RING - 1 reads ahead.  Four sets -- three reads = 3-6 MFMAs = 100-200 cycles ahead --
    //  no longer cover a ds_read_b128 that queues behind seven other waves' reads once float16 planes halve the MFMAs
    //  between two reads; eight sets at d = 128 with float16 planes, where two waves per SIMD leave the registers (236 of 256).  The ring index of a fragment
    //  must not change across a step boundary: NP and NP + 8 are multiples of RING.)
    constexpr int RING = (NT == 4 && NPL == 2) ? X6_RING4 : 4, RA = RING - 1;
    constexpr bool SPLIT_ACC = NT == 4 && NPL == 2 && X6_SPLIT_ACC4; // one-tile steps: the h.h products on their own accumulator (below)
    static_assert(NP % RING == 0 && (NP + 8) % RING == 0 && RA <= NP / 2, "ring index continuity; next-step reads stay behind the barrier");
    x6_u32x4 af[RING];
    auto landed_all = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < RING; ++q) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[q]));
    };
    x6_static_for<0, RING>([&](auto qc) __attribute__((always_inline)) {
        // (the last set gets its own read of the fragment before it: a register copy would be taken before the data has landed)
        constexpr int q = decltype(qc)::value, off = (S0 % NSLOT) * STEP_B + (q < RA ? q : RA - 1) * 1024;
        if constexpr (NT == 8) af[q] = x6_rd_sync<off>(fr_addr);
        else af[q] = x6_rd<off>(fr_addr);
    });
    if constexpr (NT != 8) landed_all();

    // SEQ: the steps whose read-ahead of the NEXT step's head stays pending across compiler-written code with spills in it (the
    // attention region behind a v step, LayerNorm 1 behind the last out-projection step, LayerNorm 3 behind the last FFN-2
    // step): there the reads wait inside their own asm statement.  (The first version spilled three pending fragments in front
    // of LayerNorm 1 -- a tenth of the tiles came out wrong; tools/isa_pending_read_scan.py finds such uses in the ISA.)
    auto seq_sync_tail = [&](int i) __attribute__((always_inline)) {
        return (i < NFRONT && i % 3 == 2) || i == NFRONT + NOUT - 1 || i == NFRONT + NPRE - 1;
    };
    const float invn = 1.0f / (float)D;
    x6_plane X[NPL];
    unsigned long long st_a = 0, st_b = 0, st_c = 0, st_d = 0, st_wait = 0, st_bar = 0, st_iss = 0, st_0 = 0, st_1 = 0, st_steps = 0;
    X6_T(st_0)
    // one pipeline step: 8 groups x 6 MFMAs onto T[0..3]; the mid-step barrier publishes step i + 1 and frees the slot
    // of step i - 1 for the DMA of step i + 2
    // (lab switches, tools/x6_lab.hip: X6_NO_MFMA drops the matrix instructions, X6_NO_DMA the ring refills past the first
    // three steps, X6_NO_SPLIT the plane split arithmetic -- timing experiments, results are then wrong)
#ifdef X6_NO_MFMA
#define X6_MM(W_, X_, T_) asm volatile("" :: "v"(W_), "v"(X_))
#else
#define X6_MM(W_, X_, T_) T_ = x6_mma(W_, X_, T_)
#endif
#define X6_MFMA(Wreg, Xp, T_) X6_MM(__builtin_bit_cast(x6_plane, Wreg), X[Xp], T_)
#ifdef X6_DUMP
#define X6_DUMP_FRAG(I_, f_) if (blockIdx.x == 0 && wave == 0) a.dbg[((I_) * NP + (f_)) * 64 + lane] = __builtin_bit_cast(uint4, af[(f_) & RA]);
#else
#define X6_DUMP_FRAG(I_, f_)
#endif
#ifdef X6_NO_READS
#define X6_READ_AHEAD(f_) asm volatile("" : "+v"(af[(f_) & RA]));
#else
#define X6_READ_AHEAD(f_)                                                                                                \
    if constexpr ((f_) + RA < NP) af[((f_) + RA) & RA] = x6_rd<((f_) + RA) * 1024>(sb_);                                 \
    else if (SEQ && !(SEQ_EXP & 2) && seq_sync_tail(step_)) af[((f_) + RA) & RA] = x6_rd_sync<((f_) + RA - NP) * 1024>(sn_); /* (SEQ: a read that stays pending across a compiler-written phase may be spilled before it lands) */ \
    else af[((f_) + RA) & RA] = x6_rd<((f_) + RA - NP) * 1024>(sn_); /* next step's head (behind this step's barrier) */ \
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(af[(f_) & RA]) : "i"(RA));                                               \
    X6_DUMP_FRAG(step_, f_)
#endif
#ifdef X6_NO_DMA
#define X6_ISSUE(i_)
#else
#define X6_ISSUE(i_) issue(i_)
#endif
#ifdef X6_NO_BARRIER
#define X6_PUBLISH(I_) if ((I_) + LEAD < nsteps) X6_ISSUE((I_) + LEAD);
#else
    // mid-step I: step I + 1 must have landed before the barrier.  With three slots that is this wave's newest DMA group
    // (vmcnt(0)); with four, the group of step I + 2 went out behind it and may stay in flight: vector memory operations
    // retire in order, so "all but the newest PPW" covers step I + 1 whatever plain loads or stores were issued since (they
    // only make the wait stricter).
#define X6_PUBLISH(I_)                                                                                                   \
    if ((I_) + 1 < nsteps) {                                                                                             \
        X6_TI(st_a)                                                                                                       \
        if (RESID_LATE && (I_) < 4) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(PPW * (LEAD - 2) + NLOAD) : "memory"); /* (one tile set lies between the DMA groups) */ \
        else if (LEAD > 2 && (I_) + 2 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(PPW * (LEAD - 2)) : "memory");      \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                            \
        X6_TI(st_b)                                                                                                       \
        __builtin_amdgcn_s_barrier();                                                                                    \
        X6_TI(st_c)                                                                                                       \
        if ((I_) + LEAD < nsteps && !(SEQ && !SEQ_ASM_DMA && (I_) < NFRONT && (I_) % 3 == 2)) X6_ISSUE((I_) + LEAD); /* (SEQ with the builtin DMA: see the front phase) */ \
        if constexpr (RESID_LATE) { if ((I_) == 1) load_res(); }                                                         \
        X6_TI(st_d)                                                                                                       \
        st_wait += st_b - st_a, st_bar += st_c - st_b, st_iss += st_d - st_c;                                            \
        if (SEQ && (I_) < NFRONT && (I_) % 3 == 0) st_ph[10] += st_c - st_b; /* (the first barrier behind a head's attention) */ \
        if (SEQ && (I_) == NFRONT) st_ph[11] += st_c - st_b; /* (the first barrier of the layer body) */                 \
    }
#endif
#define X6_STEP(I_, SRC, T0, T1, T2, T3)                                                                                \
    {                                                                                                                    \
        const int step_ = (I_);                                                                                          \
        unsigned long long sq_0 = 0, sq_1 = 0;                                                                           \
        X6_TI(sq_0)                                                                                                       \
        const unsigned int sb_ = fr_addr + (unsigned int)((step_ % NSLOT) * STEP_B);                                     \
        const unsigned int sn_ = fr_addr + (unsigned int)(((step_ + 1) % NSLOT) * STEP_B);                               \
        x6_static_for<0, NP>([&](auto fc_) __attribute__((always_inline)) {                                              \
            constexpr int f_ = decltype(fc_)::value, g_ = f_ / NPL, pl_ = f_ % NPL;                                      \
            if constexpr (f_ == 0) x6_split(SRC, 0, X);                                                                  \
            if constexpr (f_ == NP / 2) {                                                                                \
                X6_PUBLISH(step_)                                                                                        \
                x6_split(SRC, 1, X);                                                                                     \
            }                                                                                                            \
            X6_READ_AHEAD(f_)                                                                                            \
            f32x16 &T_ = (g_ & 3) == 0 ? T0 : (g_ & 3) == 1 ? T1 : (g_ & 3) == 2 ? T2 : T3;                              \
            /* weight plane pl_ x activation planes NPL - 1 - pl_ .. 0 (the products up to the dropped order), small first */ \
            x6_static_for<0, NPL - pl_>([&](auto qc_) __attribute__((always_inline)) {                                   \
                X6_MFMA(af[f_ & RA], NPL - 1 - pl_ - decltype(qc_)::value, T_);                                           \
            });                                                                                                          \
        });                                                                                                              \
        X6_TI(sq_1)                                                                                                       \
        st_steps += sq_1 - sq_0;                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
    }
    // (reads past the last step of the sequence fetch a stale slot and are never multiplied: the counted waits assume
    // every read of the schedule is in flight)

    // ---- the parameter vectors of the LayerNorm phases (d = 128): inline-asm LDS reads at immediate offsets from ONE base
    // register, two or three groups ahead of their use through a small register ring with counted waits.  (Compiler-visible
    // reads: every one waits for the DMA in flight -- the compiler cannot tell the ring from the vectors --, their 48 addresses
    // are computed steps ahead into 48 registers, and only scheduling barriers kept the reads themselves from being hoisted.)
    // use(gc, p0, p1, p2): group gi = 4 tn + g of the lane's 16 float4 groups; array k's float4 at vecs[Ok + 32 tn + 8 g + 4 lk].
    auto vec_stream = [&](auto na_c, auto o0_c, auto o1_c, auto o2_c, auto &&use) __attribute__((always_inline)) {
        constexpr int NA = decltype(na_c)::value, O0 = decltype(o0_c)::value, O1 = decltype(o1_c)::value, O2 = decltype(o2_c)::value;
        constexpr int NG = 4 * NT, AH = NA == 1 ? 3 : 2, RG = AH + 1;
        x6_u32x4 pr[RG][NA];
        auto rd = [&](auto gc) __attribute__((always_inline)) {
            constexpr int gi = decltype(gc)::value, goff = ((gi >> 2) * 32 + (gi & 3) * 8) * 4;
            pr[gi % RG][0] = x6_rd<O0 * 4 + goff>(vecs_addr);
            if constexpr (NA > 1) pr[gi % RG][1] = x6_rd<O1 * 4 + goff>(vecs_addr);
            if constexpr (NA > 2) pr[gi % RG][2] = x6_rd<O2 * 4 + goff>(vecs_addr);
        };
        x6_static_for<0, AH>(rd);
        x6_static_for<0, NG>([&](auto gc) __attribute__((always_inline)) {
            constexpr int gi = decltype(gc)::value, newer = NA * (gi + AH < NG ? AH : NG - 1 - gi);
            if constexpr (gi + AH < NG) rd(std::integral_constant<int, gi + AH>{});
            if constexpr (NA == 1) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(pr[gi % RG][0]) : "i"(newer));
            if constexpr (NA == 2) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(pr[gi % RG][0]), "+v"(pr[gi % RG][1]) : "i"(newer));
            if constexpr (NA == 3) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(pr[gi % RG][0]), "+v"(pr[gi % RG][1]), "+v"(pr[gi % RG][2]) : "i"(newer));
            use(gc, __builtin_bit_cast(float4, pr[gi % RG][0]), __builtin_bit_cast(float4, pr[gi % RG][NA > 1 ? 1 : 0]),
                __builtin_bit_cast(float4, pr[gi % RG][NA > 2 ? 2 : 0]));
        });
    };
    constexpr bool ASM_VECS = NT == 4;
    using x6_ic0 = std::integral_constant<int, 0>;
    // ---- + b_o, LN1, + c, LN2: register-local (64 of the 128 values here, 64 in lane ^ 32)
    auto layer_norm = [&](auto vb_c, auto vg_c, auto vbeta_c, auto vadd_c) __attribute__((always_inline)) {
        constexpr int vb = decltype(vb_c)::value, vg = decltype(vg_c)::value, vbeta = decltype(vbeta_c)::value, vadd = decltype(vadd_c)::value;
        float sum = 0.f;
        if constexpr (ASM_VECS && vb >= 0) {
            vec_stream(std::integral_constant<int, 1>{}, vb_c, x6_ic0{}, x6_ic0{}, [&](auto gc, const float4 bb, const float4, const float4) __attribute__((always_inline)) {
                constexpr int tn = decltype(gc)::value >> 2, g = decltype(gc)::value & 3;
                // (vb >= 0 = the out-projection's epilogue: the accumulators hold WS x (product [+ residual]))
                if constexpr (RESID_SEP) {
                    const float4 rs = __builtin_bit_cast(float4, resq[tn][g]);
                    acc[tn][4 * g + 0] = __builtin_fmaf(acc[tn][4 * g + 0], IWS, rs.x), acc[tn][4 * g + 1] = __builtin_fmaf(acc[tn][4 * g + 1], IWS, rs.y);
                    acc[tn][4 * g + 2] = __builtin_fmaf(acc[tn][4 * g + 2], IWS, rs.z), acc[tn][4 * g + 3] = __builtin_fmaf(acc[tn][4 * g + 3], IWS, rs.w);
                    acc[tn][4 * g + 0] += bb.x, acc[tn][4 * g + 1] += bb.y, acc[tn][4 * g + 2] += bb.z, acc[tn][4 * g + 3] += bb.w;
                } else {
                    acc[tn][4 * g + 0] = __builtin_fmaf(acc[tn][4 * g + 0], IWS, bb.x), acc[tn][4 * g + 1] = __builtin_fmaf(acc[tn][4 * g + 1], IWS, bb.y);
                    acc[tn][4 * g + 2] = __builtin_fmaf(acc[tn][4 * g + 2], IWS, bb.z), acc[tn][4 * g + 3] = __builtin_fmaf(acc[tn][4 * g + 3], IWS, bb.w);
                }
                sum += (acc[tn][4 * g + 0] + acc[tn][4 * g + 1]) + (acc[tn][4 * g + 2] + acc[tn][4 * g + 3]);
            });
        } else {
#pragma unroll
            for (int tn = 0; tn < NT; ++tn)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if constexpr (vb >= 0) {
                        const float4 bb = *reinterpret_cast<const float4 *>(vecs + vb + tn * 32 + 8 * g + 4 * lk);
                        acc[tn][4 * g + 0] = __builtin_fmaf(acc[tn][4 * g + 0], IWS, bb.x), acc[tn][4 * g + 1] = __builtin_fmaf(acc[tn][4 * g + 1], IWS, bb.y);
                        acc[tn][4 * g + 2] = __builtin_fmaf(acc[tn][4 * g + 2], IWS, bb.z), acc[tn][4 * g + 3] = __builtin_fmaf(acc[tn][4 * g + 3], IWS, bb.w);
                    }
                    sum += (acc[tn][4 * g + 0] + acc[tn][4 * g + 1]) + (acc[tn][4 * g + 2] + acc[tn][4 * g + 3]);
                    if constexpr (!ASM_VECS) __builtin_amdgcn_sched_barrier(0); // (the scheduler otherwise hoists every parameter read of the phase: ~480 registers)
                }
        }
        const float mu = lanes_sum<32>(sum) * invn;
        float qs = 0.f;
#pragma unroll
        for (int tn = 0; tn < NT; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float dlt = acc[tn][r] - mu;
                qs += dlt * dlt;
            }
        const float rstd = 1.0f / sqrtf(lanes_sum<32>(qs) * invn + 1e-5f);
        if constexpr (ASM_VECS) {
            auto norm = [&](auto gc, const float4 gg, const float4 be, const float4 ad_) __attribute__((always_inline)) {
                constexpr int tn = decltype(gc)::value >> 2, g = decltype(gc)::value & 3;
                const float4 ad = vadd >= 0 ? ad_ : make_float4(0.f, 0.f, 0.f, 0.f);
                acc[tn][4 * g + 0] = (acc[tn][4 * g + 0] - mu) * rstd * gg.x + be.x + ad.x;
                acc[tn][4 * g + 1] = (acc[tn][4 * g + 1] - mu) * rstd * gg.y + be.y + ad.y;
                acc[tn][4 * g + 2] = (acc[tn][4 * g + 2] - mu) * rstd * gg.z + be.z + ad.z;
                acc[tn][4 * g + 3] = (acc[tn][4 * g + 3] - mu) * rstd * gg.w + be.w + ad.w;
            };
            if constexpr (vadd >= 0) vec_stream(std::integral_constant<int, 3>{}, vg_c, vbeta_c, vadd_c, norm);
            else vec_stream(std::integral_constant<int, 2>{}, vg_c, vbeta_c, x6_ic0{}, norm);
        } else {
#pragma unroll
            for (int tn = 0; tn < NT; ++tn)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = tn * 32 + 8 * g + 4 * lk;
                    const float4 gg = *reinterpret_cast<const float4 *>(vecs + vg + n);
                    const float4 be = *reinterpret_cast<const float4 *>(vecs + vbeta + n);
                    float4 ad = make_float4(0.f, 0.f, 0.f, 0.f);
                    if constexpr (vadd >= 0) ad = *reinterpret_cast<const float4 *>(vecs + vadd + n);
                    acc[tn][4 * g + 0] = (acc[tn][4 * g + 0] - mu) * rstd * gg.x + be.x + ad.x;
                    acc[tn][4 * g + 1] = (acc[tn][4 * g + 1] - mu) * rstd * gg.y + be.y + ad.y;
                    acc[tn][4 * g + 2] = (acc[tn][4 * g + 2] - mu) * rstd * gg.z + be.z + ad.z;
                    acc[tn][4 * g + 3] = (acc[tn][4 * g + 3] - mu) * rstd * gg.w + be.w + ad.w;
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
    };
    // a 32-value bias tile (vecs offset vo) into an accumulator tile: asm reads (a compiler-visible LDS read here would wait
    // for the DMA issued half a step ago); the wait drains the fragment read-aheads with it, which only makes the next
    // step's counted waits pass early
    auto bias_tile = [&](f32x16 &t, int vo) __attribute__((always_inline)) {
        x6_u32x4 bq[4];
        const unsigned int ba_ = vecs_addr + (unsigned int)(vo * 4);
        bq[0] = x6_rd<0>(ba_), bq[1] = x6_rd<32>(ba_), bq[2] = x6_rd<64>(ba_), bq[3] = x6_rd<96>(ba_);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]));
        landed_all();
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bb = __builtin_bit_cast(float4, bq[g]);
            t[4 * g + 0] = bb.x, t[4 * g + 1] = bb.y, t[4 * g + 2] = bb.z, t[4 * g + 3] = bb.w;
        }
    };
    // One-tile step (FFN-1, QKV): ONE 32-row output tile over the whole K = 128, B operand = the cached planes
    // XP_[k tile][k-step][plane].  The bf16 matrix pipe adds into its float32 accumulator by truncation, so every MFMA on an
    // accumulator of full magnitude costs up to one ulp of it; here the 40 small products (h.m, h.l, m.m, m.h, l.h) go FIRST
    // into the zero-started accumulator, where they and their truncation are 2^-8 of the result, and the 8 h.h products
    // last: 8 full-size truncations per output instead of 48 (the W_h fragments are read from the slot a second time: the
    // step's virtual read sequence is its 24 fragments, then fragments 0, 3, .., 21 again).
    // (round 4, SPLIT_ACC at d = 128: instead of the second read of the W_h fragments the h.h products run on the step's
    //  accumulator and the small products on a zero-started second one, added once at the end of the step -- the same eight
    //  full-size truncations per output, a third fewer LDS reads in these steps (with float16 planes a one-tile step was 24
    //  reads for 24 MFMAs: every wave of the CU asking the LDS pipe for 1 KB per 32 matrix cycles is the pipe's whole
    //  bandwidth), and two independent MFMA chains instead of one.)
    constexpr int NV = SPLIT_ACC ? NP : NP + 8; // fragment reads of a one-tile step
#ifdef X6_NO_READS
#define X6_RD_V(v_)
#define X6_WAIT_V(v_) asm volatile("" : "+v"(af[(v_) & RA]));
#else
#define X6_RD_V(v_)                                                                                                      \
    if constexpr ((v_) < NP) af[(v_) & RA] = x6_rd<(v_) * 1024>(sb_);                                                    \
    else if constexpr ((v_) < NV) af[(v_) & RA] = x6_rd<NPL * ((v_) - NP) * 1024>(sb_);                                  \
    else if (SEQ && !(SEQ_EXP & 2) && seq_sync_tail(step_)) af[(v_) & RA] = x6_rd_sync<((v_) - NV) * 1024>(sn_);             \
    else af[(v_) & RA] = x6_rd<((v_) - NV) * 1024>(sn_); /* next step's head (behind this step's barrier) */
#define X6_WAIT_V(v_) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(af[(v_) & RA]) : "i"(RA));
#endif
#define X6_STEP1(I_, XP_, T_)                                                                                            \
    {                                                                                                                    \
        const int step_ = (I_);                                                                                          \
        unsigned long long sq_0 = 0, sq_1 = 0;                                                                           \
        X6_TI(sq_0)                                                                                                       \
        const unsigned int sb_ = fr_addr + (unsigned int)((step_ % NSLOT) * STEP_B);                                     \
        const unsigned int sn_ = fr_addr + (unsigned int)(((step_ + 1) % NSLOT) * STEP_B);                               \
        f32x16 ts_;                                                                                                      \
        if constexpr (SPLIT_ACC) {                                                                                       \
            _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) ts_[r_] = 0.f;                                             \
        }                                                                                                                \
        x6_static_for<0, NV>([&](auto vc_) __attribute__((always_inline)) {                                              \
            constexpr int v_ = decltype(vc_)::value;                                                                     \
            if constexpr (v_ == NV / 2) { X6_PUBLISH(step_) }                                                            \
            X6_RD_V(v_ + RA)                                                                                             \
            X6_WAIT_V(v_)                                                                                                \
            const x6_plane w_ = __builtin_bit_cast(x6_plane, af[v_ & RA]);                                               \
            if constexpr (v_ < NP) {                                                                                     \
                constexpr int g_ = v_ / NPL, pl_ = v_ % NPL;                                                             \
                X6_DUMP_FRAG(step_, v_)                                                                                  \
                /* the small products of weight plane pl_: activation planes NPL - 1 - pl_ .. (the h.h product waits) */ \
                x6_static_for<0, NPL - pl_>([&](auto qc_) __attribute__((always_inline)) {                               \
                    constexpr int xp_ = NPL - 1 - pl_ - decltype(qc_)::value;                                            \
                    if constexpr (pl_ > 0 || xp_ > 0) {                                                                  \
                        if constexpr (SPLIT_ACC) { X6_MM(w_, XP_[g_ >> 1][g_ & 1][xp_], ts_); }                          \
                        else { X6_MM(w_, XP_[g_ >> 1][g_ & 1][xp_], T_); }                                               \
                    } else if constexpr (SPLIT_ACC) { X6_MM(w_, XP_[g_ >> 1][g_ & 1][0], T_); }                          \
                });                                                                                                      \
            } else {                                                                                                     \
                constexpr int g_ = v_ - NP;                                                                              \
                X6_MM(w_, XP_[g_ >> 1][g_ & 1][0], T_);                                                                  \
            }                                                                                                            \
        });                                                                                                              \
        if constexpr (SPLIT_ACC) {                                                                                       \
            _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) T_[r_] += ts_[r_];                                         \
        }                                                                                                                \
        X6_TI(sq_1)                                                                                                       \
        st_steps += sq_1 - sq_0;                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
    }
    x6_plane Yp[NT][2][NPL];
    if constexpr (SEQ) { // the planes of layer 0's x (every later layer finds them where LayerNorm 3's split left them)
#pragma unroll
        for (int tn = 0; tn < NT; ++tn)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                x6_split(acc[tn], s2, Yp[tn][s2]);
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    float nv[SEQ ? (X6_SEQ_VECS + 64 * NW - 1) / (64 * NW) : 1]; // (SEQ) the next layer's parameter vectors on their way into LDS
    unsigned long long st_q0 = 0, st_q1 = 0, st_qst = 0;
    unsigned long long st_ph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_prev = 0; // (lab, SEQ) cycles per phase, summed over the layers
    X6_T(st_prev)
    auto layer_body = [&]() __attribute__((always_inline)) { // (SEQ: once per layer of this launch; otherwise once)
    if constexpr (SEQ) {
        // Launder what every address of the body is computed from: with the layers in a loop the compiler hoists each of the
        // ~70 loop-invariant addresses (bias tiles, x' stores, attention-tile stores, image rows) out of it and then spills
        // them -- 130 spilled registers, a scratch reload in every step.  Behind these statements the addresses are values of
        // THIS trip and are recomputed where they are used (a few hundred vector instructions per layer).
        {
            unsigned int fb_lo = (unsigned int)fbase, fb_hi = (unsigned int)(fbase >> 32);
            asm volatile("" : "+v"(vecs_addr), "+v"(fr_addr), "+v"(mtile), "+v"(fb_lo), "+v"(fb_hi));
            fbase = ((size_t)fb_hi << 32) | fb_lo;
        }
        last_rt = ly == a.n_lay; // the extra trip: only the front of the model's last layer, the ring ends behind its 12 steps
        nsteps_rt = last_rt ? NFRONT : (1 << 20);
        dma_src = a.Wbase + (long long)ly * a.wstride + (PPW * wave + PPG / 2) * 64 + lane;
        dma_src_q = a.Wbase + (long long)(ly == 0 ? a.nl_total - 1 : ly - 1) * a.wstride + (PPW * wave + PPG / 2) * 64 + lane;
        dma_src_qn = dma_src;
    }
    if constexpr (SEQ) {
        // ================= the sequence-resident front (round 5): this layer's q | k | v from x, head by head, K / V of the head
        // into the workgroup's LDS images, the attention of this wave's own 32 queries, its output to the scratch tiles the
        // out-projection below reads back.  The three steps of a head run back to back with their result tiles in registers; the
        // LDS traffic the compiler can see (image writes, the attention's reads) sits in ONE region per head, behind the v step,
        // and that step leaves its ring refill to the end of the region: a compiler-visible LDS access behind an LDS-DMA issue
        // gets a vmcnt(0) in front, i.e. it would drain the ring at every step.
        float *Kimg = reinterpret_cast<float *>(smem + X6_SEQ_KIMG);
        char *Vimg = smem + X6_SEQ_VIMG;
        float *scr = reinterpret_cast<float *>(smem + X6_SEQ_SCR + wave * X6_SEQ_SCR_B);
        int irow = l_row0 + s_j, li_l = li; // this lane's row of the images (rows [L, 32 T) of a sequence hold finite values of dead tokens)
        asm volatile("" : "+v"(irow), "+v"(li_l)); // (laundered like the addresses above)
        const bool irn_ = a.mask_mode == IRS_MASK_IRN;
        float4 *ao4 = reinterpret_cast<float4 *>(const_cast<float *>(a.Af)) + (size_t)mtile * NT * 4 * 64;
#pragma unroll
        for (int h = 0; h < NT; ++h) {
            f32x16 tq, tk, tv, bt;
#pragma unroll
            for (int r = 0; r < 16; ++r) tq[r] = 0.f, tk[r] = 0.f, tv[r] = 0.f;
            X6_STEP1(3 * h, Yp, tq)
            X6_STEP1(3 * h + 1, Yp, tk)
            X6_STEP1(3 * h + 2, Yp, tv) // (builtin DMA: no ring refill behind this step, see X6_PUBLISH)
            X6_PH(9)
            bias_tile(bt, V_BQ + 32 * h);
            {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4 *>(scr + li_l * 36 + 8 * g + 4 * lk) =
                        make_float4(__builtin_fmaf(tq[4 * g + 0], IWS, bt[4 * g + 0]), __builtin_fmaf(tq[4 * g + 1], IWS, bt[4 * g + 1]),
                                    __builtin_fmaf(tq[4 * g + 2], IWS, bt[4 * g + 2]), __builtin_fmaf(tq[4 * g + 3], IWS, bt[4 * g + 3]));
            }
            bias_tile(bt, V_BQ + D + 32 * h);
            if (l_b >= 0) { // (rows [cnt, 16 nb) hold finite values of dead tokens: read with p = 0)
                // K as two float16 planes (the score products of seq_attn_block): 128 B per key = chunks 0 .. 3 plane h, 4 .. 7
                // plane l (chunk = 8 channels), chunk index ^ ((key >> 1) & 7): a tile's 16 keys x one chunk = 16 bank groups
                typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
                const int swk = (s_j >> 1) & 7;
                char *krow = reinterpret_cast<char *>(Kimg) + irow * 128 + 8 * lk;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f16x4 hp, lp;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = __builtin_fmaf(tk[4 * g + e], IWS, bt[4 * g + e]) * SEQ_KQ_SCALE; // (exact)
                        const _Float16 hv = (_Float16)v;
                        hp[e] = hv;
                        lp[e] = (_Float16)(v - (float)hv);
                    }
                    *reinterpret_cast<f16x4 *>(krow + ((g ^ swk) << 4)) = hp;
                    *reinterpret_cast<f16x4 *>(krow + (((4 + g) ^ swk) << 4)) = lp;
                }
            }
            bias_tile(bt, V_BQ + 2 * D + 32 * h);
            if (l_b >= 0) {
                typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
                const int swv = ((s_j >> 2) & 1) << 1;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f16x4 hp, lp;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = __builtin_fmaf(tv[4 * g + e], IWS, bt[4 * g + e]);
                        const _Float16 hv = (_Float16)v;
                        hp[e] = hv;
                        lp[e] = (_Float16)(v - (float)hv);
                    }
                    char *slot = Vimg + irow * 64 + ((g ^ swv) << 4) + 8 * lk;
                    *reinterpret_cast<f16x4 *>(slot) = hp;
                    *reinterpret_cast<f16x4 *>(slot + 16384) = lp;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            X6_PH(0)
            __builtin_amdgcn_s_barrier(); // every wave's rows of head h are in the images
            asm volatile("" ::: "memory");
            X6_PH(1)
            if (!(SEQ_EXP & 1)) {
#pragma unroll 1
                for (int blk = 0; blk < 2; ++blk) {
                    const int b_ = blk ? sh_b[1] : sh_b[0], qb_ = blk ? sh_qb[1] : sh_qb[0], row0_ = blk ? sh_row0[1] : sh_row0[0];
                    if (b_ >= 0 && last_rt && qb_ != (blk ? sh_pb[1] : sh_pb[0])) continue; // (the model's last layer: only the consumed token's block)
                    // (an empty half: block index beyond a zero-length sequence -> zeros, the layer body multiplies whole tiles)
                    seq_attn_block(Kimg + row0_ * 32, Vimg + row0_ * 64, 16384, b_ >= 0 ? (blk ? sh_cnt[1] : sh_cnt[0]) : 0, b_ >= 0 ? qb_ : 16, irn_,
                                   irn_ ? (1.0f - (blk ? sh_ru[1] : sh_ru[0])) * 1.4426950408889634f : 0.f, blk ? sh_tgt[1] : sh_tgt[0],
                                   blk ? sh_pq[1] : sh_pq[0], scr + 16 * blk * 36, ao4 + (size_t)h * 4 * 64 + 16 * blk, true);
                }
            }
            asm volatile("" ::: "memory");
            // (the next head's image writes come three steps -- three workgroup barriers -- later: every wave is past its reads)
            X6_PH(2)
            if (!SEQ_ASM_DMA && 3 * h + 2 + LEAD < nsteps) issue(3 * h + 2 + LEAD); // (builtin DMA: the refill the v step left out)
        }
        if (last_rt) return; // (the model's last layer goes on, for one row per sequence, in the small-batch kernels)
        // ---- the layer body's inputs: the attention tiles this wave wrote and the residual, as loads the compiler does not see (it
        // would hoist and spread them over the front: registers) with ONE wait; the out-projection accumulates from zero and the
        // residual is added with b_o in front of LayerNorm 1, as in the RESID_LATE form.
        // (the same WAVE wrote the tiles it reads back: workgroup scope -- its stores are complete and its L1 holds no older copy.
        //  An agent-scope fence here is a write-back of the XCD's whole L2 on this chip: the first version spent most of its time in it)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        {
            const float4 *afrag = reinterpret_cast<const float4 *>(a.Af) + fbase;
            x6_static_for<0, NT>([&](auto tc) __attribute__((always_inline)) {
                constexpr int tn = decltype(tc)::value;
                const float4 *tp = afrag + tn * 4 * 64;
                atq[tn][0] = x6_gld<0>(tp), atq[tn][1] = x6_gld<1024>(tp), atq[tn][2] = x6_gld<2048>(tp), atq[tn][3] = x6_gld<3072>(tp);
            });
            load_res();
            static_assert(NT == 4, "sixteen operands per statement");
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(atq[0][0]), "+v"(atq[0][1]), "+v"(atq[0][2]), "+v"(atq[0][3]), "+v"(atq[1][0]), "+v"(atq[1][1]), "+v"(atq[1][2]),
                           "+v"(atq[1][3]), "+v"(atq[2][0]), "+v"(atq[2][1]), "+v"(atq[2][2]), "+v"(atq[2][3]), "+v"(atq[3][0]), "+v"(atq[3][1]),
                           "+v"(atq[3][2]), "+v"(atq[3][3]));
            asm volatile("" : "+v"(resq[0][0]), "+v"(resq[0][1]), "+v"(resq[0][2]), "+v"(resq[0][3]), "+v"(resq[1][0]), "+v"(resq[1][1]),
                              "+v"(resq[1][2]), "+v"(resq[1][3]), "+v"(resq[2][0]), "+v"(resq[2][1]), "+v"(resq[2][2]), "+v"(resq[2][3]),
                              "+v"(resq[3][0]), "+v"(resq[3][1]), "+v"(resq[3][2]), "+v"(resq[3][3]));
#pragma unroll
            for (int tn = 0; tn < NT; ++tn)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 u4 = __builtin_bit_cast(float4, atq[tn][g]);
                    at[tn][4 * g + 0] = u4.x, at[tn][4 * g + 1] = u4.y, at[tn][4 * g + 2] = u4.z, at[tn][4 * g + 3] = u4.w;
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[tn][4 * g + e] = 0.f;
                }
        }
        X6_PH(3)
    }
    // ---- out-projection: acc += W_o . ao^T, k tile t, output half oh: step t HT + oh
    if constexpr (!EMBED) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            // (RESID_LATE: an attention tile "appears" here -- without the statement the compiler splits all four tiles into
            //  planes right behind their loads, i.e. waits for every input tile and DMA piece before the start barrier)
            if constexpr (RESID_LATE) {
                // vector memory operations behind tile t's loads when its step starts: the later tiles, the DMA groups issued
                // at mid-step (one per finished step: steps 3 ..), the residual's 16 loads (behind step 1's barrier)
                const int newer_ = 4 * (NT - 1 - t) + PPW * t + (t >= 2 ? NLOAD : 0);
                asm volatile("s_waitcnt vmcnt(%4)" : "+v"(atq[t][0]), "+v"(atq[t][1]), "+v"(atq[t][2]), "+v"(atq[t][3]) : "n"(newer_));
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 t4 = __builtin_bit_cast(float4, atq[t][g]);
                    at[t][4 * g + 0] = t4.x, at[t][4 * g + 1] = t4.y, at[t][4 * g + 2] = t4.z, at[t][4 * g + 3] = t4.w;
                }
            }
            X6_STEP(NFRONT + t * HT, at[t], acc[0], acc[1], acc[2], acc[3])
            if constexpr (HT == 2) { X6_STEP(t * HT + 1, at[t], acc[4], acc[5], acc[6], acc[7]) }
        }
    }
    // (NT = 8: the LayerNorm phases run at full register pressure and the compiler parks the fragments read ahead for the next
    //  step in AGPRs across them -- copies taken before the data has landed unless the reads are waited for first)
    if constexpr (NT == 8) landed_all();
    X6_T(st_p[1])
    X6_PH(4)
    if constexpr (!EMBED) {
        using x6_icn = std::integral_constant<int, -1>;
        if constexpr (RESID_LATE) { // the residual has landed once only the two DMA groups issued since are in flight
            static_assert(NT == 4, "sixteen operands");
            asm volatile("s_waitcnt vmcnt(%16)"
                         : "+v"(resq[0][0]), "+v"(resq[0][1]), "+v"(resq[0][2]), "+v"(resq[0][3]), "+v"(resq[1][0]), "+v"(resq[1][1]),
                           "+v"(resq[1][2]), "+v"(resq[1][3]), "+v"(resq[2][0]), "+v"(resq[2][1]), "+v"(resq[2][2]), "+v"(resq[2][3]),
                           "+v"(resq[3][0]), "+v"(resq[3][1]), "+v"(resq[3][2]), "+v"(resq[3][3])
                         : "n"(2 * PPW));
        }
        layer_norm(std::integral_constant<int, V_O + 0 * D>{}, std::integral_constant<int, V_O + 1 * D>{},
                   std::integral_constant<int, V_O + 2 * D>{}, std::integral_constant<int, V_O + 3 * D>{});
        if (a.c) layer_norm(x6_icn{}, std::integral_constant<int, V_O + 4 * D>{}, std::integral_constant<int, V_O + 5 * D>{}, x6_icn{});
    }
    X6_T(st_p[2])
    X6_PH(5)

    // ---- feed-forward, one hidden tile (32 units) at a time: FFN-1 step h_ft = W1[32 ft ..] y^T over the whole K = 128
    //      (B operand: y's bf16 planes, split ONCE and kept -- 96 registers -- while acc itself goes on as the residual
    //      accumulator), bias + relu on the 16 values, FFN-2 step acc += W2[:, 32 ft ..] h_ft^T.  (The round-2 order --
    //      all eight hidden tiles, then FFN-2 -- keeps 128 registers of h beside the 64 of y: with the split's
    //      temporaries that is ~265 of the 256 registers two waves per SIMD have: 173 spilled.)
    if constexpr (!EMBED) {
#pragma unroll
        for (int tn = 0; tn < NT; ++tn)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                x6_split(acc[tn], s2, Yp[tn][s2]);
                __builtin_amdgcn_sched_barrier(0);
            }
        if constexpr (SEQ) {
            if (!last) {
#pragma unroll
                for (int k = 0; k < (X6_SEQ_VECS + 64 * NW - 1) / (64 * NW); ++k)
                    nv[k] = (tid + 64 * NW * k < X6_SEQ_VECS) ? a.vecpack[(size_t)(ly + 1) * X6_SEQ_VECS + tid + 64 * NW * k] : 0.f;
            }
        }
        if constexpr (NPL == 2) { // FFN-2 accumulates WS x its product onto y in place: y goes on as WS y (exact)
#pragma unroll
            for (int tn = 0; tn < NT; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tn][r] *= WS;
        }
    }
    if constexpr (!EMBED) {
#pragma unroll
        for (int ft = 0; ft < 8; ++ft) {
            f32x16 hft, bt;
#pragma unroll
            for (int r = 0; r < 16; ++r) hft[r] = 0.f;
            X6_STEP1(NFRONT + NOUT + 2 * HT * ft, Yp, hft)
            // (k tiles 4 .. 7 go onto the same accumulator.  A zero-started second accumulator + one rounded addition --
            //  the first half's full-size sum then sees no further truncating additions -- was measured: the rows' mean
            //  distance to the float32 kernels 1.33e-6 vs 1.39e-6, but 16 more live registers tip the LayerNorm phases
            //  into scratch: 108 spill instructions, 2651 vs 1394 AGPR moves, 4.2 vs 3.85 ms per decode of 1024 users)
            if constexpr (HT == 2) { X6_STEP1(NOUT + 4 * ft + 1, (&Yp[4]), hft) }
            bias_tile(bt, V_B1 + ft * 32);
#pragma unroll
            for (int r = 0; r < 16; ++r) hft[r] = fmaxf(__builtin_fmaf(hft[r], IWS, bt[r]), 0.f);
            X6_STEP(NFRONT + NOUT + 2 * HT * ft + HT, hft, acc[0], acc[1], acc[2], acc[3])
            if constexpr (HT == 2) { X6_STEP(NOUT + 4 * ft + 3, hft, acc[4], acc[5], acc[6], acc[7]) }
        }
    }
    if constexpr (NT == 8) landed_all();
    X6_T(st_p[3])
    X6_PH(6)
    // ---- + b2, LN3 -> x' (fragment-major store), then split ONCE into the plane registers as the QKV tail's B operand
    if constexpr (!EMBED) {
        float sum = 0.f;
        if constexpr (ASM_VECS) {
            vec_stream(std::integral_constant<int, 1>{}, std::integral_constant<int, V_B2>{}, x6_ic0{}, x6_ic0{},
                       [&](auto gc, const float4 bb, const float4, const float4) __attribute__((always_inline)) {
                constexpr int tn = decltype(gc)::value >> 2, g = decltype(gc)::value & 3; // (acc = WS (y + h W2^T))
                acc[tn][4 * g + 0] = __builtin_fmaf(acc[tn][4 * g + 0], IWS, bb.x), acc[tn][4 * g + 1] = __builtin_fmaf(acc[tn][4 * g + 1], IWS, bb.y);
                acc[tn][4 * g + 2] = __builtin_fmaf(acc[tn][4 * g + 2], IWS, bb.z), acc[tn][4 * g + 3] = __builtin_fmaf(acc[tn][4 * g + 3], IWS, bb.w);
                sum += (acc[tn][4 * g + 0] + acc[tn][4 * g + 1]) + (acc[tn][4 * g + 2] + acc[tn][4 * g + 3]);
            });
        } else {
#pragma unroll
        for (int tn = 0; tn < NT; ++tn)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bb = *reinterpret_cast<const float4 *>(vecs + V_B2 + tn * 32 + 8 * g + 4 * lk);
                acc[tn][4 * g + 0] = __builtin_fmaf(acc[tn][4 * g + 0], IWS, bb.x), acc[tn][4 * g + 1] = __builtin_fmaf(acc[tn][4 * g + 1], IWS, bb.y);
                acc[tn][4 * g + 2] = __builtin_fmaf(acc[tn][4 * g + 2], IWS, bb.z), acc[tn][4 * g + 3] = __builtin_fmaf(acc[tn][4 * g + 3], IWS, bb.w);
                sum += (acc[tn][4 * g + 0] + acc[tn][4 * g + 1]) + (acc[tn][4 * g + 2] + acc[tn][4 * g + 3]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const float mu = lanes_sum<32>(sum) * invn;
        float qs = 0.f;
#pragma unroll
        for (int tn = 0; tn < NT; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float dlt = acc[tn][r] - mu;
                qs += dlt * dlt;
            }
        const float rstd = 1.0f / sqrtf(lanes_sum<32>(qs) * invn + 1e-5f);
        if constexpr (ASM_VECS) {
            vec_stream(std::integral_constant<int, 2>{}, std::integral_constant<int, V_G>{}, std::integral_constant<int, V_B>{}, x6_ic0{},
                       [&](auto gc, const float4 gg, const float4 be, const float4) __attribute__((always_inline)) {
                constexpr int tn = decltype(gc)::value >> 2, g = decltype(gc)::value & 3;
                const float4 o = make_float4((acc[tn][4 * g + 0] - mu) * rstd * gg.x + be.x, (acc[tn][4 * g + 1] - mu) * rstd * gg.y + be.y,
                                             (acc[tn][4 * g + 2] - mu) * rstd * gg.z + be.z, (acc[tn][4 * g + 3] - mu) * rstd * gg.w + be.w);
                if (a.Xf) reinterpret_cast<float4 *>(a.Xf)[((size_t)(mtile * NT + tn) * 4 + g) * 64 + lane] = o;
                acc[tn][4 * g + 0] = o.x, acc[tn][4 * g + 1] = o.y, acc[tn][4 * g + 2] = o.z, acc[tn][4 * g + 3] = o.w;
            });
        } else {
#pragma unroll
        for (int tn = 0; tn < NT; ++tn)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = tn * 32 + 8 * g + 4 * lk;
                const float4 gg = *reinterpret_cast<const float4 *>(vecs + V_G + n);
                const float4 be = *reinterpret_cast<const float4 *>(vecs + V_B + n);
                const float4 o = make_float4((acc[tn][4 * g + 0] - mu) * rstd * gg.x + be.x, (acc[tn][4 * g + 1] - mu) * rstd * gg.y + be.y,
                                             (acc[tn][4 * g + 2] - mu) * rstd * gg.z + be.z, (acc[tn][4 * g + 3] - mu) * rstd * gg.w + be.w);
                if (a.Xf) reinterpret_cast<float4 *>(a.Xf)[((size_t)(mtile * NT + tn) * 4 + g) * 64 + lane] = o;
                acc[tn][4 * g + 0] = o.x, acc[tn][4 * g + 1] = o.y, acc[tn][4 * g + 2] = o.z, acc[tn][4 * g + 3] = o.w;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#pragma unroll
    for (int tn = 0; tn < NT; ++tn)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            x6_split(acc[tn], s2, Yp[tn][s2]);
            __builtin_amdgcn_sched_barrier(0);
        }
    X6_T(st_p[4])
    X6_PH(7)
    if (SEQ && !last) {
        // ---- the empty step between two layers: publishes the next layer's first step (fetched at mid-step 31), frees step
        // 31's slot for its second one, re-stages the parameter vectors (every wave is past LayerNorm 3: nothing reads the old
        // ones) BEFORE the refill goes out (a compiler-visible LDS store behind a DMA issue would wait for it), and reads the next
        // step's head inside its own asm statement (compiler code follows)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < (X6_SEQ_VECS + 64 * NW - 1) / (64 * NW); ++k)
            if (tid + 64 * NW * k < X6_SEQ_VECS) vecs[tid + 64 * NW * k] = nv[k];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (visible to the others behind the next step's barrier)
        issue(SEQ_LSTEPS - 1 + LEAD);
        af[0] = x6_rd_sync<0>(fr_addr), af[1] = x6_rd_sync<1024>(fr_addr), af[2] = x6_rd_sync<2048>(fr_addr); // step 0's slot is slot 0
        X6_PH(8)
        return;
    }
    // ---- the next layer's QKV: passes of d output columns (NT tiles); sequence steps NPRE ..
#pragma unroll
    for (int pp = 0; pp < 3; ++pp) {
        if (pp >= 3 - QP0) break;
        const int c0 = D * (pp + QP0);
#pragma unroll
        for (int i = 0; i < NT; ++i) { // output tile NT pp + i: columns c0 + 32 i ..
            f32x16 qt, bt;
#pragma unroll
            for (int r = 0; r < 16; ++r) qt[r] = 0.f;
            X6_STEP1(NFRONT + NPRE + HT * (NT * pp + i), Yp, qt)
            if constexpr (HT == 2) { X6_STEP1(NPRE + 2 * (NT * pp + i) + 1, (&Yp[4]), qt) }
            bias_tile(bt, V_BIN + c0 + i * 32); // (its wait lands every fragment register too: control flow ahead)
            X6_T(st_q0)
#ifdef X6_NO_QKV_STORE
            asm volatile("" :: "v"(qt[0]), "v"(qt[5]), "v"(qt[10]), "v"(qt[15]), "v"(bt[0]));
            if (mt < M && a.M < 0) {
#else
            if (mt < M) {
#endif
                float *qrow = a.QKV + (int64_t)mt * (3 * D) + c0 + i * 32 + 4 * lk;
                if (QP0 == 0 && a.kv_planes && pp == 2) { // V head i of this token as two float16 planes
                    typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
                    char *slot = reinterpret_cast<char *>(a.QKV + (int64_t)mt * (3 * D) + c0 + i * 32) + 8 * lk;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f16x4 hp, lp;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v = __builtin_fmaf(qt[4 * g + e], IWS, bt[4 * g + e]);
                            const _Float16 hv = (_Float16)v;
                            hp[e] = hv;
                            lp[e] = (_Float16)(v - (float)hv);
                        }
                        *reinterpret_cast<f16x4 *>(slot + 16 * g) = hp;       // channels 8 g + 4 lk .. + 3 of plane h
                        *reinterpret_cast<f16x4 *>(slot + 64 + 16 * g) = lp;  // ... of plane l
                    }
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<float4 *>(qrow + 8 * g) =
                            make_float4(__builtin_fmaf(qt[4 * g + 0], IWS, bt[4 * g + 0]), __builtin_fmaf(qt[4 * g + 1], IWS, bt[4 * g + 1]),
                                        __builtin_fmaf(qt[4 * g + 2], IWS, bt[4 * g + 2]), __builtin_fmaf(qt[4 * g + 3], IWS, bt[4 * g + 3]));
                }
            }
            X6_T(st_q1)
            st_qst += st_q1 - st_q0;
        }
    }
    }; // (layer_body)
    if constexpr (SEQ) {
#pragma unroll 1
        for (ly = 0; ly <= a.n_lay; ++ly) layer_body();
    } else
        layer_body();
    landed_all(); // the reads issued past the last step
#ifdef X6_STAMP
    X6_T(st_1)
    if (lane == 0 && a.stamps) {
        unsigned long long *o = a.stamps + (size_t)(blockIdx.x * NW + wave) * (SEQ ? 24 : 8);
        if constexpr (SEQ) {
#pragma unroll
            for (int i = 0; i < 12; ++i) o[8 + i] = st_ph[i];
        }
        o[0] = st_1 - st_p[0], o[1] = st_wait, o[2] = st_bar, o[3] = st_iss, o[4] = st_steps;
        o[5] = st_0 - st_p[0];                                   // prologue
        o[6] = (st_p[2] - st_p[1]) + (st_p[4] - st_p[3]);         // layer norms + plane splits
        o[7] = st_qst;                                            // q | k | v stores
    }
#endif
    (void)st_ph, (void)st_prev;
    (void)st_p, (void)st_q0, (void)st_q1, (void)st_qst;
    (void)st_a, (void)st_b, (void)st_c, (void)st_d, (void)st_wait, (void)st_bar, (void)st_iss, (void)st_0, (void)st_1, (void)st_steps;
#undef nsteps
#undef last
#undef X6_STEP
#undef X6_STEP1
#undef X6_RD_V
#undef X6_WAIT_V
#undef X6_PUBLISH
#undef X6_ISSUE
#undef X6_READ_AHEAD
#undef X6_MFMA
#undef X6_MM
}

// ------------------------------------------------------------------ embed + layer 0's QKV (d = 128, throughput shapes)
// x = item_emb[seq] * sqrt(d) + pe[pos] is computed straight into the transposed accumulator layout (lane = token),
// written once to the fragment-major image (layer 0's residual) and used as the B operand of the QKV projection,
// exactly like k_block's tail: the first layer needs no separate embed kernel and no QKV GEMM reading x back.
struct EmbedQkvArgs {
    const int64_t *seq;
    const float *E, *pe;
    const int32_t *tok_row, *m_dev;
    int rows, L;
    float sqrtd;
    int64_t n_item;
    float *Xf;
    const float *Win, *bin;
    float *QKV;
};

__global__ void __launch_bounds__(256, 2) k_embed_qkv(EmbedQkvArgs a) {
    constexpr int D = 128, BK = 16, NR = 192; // 192 W_in rows (6 output tiles) per pass
    __shared__ __attribute__((aligned(16))) float sm[2 * NR * BK + 3 * D];
    float *vb = sm + 2 * NR * BK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int m0 = blockIdx.x * 128;
    const int M = a.m_dev ? min(a.rows, a.m_dev[0]) : a.rows;
    if (m0 >= M) return;
    vb[tid] = a.bin[tid];
    if (tid < D) vb[256 + tid] = a.bin[256 + tid];
    const int mtile = (m0 >> 5) + wave;
    const int mt = m0 + wave * 32 + li;
    const bool live = mt < M;
    float4 wv[3];
    auto load_slab = [&](int t) { // t = 8 pp + ss: W_in[192 pp .. 192 pp + 191][16 ss ..]
        const int pp = t >> 3, ss = t & 7;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int idx = tid + i * 256;
            const float *p = a.Win + (size_t)(NR * pp + (idx >> 2)) * D + ss * BK + (idx & 3) * 4;
            wv[i] = make_float4(p[0], p[1], p[2], p[3]);
        }
    };
    auto store_slab = [&](float *S) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int idx = tid + i * 256;
            const int r = idx >> 2, c = idx & 3;
            *reinterpret_cast<float4 *>(S + r * BK + ((c ^ lin_swz<BK>(r)) << 2)) = wv[i];
        }
    };
    load_slab(0);
    // ---- embed: this lane's token, columns 32tn + 8g + 4lk + e (same arithmetic as k_embed_frag)
    f32x16 acc[4];
    {
        const int orig = live ? (a.tok_row ? a.tok_row[mt] : mt) : 0;
        int64_t id = live ? a.seq[orig] : 0;
        if (id < 0) id = 0;
        if (id > a.n_item) id = a.n_item;
        const float *e = a.E + id * (int64_t)D;
        const float *p = a.pe + (int64_t)(orig % a.L) * D;
        float4 *xo = reinterpret_cast<float4 *>(a.Xf) + (size_t)mtile * 16 * 64 + lane;
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = tn * 32 + 8 * g + 4 * lk;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (live) {
                    const float4 ev = *reinterpret_cast<const float4 *>(e + n), pv = *reinterpret_cast<const float4 *>(p + n);
                    v = make_float4(__fadd_rn(__fmul_rn(ev.x, a.sqrtd), pv.x), __fadd_rn(__fmul_rn(ev.y, a.sqrtd), pv.y),
                                    __fadd_rn(__fmul_rn(ev.z, a.sqrtd), pv.z), __fadd_rn(__fmul_rn(ev.w, a.sqrtd), pv.w));
                }
                xo[(tn * 4 + g) * 64] = v;
                acc[tn][4 * g + 0] = v.x, acc[tn][4 * g + 1] = v.y, acc[tn][4 * g + 2] = v.z, acc[tn][4 * g + 3] = v.w;
            }
    }
    store_slab(sm);
    __syncthreads();
    const int sw = lin_swz<BK>(li);
    int cur = 0;
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
        f32x16 qa[6];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) qa[i][r] = 0.f;
#pragma unroll
        for (int ss = 0; ss < 8; ++ss) {
            const int t = pp * 8 + ss;
            if (t + 1 < 16) load_slab(t + 1);
            const float *wa = sm + cur * NR * BK + li * BK;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int off = ((2 * q + lk) ^ sw) << 2;
                const int tn = ss >> 1, g = 2 * (ss & 1) + q;
                float4 w[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) w[i] = *reinterpret_cast<const float4 *>(wa + i * 32 * BK + off);
#define EQ_STEP(E_, R)                                                                                        \
    _Pragma("unroll") for (int i = 0; i < 6; ++i)                                                             \
        qa[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[i].E_, acc[tn][4 * g + R], qa[i], 0, 0, 0);
                EQ_STEP(x, 0) EQ_STEP(y, 1) EQ_STEP(z, 2) EQ_STEP(w, 3)
#undef EQ_STEP
            }
            if (t + 1 < 16) store_slab(sm + (cur ^ 1) * NR * BK);
            __syncthreads();
            cur ^= 1;
        }
        if (live) {
            float *qrow = a.QKV + (int64_t)mt * (3 * D) + pp * NR + 4 * lk;
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 bb = *reinterpret_cast<const float4 *>(vb + pp * NR + i * 32 + 8 * g + 4 * lk);
                    *reinterpret_cast<float4 *>(qrow + i * 32 + 8 * g) =
                        make_float4(qa[i][4 * g + 0] + bb.x, qa[i][4 * g + 1] + bb.y, qa[i][4 * g + 2] + bb.z, qa[i][4 * g + 3] + bb.w);
                }
        }
    }
}

// ------------------------------------------------------------------ small-M linear (latency path, few sequences)
// With a few hundred rows the 128x128 tiling uses 2-6 workgroups of the 256 CUs.  Here one WAVE
// owns one 32x32 output tile and loads its operands straight from L2 into MFMA-fragment registers
// (lane (i, kk) reads 16-byte pieces k = 8q + 4kk of row i of X and of W): no LDS, no barriers,
// M/32 x N/32 waves in flight.  K <= 256.
template <bool LNV>
__global__ void __launch_bounds__(256) k_linear_small(LinArgs a) {
    __shared__ float part[2][4][32];      // LN variant: per-wave partial row statistics
    __shared__ float vecs[LNV ? 6 * 128 : 4]; // LN variant: bias, g1, b1, c, g2, b2 (zero padded)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int M = a.m_dev ? min(a.M, a.m_dev[0]) : a.M, N = a.N, K = a.K;
    const int m0 = blockIdx.y * 32;
    if (m0 >= M) return;
    const int n0 = LNV ? wave * 32 : (blockIdx.x * 4 + wave) * 32;
    const bool tile_ok = n0 < N; // whole-wave condition
    const bool kvec = (K % 4 == 0) && ((((uintptr_t)a.X) & 15) == 0) && ((((uintptr_t)a.W) & 15) == 0);
    if (LNV && tid < 128) {
        const bool in = tid < N;
        vecs[0 * 128 + tid] = (in && a.bias) ? a.bias[tid] : 0.f;
        vecs[1 * 128 + tid] = in ? a.g1[tid] : 0.f;
        vecs[2 * 128 + tid] = in ? a.b1[tid] : 0.f;
        vecs[3 * 128 + tid] = (in && a.c) ? a.c[tid] : 0.f;
        vecs[4 * 128 + tid] = (in && a.c) ? a.g2[tid] : 0.f;
        vecs[5 * 128 + tid] = (in && a.c) ? a.b2[tid] : 0.f;
    }
    // LNV: D[n][m] (lane = token, A = W rows, B = X rows); else D[m][n] (lane = column)
    const int xr = m0 + li, wr = n0 + li;
    const float *xrow = a.X + (int64_t)(xr < M ? xr : 0) * K;
    const float *wrow = a.W + (int64_t)(wr < N ? wr : 0) * K;
    const int nq = (K + 7) / 8;
    auto load_chunk = [&](int q0, float4 (&xf)[8], float4 (&wf)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k0 = 8 * (q0 + j) + 4 * lk;
            float4 xt = make_float4(0.f, 0.f, 0.f, 0.f), wt = xt;
            if (q0 + j < nq && k0 < K) {
                if (kvec && k0 + 3 < K) {
                    if (xr < M) xt = *reinterpret_cast<const float4 *>(xrow + k0);
                    if (wr < N) wt = *reinterpret_cast<const float4 *>(wrow + k0);
                } else {
                    if (xr < M) {
                        xt.x = xrow[k0];
                        if (k0 + 1 < K) xt.y = xrow[k0 + 1];
                        if (k0 + 2 < K) xt.z = xrow[k0 + 2];
                        if (k0 + 3 < K) xt.w = xrow[k0 + 3];
                    }
                    if (wr < N) {
                        wt.x = wrow[k0];
                        if (k0 + 1 < K) wt.y = wrow[k0 + 1];
                        if (k0 + 2 < K) wt.z = wrow[k0 + 2];
                        if (k0 + 3 < K) wt.w = wrow[k0 + 3];
                    }
                }
            }
            xf[j] = xt;
            wf[j] = wt;
        }
    };
    auto mma_chunk = [&](f32x16 &acc, const float4 (&xf)[8], const float4 (&wf)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (LNV) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j].x, xf[j].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j].y, xf[j].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j].z, xf[j].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j].w, xf[j].w, acc, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xf[j].x, wf[j].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xf[j].y, wf[j].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xf[j].z, wf[j].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xf[j].w, wf[j].w, acc, 0, 0, 0);
            }
        }
    };
    // epilogue inputs issued before the K loop (independent of it)
    float resv[16];
    const int mt = m0 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        resv[r] = 0.f;
        if (LNV) {
            const int n = n0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (n < N && mt < M) resv[r] = a.R[(int64_t)mt * N + n];
        } else if (a.R) {
            const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * lk, n = n0 + li;
            if (m < M && n < N) resv[r] = a.R[(int64_t)m * N + n];
        }
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (tile_ok) { // two register sets: the next 64 k are in flight while the current 64 are multiplied
        float4 xa[8], wa[8], xb[8], wb[8];
        load_chunk(0, xa, wa);
        for (int q0 = 0; q0 < nq; q0 += 16) {
            if (q0 + 8 < nq) load_chunk(q0 + 8, xb, wb);
            mma_chunk(acc, xa, wa);
            if (q0 + 8 < nq) {
                if (q0 + 16 < nq) load_chunk(q0 + 16, xa, wa);
                mma_chunk(acc, xb, wb);
            }
        }
    }
    if (!LNV) {
        const int n = n0 + li;
        if (!tile_ok || n >= N) return;
        const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (m < M) {
                float v = acc[r] + bv;
                if (a.relu) v = fmaxf(v, 0.f);
                v += resv[r];
                a.Y[(int64_t)m * N + n] = v;
            }
        }
        return;
    }
    // ---- LN variant: lane = token m0 + li; registers = columns n0 + (r&3) + 8(r>>2) + 4lk of this wave's 32
    __syncthreads(); // vecs
    const float invn = 1.0f / (float)N;
    float z[16];
    float s1 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = n0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        float v = 0.f;
        if (n < N && mt < M) v = acc[r] + vecs[n] + resv[r];
        z[r] = v;
        s1 += v;
    }
    auto row_total = [&](float v, int slot) { // sum over the 128 columns of each token (4 waves x 2 lane halves)
        v = lanes_sum<32>(v);
        if (lk == 0) part[slot][wave][li] = v;
        __syncthreads();
        float t = part[slot][0][li] + part[slot][1][li] + part[slot][2][li] + part[slot][3][li];
        __syncthreads();
        return t;
    };
    float mu = row_total(s1, 0) * invn;
    float q = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = n0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        const float dlt = (n < N) ? z[r] - mu : 0.f;
        q += dlt * dlt;
    }
    float rstd = 1.0f / sqrtf(row_total(q, 1) * invn + 1e-5f);
    s1 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = n0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        float y = 0.f;
        if (n < N) y = (z[r] - mu) * rstd * vecs[128 + n] + vecs[256 + n] + vecs[384 + n];
        z[r] = y;
        s1 += y;
    }
    if (a.c) {
        mu = row_total(s1, 0) * invn;
        q = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            const float dlt = (n < N) ? z[r] - mu : 0.f;
            q += dlt * dlt;
        }
        rstd = 1.0f / sqrtf(row_total(q, 1) * invn + 1e-5f);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (n < N) z[r] = (z[r] - mu) * rstd * vecs[512 + n] + vecs[640 + n];
        }
    }
    if (mt < M) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = n0 + 8 * g + 4 * lk;
            float *dst = a.Y + (int64_t)mt * N + n;
            if ((N % 4 == 0) && n + 3 < N && ((((uintptr_t)a.Y) & 15) == 0))
                *reinterpret_cast<float4 *>(dst) = make_float4(z[4 * g], z[4 * g + 1], z[4 * g + 2], z[4 * g + 3]);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < N) dst[e] = z[4 * g + e];
            }
        }
    }
}

// ------------------------------------------------------------------ small-batch fused layer tail (latency path, d = 128, F = 256)
// With up to a few thousand rows a decoder layer is launch and latency bound: out-proj+LN, FFN1, FFN2+LN and the next
// QKV as separate launches cost ~10-15 us each for ~6 us of dependent work, and the 128-token kernels of the
// throughput path have a ~100 us floor per layer.  Here one workgroup owns 16 tokens and runs the whole chain on
// v_mfma_f32_16x16x4_f32, the waves splitting the OUTPUT columns of every GEMM in the transposed orientation
// (lane = (token lq, k-slot gq): B value x[token][16j + 4gq + e], A value W[n0 + lq][16j + 4gq + e], C register r <->
// column n0 + 4gq + r); the activation tiles pass from GEMM to GEMM through LDS ([16 tokens][K + 4] floats: the +4
// stride keeps the B-fragment ds_read_b128 conflict free), the weights come straight from L2 into A-fragment
// registers, and the LayerNorm row statistics are combined across the waves through a small LDS scratch.
struct SmallBlockArgs {
    const float *AO, *X;      // attention output, residual x: row-major [M][128]
    const float *Wo, *bo, *g1, *b1n, *c, *g2, *b2n;
    const float *W1, *b1, *W2, *b2, *g3, *b3n;
    float *Xo;                // x' row-major [M][128]
    const float *Win, *bin;   // next layer's in-projection (may be null)
    float *QKV;               // [M][384]
    int M;
    const int32_t *m_dev;
    const int32_t *xidx;      // optional: residual row of token m is X[xidx[m]] (last layer: the consumed rows)
    const float *Wf, *Wfin;   // fragment-packed copies (k_pack_frag16) of {Wo, W1, W2} of this layer and of Win
    // fused self-attention (single sequence, see k_block_small16<.., ATT = true>): this layer's q | k | v rows
    const float *QKVin, *r_u;
    const int64_t *seq_last;  // the window's last token (the IRN target: 0 = no target)
    const int32_t *cnt, *padq;
    int mask_mode;
};

// Weights of the 16-token latency kernel, re-ordered so that every wave load instruction of an MFMA A fragment reads
// 1 KB of consecutive addresses: out[(((nt * K/64 + r) * 4 + j) * 64 + lane) * 4 + e] = W[16 nt + lq][64 r + 16 j + 4 gq + e],
// lane = 16 gq + lq.  A lone workgroup fetches row-major fragments (16 rows x 64 B per instruction) at ~30 GB/s and
// packed ones at ~59 GB/s (tools/small_lab.hip), and that fetch is what bounds the kernel.
#define SMALL_WF_WO 0
#define SMALL_WF_W1 16384
#define SMALL_WF_W2 49152
#define SMALL_WF_LAYER 81920 // floats per layer of {Wo, W1, W2}
#define SMALL_WF_WIN 49152   // floats per layer of Win
__global__ void k_pack_frag16(const float *__restrict__ W, float *__restrict__ out, int N, int K) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x; // one float4 of the packed image
    if (i >= N * K / 4) return;
    const int lane = i & 63, j = (i >> 6) & 3, rr = i >> 8;
    const int r = rr % (K / 64), nt = rr / (K / 64);
    const int lq = lane & 15, gq = lane >> 4;
    reinterpret_cast<float4 *>(out)[i] = *reinterpret_cast<const float4 *>(W + (size_t)(16 * nt + lq) * K + 64 * r + 16 * j + 4 * gq);
}

// What bounds it (tools/small_lab.hip): a workgroup fetches its 512 KB of weights at ~38 B/clk per CU (~14K cycles),
// its 2048 MFMAs keep each SIMD busy 16.4K cycles, and ~10K cycles go to the prologue and the LayerNorm exchanges.
#ifdef IRS_SMALL_TIMING
__device__ unsigned long long g_small_t[16];
#define STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_small_t[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(i)
#endif
#define SMALL_ROWS_MAX 32768 // packed rows up to which the 16-token kernel beats the 128-token path (see irs_launch_decode)
#define SMALL_MT2_ROWS 8192
#define SB_NW 8 // waves per workgroup: two per SIMD, so that one wave's barrier / LDS / load waits hide behind the other's MFMAs
// MT = 16-token tiles per workgroup: 1 on the latency path; 2 above SMALL_MT2_ROWS rows, where every weight fragment
// is then fetched once per 32 tokens and feeds two MFMAs.
// ATT = true (one sequence, head dim 32, 8 waves, MT = 1): the self-attention of the 16-token tile runs in front, in
// the same launch -- wave w takes head w & 3 and the key tiles of parity w >> 2, K / Q / V fragments come straight
// from global memory (no LDS staging, nothing to wait for but one round trip), the two halves of a head are combined
// through LDS and the attention output lands in the LDS tile the out-projection reads.  The first weight rounds are
// requested before the attention starts, so their latency hides behind it.  q | k | v ping-pong between two buffers
// (other workgroups of this launch still read this layer's k | v while this one writes the next layer's).
template <bool QKV, int MT, bool ATT = false>
__global__ void __launch_bounds__(64 * SB_NW) k_block_small16(SmallBlockArgs a) {
    static_assert(!ATT || (MT == 1 && SB_NW == 8), "fused attention: one token tile, eight waves");
    constexpr int T1 = 8 / SB_NW, T2 = 16 / SB_NW, T3 = 24 / SB_NW; // 16-column tiles per wave of a 128 / 256 / 384 wide output
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    constexpr int D = 128, F = 256, LDA = D + 4, LDH = F + 4;
    __shared__ __attribute__((aligned(16))) float bufA[16 * MT * LDA]; // ao -> y -> x'
    __shared__ __attribute__((aligned(16))) float bufH[16 * MT * LDH]; // h
    __shared__ float part[2][SB_NW][16 * MT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    STAMP(0);
    const int lq = lane & 15, gq = lane >> 4;
    // One ROUND = 64 k of NT independent 16-column tiles: A values W[n0 + 16t + lq][64r + 16j + 4gq + e] come
    // straight from L2 (fragment-packed copies, see k_pack_frag16), B values from the LDS tile.  At this size nothing but the kernel's own instruction stream
    // can hide the L2 round trip, so the rounds of the four GEMMs form one software pipeline: round i+1's weight
    // fragments (they do not depend on the activations) are requested before round i's MFMAs, across the LayerNorm
    // epilogues and barriers as well.  ~11 workgroups are resident in this regime and every kernel starts with cold
    // caches (a weight fragment takes ~1 us to arrive), so the look-ahead is as deep as the register file allows:
    // up to 16 tile-rounds (256 registers) in flight, the first 12 requested before anything else happens.
    float4 wo[2][T1][4], w1[2][T2][4], w2[4][T1][4], wq[2][T3][4];
#define W_LOAD(wf, NT, W, n0_, K, r)                                                                                  \
    _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                                                  \
        const float *wfr = (W) + ((((n0_) / 16 + t) * ((K) / 64) + (r)) * 4) * 256 + lane * 4;                        \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) wf[t][j] = *reinterpret_cast<const float4 *>(wfr + 256 * j);    \
    }                                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);
#define MMA_ROUND(acc, wf, NT, B, ldb, r)                                                                             \
    {                                                                                                                 \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                               \
            float4 bf[MT];                                                                                            \
            _Pragma("unroll") for (int u = 0; u < MT; ++u)                                                            \
                bf[u] = *reinterpret_cast<const float4 *>((B) + (16 * u + lq) * (ldb) + 4 * gq + 64 * (r) + 16 * j);  \
            _Pragma("unroll") for (int t = 0; t < NT; ++t) _Pragma("unroll") for (int u = 0; u < MT; ++u) {           \
                acc[u][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t][j].x, bf[u].x, acc[u][t], 0, 0, 0);            \
                acc[u][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t][j].y, bf[u].y, acc[u][t], 0, 0, 0);            \
                acc[u][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t][j].z, bf[u].z, acc[u][t], 0, 0, 0);            \
                acc[u][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t][j].w, bf[u].w, acc[u][t], 0, 0, 0);            \
            }                                                                                                         \
        }                                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
    }
    // the per-column vectors ride the same pipeline: requested a phase ahead, never on the critical path
#define V_LOAD(v, NT, p, n0_)                                                                                         \
    _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                                    \
    {                                                                                                                 \
        const float4 ld_ = *reinterpret_cast<const float4 *>(((p) ? (p) : a.g3) + (n0_) + 16 * t + 4 * gq);          \
        v[t] = (p) ? ld_ : make_float4(0.f, 0.f, 0.f, 0.f);                                                           \
    }
    float4 zero2[T1];
#pragma unroll
    for (int t = 0; t < T1; ++t) zero2[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 vbo[T1], vg1[T1], vb1n[T1], vc[T1], vg2[T1], vb2n[T1], vb1[T2], vb2[T1], vg3[T1], vb3n[T1], vbin[T3];
    // A wave's loads return in order: the activation tile (and the residual row index) go first, clamped to the
    // static row bound and masked once the device-side count is known, then the 6-12 tile-rounds of weights.
    const int m0 = blockIdx.x * 16 * MT;
    const int mt = m0 + lq; // this lane's tokens: mt + 16 u
    constexpr int NAO = 16 * MT * (D / 4) / (64 * SB_NW);
    float4 aov[NAO];
    if constexpr (!ATT) {
#pragma unroll
        for (int u = 0; u < NAO; ++u) {
            const int i = tid + u * 64 * SB_NW, rr = i / (D / 4), c4 = i % (D / 4);
            aov[u] = *reinterpret_cast<const float4 *>(a.AO + (int64_t)min(m0 + rr, a.M - 1) * D + 4 * c4);
        }
    }
    // ---- fused attention, part 1: requests.  lane (lq, gq): query m0 + lq / key 16kt + lq, head columns 8gq .. 8gq+7
    constexpr int ATN = 8; // key tiles per wave (L <= 256: 16 tiles over the two halves)
    const int ah = wave & 3, ahalf = wave >> 2, aqb = blockIdx.x;
    const int aL = ATT ? a.cnt[0] : 0;
    const int ant = ATT ? (aqb >= ahalf ? (aqb - ahalf) / 2 + 1 : 0) : 0; // my key tiles: ahalf, ahalf + 2, ... <= aqb
    float4 aq0, aq1, akf[ATN][2], aktg0, aktg1, avt0, avt1;
    float avf[ATN][2][4]; // V fragments (A operand of O^T += V^T P^T: lane (col lq, k-slot gq), MFMA r <-> key 16kt + 4gq + r)
    if constexpr (ATT) {
        const int ld = 3 * D;
        const float *qrow = a.QKVin + (int64_t)min(m0 + lq, aL - 1) * ld + ah * 32 + 8 * gq;
        aq0 = *reinterpret_cast<const float4 *>(qrow);
        aq1 = *reinterpret_cast<const float4 *>(qrow + 4);
#pragma unroll
        for (int i = 0; i < ATN; ++i) {
            if (i < ant) { // wave-uniform
                const float *kr = a.QKVin + (int64_t)min(16 * (ahalf + 2 * i) + lq, aL - 1) * ld + D + ah * 32 + 8 * gq;
                akf[i][0] = *reinterpret_cast<const float4 *>(kr);
                akf[i][1] = *reinterpret_cast<const float4 *>(kr + 4);
            }
        }
        const float *ktr = a.QKVin + (int64_t)(aL - 1) * ld + D + ah * 32 + 8 * gq; // the IRN target column's key
        aktg0 = *reinterpret_cast<const float4 *>(ktr);
        aktg1 = *reinterpret_cast<const float4 *>(ktr + 4);
        const float *vbase = a.QKVin + 2 * D + ah * 32 + lq;
#pragma unroll
        for (int i = 0; i < ATN; ++i) {
            if (i < ant) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float *vr = vbase + (int64_t)min(16 * (ahalf + 2 * i) + 4 * gq + r, aL - 1) * ld;
                    avf[i][0][r] = vr[0];
                    avf[i][1][r] = vr[16];
                }
            }
        }
        const float *vtr = a.QKVin + (int64_t)(aL - 1) * ld + 2 * D + ah * 32 + 4 * gq; // V[L-1][16ct + 4gq ..]
        avt0 = *reinterpret_cast<const float4 *>(vtr);
        avt1 = *reinterpret_cast<const float4 *>(vtr + 16);
    }
    int xrow[MT];
#pragma unroll
    for (int u = 0; u < MT; ++u) xrow[u] = a.xidx ? a.xidx[min(mt + 16 * u, a.M - 1)] : mt + 16 * u;
    __builtin_amdgcn_sched_barrier(0);
    W_LOAD(wo[0], T1, a.Wf + SMALL_WF_WO, wave * 16 * T1, D, 0);
    W_LOAD(wo[1], T1, a.Wf + SMALL_WF_WO, wave * 16 * T1, D, 1);
    const int M = a.m_dev ? min(a.M, a.m_dev[0]) : a.M;
    bool live[MT];
#pragma unroll
    for (int u = 0; u < MT; ++u) live[u] = mt + 16 * u < M;
    if constexpr (!ATT) { // (with the attention in front, FFN1's weights are requested after it: register budget)
        W_LOAD(w1[0], T2, a.Wf + SMALL_WF_W1, wave * 16 * T2, D, 0);
        W_LOAD(w1[1], T2, a.Wf + SMALL_WF_W1, wave * 16 * T2, D, 1);
    }
    V_LOAD(vbo, T1, a.bo, wave * 16 * T1);
    V_LOAD(vg1, T1, a.g1, wave * 16 * T1);
    V_LOAD(vb1n, T1, a.b1n, wave * 16 * T1);
    V_LOAD(vc, T1, a.c, wave * 16 * T1);
    V_LOAD(vg2, T1, a.g2, wave * 16 * T1);
    V_LOAD(vb2n, T1, a.b2n, wave * 16 * T1);
    __builtin_amdgcn_sched_barrier(0);
    if (m0 >= M) return;
    if constexpr (!ATT) {
#pragma unroll
        for (int u = 0; u < NAO; ++u) {
            const int i = tid + u * 64 * SB_NW, rr = i / (D / 4), c4 = i % (D / 4);
            *reinterpret_cast<float4 *>(bufA + rr * LDA + 4 * c4) = (m0 + rr < M) ? aov[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    } else {
        // ---- fused attention, part 2 (same mathematics as k_attn16: log2-domain scores, two passes, the IRN target
        //      column as a separate key with +1.0 where every other visible key carries +r_u)
        const int ld = 3 * D;
        const float LOG2E = 1.4426950408889634f;
        const bool irn = a.mask_mode == IRS_MASK_IRN;
        const bool tgt_ok = irn && a.seq_last[0] != 0;
        const float tgt_add = irn ? (1.0f - a.r_u[0]) * LOG2E : 0.f;
        const int qi = m0 + lq, pq = a.padq[0];
        const float sc = qi < aL ? LOG2E / sqrtf(32.0f) : 0.f;
        const float qf[8] = {aq0.x * sc, aq0.y * sc, aq0.z * sc, aq0.w * sc, aq1.x * sc, aq1.y * sc, aq1.z * sc, aq1.w * sc};
        f32x4 sacc[ATN];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < ATN; ++i) {
            if (i < ant) {
                f32x4 sa = {0.f, 0.f, 0.f, 0.f};
                sa = __builtin_amdgcn_mfma_f32_16x16x4f32(akf[i][0].x, qf[0], sa, 0, 0, 0);
                sa = __builtin_amdgcn_mfma_f32_16x16x4f32(akf[i][0].y, qf[1], sa, 0, 0, 0);
                sa = __builtin_amdgcn_mfma_f32_16x16x4f32(akf[i][0].z, qf[2], sa, 0, 0, 0);
                sa = __builtin_amdgcn_mfma_f32_16x16x4f32(akf[i][0].w, qf[3], sa, 0, 0, 0);
                sa = __builtin_amdgcn_mfma_f32_16x16x4f32(akf[i][1].x, qf[4], sa, 0, 0, 0);
                sa = __builtin_amdgcn_mfma_f32_16x16x4f32(akf[i][1].y, qf[5], sa, 0, 0, 0);
                sa = __builtin_amdgcn_mfma_f32_16x16x4f32(akf[i][1].z, qf[6], sa, 0, 0, 0);
                sa = __builtin_amdgcn_mfma_f32_16x16x4f32(akf[i][1].w, qf[7], sa, 0, 0, 0);
                sacc[i] = sa;
            }
        }
        // masks: key j visible to query qi iff j <= qi, j < L, j is not the pad the packed window may hold, and j is not
        // the target column (that one is added below); C register r of tile kt <-> key 16kt + 4gq + r
#pragma unroll
        for (int i = 0; i < ATN; ++i) {
            if (i < ant) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = 16 * (ahalf + 2 * i) + 4 * gq + r;
                    const bool ok = j <= qi && j < aL && j != pq && !(irn && j == aL - 1);
                    const float v = ok ? sacc[i][r] : -INFINITY;
                    sacc[i][r] = v;
                    mx = fmaxf(mx, v);
                }
            }
        }
        float st = -INFINITY;
        if (tgt_ok && ahalf == 0) {
            float partv = qf[0] * aktg0.x;
            partv = __fmaf_rn(qf[1], aktg0.y, partv);
            partv = __fmaf_rn(qf[2], aktg0.z, partv);
            partv = __fmaf_rn(qf[3], aktg0.w, partv);
            partv = __fmaf_rn(qf[4], aktg1.x, partv);
            partv = __fmaf_rn(qf[5], aktg1.y, partv);
            partv = __fmaf_rn(qf[6], aktg1.z, partv);
            partv = __fmaf_rn(qf[7], aktg1.w, partv);
            st = lanes_sum<48>(partv) + tgt_add;
        }
        mx = lanes_max<48>(mx);
        mx = fmaxf(mx, st);
        float *amax = bufH;                 // [4 heads][2 halves][16 queries]
        float *aobuf = bufH + 128;          // [4 heads][64 lanes][8]: the odd half's partial O^T
        float *alsum = bufH + 128 + 4 * 64 * 8; // [4 heads][16]
        if (gq == 0) amax[(ah * 2 + ahalf) * 16 + lq] = mx;
        __syncthreads();
        float m = fmaxf(amax[(ah * 2) * 16 + lq], amax[(ah * 2 + 1) * 16 + lq]);
        if (m == -INFINITY) m = 0.f; // nothing visible: l = 0 -> NaN row like torch
        float l = 0.f;
        f32x4 o[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if (tgt_ok && ahalf == 0) {
            const float pt = __builtin_amdgcn_exp2f(st - m);
            l = (gq == 0) ? pt : 0.f;
            o[0] = {pt * avt0.x, pt * avt0.y, pt * avt0.z, pt * avt0.w};
            o[1] = {pt * avt1.x, pt * avt1.y, pt * avt1.z, pt * avt1.w};
        }
#pragma unroll
        for (int i = 0; i < ATN; ++i) {
            if (i < ant) {
                float pa[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pa[r] = __builtin_amdgcn_exp2f(sacc[i][r] - m);
                    l += pa[r];
                }
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(avf[i][ct][r], pa[r], o[ct], 0, 0, 0);
            }
        }
        l = lanes_sum<48>(l);
        if (ahalf == 1) {
            float *ob = aobuf + (ah * 64 + lane) * 8;
            *reinterpret_cast<float4 *>(ob) = make_float4(o[0][0], o[0][1], o[0][2], o[0][3]);
            *reinterpret_cast<float4 *>(ob + 4) = make_float4(o[1][0], o[1][1], o[1][2], o[1][3]);
            if (gq == 0) alsum[ah * 16 + lq] = l;
        }
        __syncthreads();
        if (ahalf == 0) {
            const float *ob = aobuf + (ah * 64 + lane) * 8;
            const float4 p0 = *reinterpret_cast<const float4 *>(ob), p1 = *reinterpret_cast<const float4 *>(ob + 4);
            const float inv = 1.0f / (l + alsum[ah * 16 + lq]);
            float *dst = bufA + lq * LDA + ah * 32 + 4 * gq; // O^T register r of column tile ct <-> column 16ct + 4gq + r
            *reinterpret_cast<float4 *>(dst) = make_float4((o[0][0] + p0.x) * inv, (o[0][1] + p0.y) * inv, (o[0][2] + p0.z) * inv, (o[0][3] + p0.w) * inv);
            *reinterpret_cast<float4 *>(dst + 16) = make_float4((o[1][0] + p1.x) * inv, (o[1][1] + p1.y) * inv, (o[1][2] + p1.z) * inv, (o[1][3] + p1.w) * inv);
        }
        W_LOAD(w1[0], T2, a.Wf + SMALL_WF_W1, wave * 16 * T2, D, 0);
        W_LOAD(w1[1], T2, a.Wf + SMALL_WF_W1, wave * 16 * T2, D, 1);
    }
    __syncthreads();
    auto row_total = [&](float (&v)[MT], int slot) { // sums over the 128 columns of each token (SB_NW waves x 4 k-slot lanes)
#pragma unroll
        for (int u = 0; u < MT; ++u) {
            v[u] = lanes_sum<48>(v[u]);
            if (gq == 0) part[slot][wave][16 * u + lq] = v[u];
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < MT; ++u) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < SB_NW; ++w) t += part[slot][w][16 * u + lq];
            v[u] = t;
        }
        // no second barrier: the slots strictly alternate (mean 0, variance 1), so slot s is rewritten only
        // by waves that have passed the barrier of slot 1-s, which every wave reaches after this read
    };
    const float invn = 1.0f / (float)D;
    const int n0 = wave * 16 * T1; // this wave's T1 tiles of a 128-wide output
    // z[u][t][r] = column n0 + 16t + 4gq + r of this lane's token mt + 16u
    auto layer_norm = [&](float (&z)[MT][T1][4], const float4 (&g)[T1], const float4 (&b)[T1], const float4 (&add)[T1]) {
        float mu[MT], q[MT];
#pragma unroll
        for (int u = 0; u < MT; ++u) {
            mu[u] = 0.f;
#pragma unroll
            for (int t = 0; t < T1; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) mu[u] += z[u][t][r];
        }
        row_total(mu, 0);
#pragma unroll
        for (int u = 0; u < MT; ++u) {
            mu[u] *= invn;
            q[u] = 0.f;
#pragma unroll
            for (int t = 0; t < T1; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) q[u] += (z[u][t][r] - mu[u]) * (z[u][t][r] - mu[u]);
        }
        row_total(q, 1);
#pragma unroll
        for (int u = 0; u < MT; ++u) {
            const float rstd = 1.0f / sqrtf(q[u] * invn + 1e-5f);
#pragma unroll
            for (int t = 0; t < T1; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    z[u][t][r] = (z[u][t][r] - mu[u]) * rstd * (&g[t].x)[r] + (&b[t].x)[r] + (&add[t].x)[r];
        }
    };
    auto to_lds2 = [&](const float (&z)[MT][T1][4], float *buf, int ld) {
#pragma unroll
        for (int u = 0; u < MT; ++u)
#pragma unroll
            for (int t = 0; t < T1; ++t)
                *reinterpret_cast<float4 *>(buf + (16 * u + lq) * ld + n0 + 16 * t + 4 * gq) =
                    make_float4(z[u][t][0], z[u][t][1], z[u][t][2], z[u][t][3]);
    };
    STAMP(1);
    float z[MT][T1][4];
    // ---- y = LN2(LN1(x + ao W_o^T + b_o) + c)
    {
        f32x4 acc[MT][T1];
        float4 res[MT][T1];
#pragma unroll
        for (int u = 0; u < MT; ++u)
#pragma unroll
            for (int t = 0; t < T1; ++t) {
                acc[u][t] = {0.f, 0.f, 0.f, 0.f};
                res[u][t] = live[u] ? *reinterpret_cast<const float4 *>(a.X + (int64_t)xrow[u] * D + n0 + 16 * t + 4 * gq)
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        V_LOAD(vb1, T2, a.b1, wave * 16 * T2);
        MMA_ROUND(acc, wo[0], T1, bufA, LDA, 0);
        W_LOAD(w2[0], T1, a.Wf + SMALL_WF_W2, n0, F, 0);
        MMA_ROUND(acc, wo[1], T1, bufA, LDA, 1);
        W_LOAD(w2[1], T1, a.Wf + SMALL_WF_W2, n0, F, 1);
#pragma unroll
        for (int u = 0; u < MT; ++u)
#pragma unroll
            for (int t = 0; t < T1; ++t) {
                z[u][t][0] = acc[u][t][0] + vbo[t].x + res[u][t].x;
                z[u][t][1] = acc[u][t][1] + vbo[t].y + res[u][t].y;
                z[u][t][2] = acc[u][t][2] + vbo[t].z + res[u][t].z;
                z[u][t][3] = acc[u][t][3] + vbo[t].w + res[u][t].w;
            }
    }
    STAMP(2);
    layer_norm(z, vg1, vb1n, vc);
    if (a.c) layer_norm(z, vg2, vb2n, zero2);
    STAMP(3);
    to_lds2(z, bufA, LDA); // every wave is past its last read of the ao tile (row_total's barriers)
    __syncthreads();
    STAMP(4);
    // ---- h = relu(y W1^T + b1): this wave's 64 columns (4 tiles) -> LDS
    {
        f32x4 acc[MT][T2];
#pragma unroll
        for (int u = 0; u < MT; ++u)
#pragma unroll
            for (int t = 0; t < T2; ++t) acc[u][t] = {0.f, 0.f, 0.f, 0.f};
        V_LOAD(vb2, T1, a.b2, n0);
        V_LOAD(vg3, T1, a.g3, n0);
        V_LOAD(vb3n, T1, a.b3n, n0);
        MMA_ROUND(acc, w1[0], T2, bufA, LDA, 0);
        W_LOAD(w2[2], T1, a.Wf + SMALL_WF_W2, n0, F, 2);
        W_LOAD(w2[3], T1, a.Wf + SMALL_WF_W2, n0, F, 3);
        MMA_ROUND(acc, w1[1], T2, bufA, LDA, 1);
        if constexpr (QKV) { // unconditional in the instantiation: a branch here would force vmcnt(0) at the merge
            V_LOAD(vbin, T3, a.bin, wave * 16 * T3);
            W_LOAD(wq[0], T3, a.Wfin, wave * 16 * T3, D, 0);
        }
#pragma unroll
        for (int u = 0; u < MT; ++u)
#pragma unroll
            for (int t = 0; t < T2; ++t) {
                const int n = wave * 16 * T2 + 16 * t + 4 * gq;
                *reinterpret_cast<float4 *>(bufH + (16 * u + lq) * LDH + n) =
                    make_float4(fmaxf(acc[u][t][0] + vb1[t].x, 0.f), fmaxf(acc[u][t][1] + vb1[t].y, 0.f),
                                fmaxf(acc[u][t][2] + vb1[t].z, 0.f), fmaxf(acc[u][t][3] + vb1[t].w, 0.f));
            }
    }
    __syncthreads();
    STAMP(5);
    // ---- x' = LN3(y + h W2^T + b2)
    {
        f32x4 acc[MT][T1];
#pragma unroll
        for (int u = 0; u < MT; ++u)
#pragma unroll
            for (int t = 0; t < T1; ++t) acc[u][t] = {0.f, 0.f, 0.f, 0.f};
        STAMP(9);
        MMA_ROUND(acc, w2[0], T1, bufH, LDH, 0);
        STAMP(10);
        MMA_ROUND(acc, w2[1], T1, bufH, LDH, 1);
        STAMP(11);
        if constexpr (QKV) { W_LOAD(wq[1], T3, a.Wfin, wave * 16 * T3, D, 1); }
        STAMP(12);
        MMA_ROUND(acc, w2[2], T1, bufH, LDH, 2);
        STAMP(13);
        MMA_ROUND(acc, w2[3], T1, bufH, LDH, 3);
#pragma unroll
        for (int u = 0; u < MT; ++u)
#pragma unroll
            for (int t = 0; t < T1; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) z[u][t][r] = acc[u][t][r] + (&vb2[t].x)[r] + z[u][t][r];
    }
    STAMP(6);
    layer_norm(z, vg3, vb3n, zero2);
    STAMP(7);
#pragma unroll
    for (int u = 0; u < MT; ++u)
        if (live[u]) {
#pragma unroll
            for (int t = 0; t < T1; ++t)
                *reinterpret_cast<float4 *>(a.Xo + (int64_t)(mt + 16 * u) * D + n0 + 16 * t + 4 * gq) =
                    make_float4(z[u][t][0], z[u][t][1], z[u][t][2], z[u][t][3]);
        }
    if constexpr (!QKV) return;
    // ---- qkv' = x' W_in^T + b_in: this wave's 96 columns (6 tiles)
    to_lds2(z, bufA, LDA);
    __syncthreads();
    {
        f32x4 acc[MT][T3];
#pragma unroll
        for (int u = 0; u < MT; ++u)
#pragma unroll
            for (int t = 0; t < T3; ++t) acc[u][t] = {0.f, 0.f, 0.f, 0.f};
        MMA_ROUND(acc, wq[0], T3, bufA, LDA, 0);
        MMA_ROUND(acc, wq[1], T3, bufA, LDA, 1);
#pragma unroll
        for (int u = 0; u < MT; ++u)
            if (live[u]) {
#pragma unroll
                for (int t = 0; t < T3; ++t) {
                    const int n = wave * 16 * T3 + 16 * t + 4 * gq;
                    *reinterpret_cast<float4 *>(a.QKV + (int64_t)(mt + 16 * u) * (3 * D) + n) =
                        make_float4(acc[u][t][0] + vbin[t].x, acc[u][t][1] + vbin[t].y, acc[u][t][2] + vbin[t].z,
                                    acc[u][t][3] + vbin[t].w);
                }
            }
    }
    STAMP(8);
}

// ------------------------------------------------------------------ small-batch fused layer tail, wide models (d = 256; round 4)
// k_block_small16's chain for the width of C4 / C5's decoder: one workgroup of 8 waves owns 16 tokens and runs
// out-projection + LN1 / LN2, FFN-1, FFN-2 + LN3 and the next layer's q | k | v on v_mfma_f32_16x16x4_f32, the waves
// splitting the output columns of every GEMM, activation tiles passing through LDS.  At d = 256 a wave's share of the
// weights is 48 "tile-rounds" (16 output columns x 64 k = 16 registers per lane each) -- far more than the register file
// holds at once -- so the four GEMMs run as ONE stream of tile-rounds through a ring of LA register sets: tile-round i + LA
// is requested as soon as tile-round i has been multiplied, across the LayerNorm exchanges and barriers (the weights do
// not depend on the activations).  C5's beam step decodes 32 windows (~3.5K packed tokens): ~50 per-GEMM launches per
// step before, 6 now.  Weights: fragment-packed copies (k_pack_frag16), [Wo | W1 | W2] per layer, then the [Win].
template <int D, bool QKV>
__global__ void __launch_bounds__(512) k_block_small_wide(SmallBlockArgs a) {
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    constexpr int F = 256, NWV = 8, LDA = D + 4, LDH = F + 4;
    constexpr int TO = D / 16 / NWV, TH = F / 16 / NWV, TQ = 3 * D / 16 / NWV; // 16-column tiles per wave: d-wide, F-wide, 3d-wide outputs
    constexpr int RD = D / 64, RF = F / 64;                                      // 64-k rounds of a K = d / K = F contraction
    constexpr int N0 = RD * TO, N1 = RD * TH, N2 = RF * TO, N3 = QKV ? RD * TQ : 0, NTR = N0 + N1 + N2 + N3;
    constexpr int LA = 8; // tile-rounds in flight (128 registers)
    static_assert(D % 128 == 0 && TO >= 1, "width: a multiple of 128");
    __shared__ __attribute__((aligned(16))) float bufA[16 * LDA]; // ao -> y -> x'
    __shared__ __attribute__((aligned(16))) float bufH[16 * LDH]; // h
    __shared__ float part[2][NWV][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 15, gq = lane >> 4;
    const int m0 = blockIdx.x * 16, mt = m0 + lq;
    const float *WfO = a.Wf, *WfH = a.Wf + (size_t)D * D, *Wf2 = a.Wf + (size_t)D * D + (size_t)F * D, *WfQ = a.Wfin;
    // tile-round i of this wave: (phase, round r, tile t), rounds outermost inside a phase (a round's B fragments serve its tiles)
    float4 w[LA][4];
    auto tr_load = [&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        if constexpr (i < NTR) {
            constexpr int ph = i < N0 ? 0 : i < N0 + N1 ? 1 : i < N0 + N1 + N2 ? 2 : 3;
            constexpr int li = i - (ph == 0 ? 0 : ph == 1 ? N0 : ph == 2 ? N0 + N1 : N0 + N1 + N2);
            constexpr int T = ph == 0 ? TO : ph == 1 ? TH : ph == 2 ? TO : TQ, K = ph == 2 ? F : D;
            constexpr int r = li / T, t = li % T;
            const float *W = ph == 0 ? WfO : ph == 1 ? WfH : ph == 2 ? Wf2 : WfQ;
            const float *wfr = W + ((size_t)((wave * T + t) * (K / 64) + r) * 4) * 256 + lane * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) w[i % LA][j] = *reinterpret_cast<const float4 *>(wfr + 256 * j);
        }
    };
    // the activation tile and the residual rows first (a wave's loads return in order), then the first LA tile-rounds
    constexpr int NAO = 16 * (D / 4) / (64 * NWV);
    float4 aov[NAO];
#pragma unroll
    for (int u = 0; u < NAO; ++u) {
        const int i = tid + u * 64 * NWV, rr = i / (D / 4), c4 = i % (D / 4);
        aov[u] = *reinterpret_cast<const float4 *>(a.AO + (int64_t)min(m0 + rr, a.M - 1) * D + 4 * c4);
    }
    const int xrow = a.xidx ? a.xidx[min(mt, a.M - 1)] : min(mt, a.M - 1);
    const int nO = wave * 16 * TO; // this wave's first column of a d-wide output
    float4 res[TO];
#pragma unroll
    for (int t = 0; t < TO; ++t) res[t] = *reinterpret_cast<const float4 *>(a.X + (int64_t)xrow * D + nO + 16 * t + 4 * gq);
    __builtin_amdgcn_sched_barrier(0);
    x6_static_for<0, LA>(tr_load);
    __builtin_amdgcn_sched_barrier(0);
    const int M = a.m_dev ? min(a.M, a.m_dev[0]) : a.M;
    if (m0 >= M) return;
    const bool live = mt < M;
    auto vec4 = [&](const float *p, int n) { return p ? *reinterpret_cast<const float4 *>(p + n) : make_float4(0.f, 0.f, 0.f, 0.f); };
#pragma unroll
    for (int u = 0; u < NAO; ++u) {
        const int i = tid + u * 64 * NWV, rr = i / (D / 4), c4 = i % (D / 4);
        *reinterpret_cast<float4 *>(bufA + rr * LDA + 4 * c4) = (m0 + rr < M) ? aov[u] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    auto row_total = [&](float v, int slot) { // sum over the d columns of this lane's token (8 waves x 4 k-slot lanes)
        v = lanes_sum<48>(v);
        if (gq == 0) part[slot][wave][lq] = v;
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int wv = 0; wv < NWV; ++wv) t += part[slot][wv][lq];
        return t; // (no second barrier: the two slots strictly alternate, see k_block_small16)
    };
    const float invn = 1.0f / (float)D;
    auto layer_norm = [&](float (&z)[TO][4], const float *g, const float *b, const float *add) {
        float mu = 0.f;
#pragma unroll
        for (int t = 0; t < TO; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) mu += z[t][r];
        mu = row_total(mu, 0) * invn;
        float q = 0.f;
#pragma unroll
        for (int t = 0; t < TO; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) q += (z[t][r] - mu) * (z[t][r] - mu);
        const float rstd = 1.0f / sqrtf(row_total(q, 1) * invn + 1e-5f);
#pragma unroll
        for (int t = 0; t < TO; ++t) {
            const float4 gg = vec4(g, nO + 16 * t + 4 * gq), bb = vec4(b, nO + 16 * t + 4 * gq), ad = vec4(add, nO + 16 * t + 4 * gq);
#pragma unroll
            for (int r = 0; r < 4; ++r) z[t][r] = (z[t][r] - mu) * rstd * (&gg.x)[r] + (&bb.x)[r] + (&ad.x)[r];
        }
    };
    auto to_lds = [&](const float (&z)[TO][4], float *buf, int ld) {
#pragma unroll
        for (int t = 0; t < TO; ++t)
            *reinterpret_cast<float4 *>(buf + lq * ld + nO + 16 * t + 4 * gq) = make_float4(z[t][0], z[t][1], z[t][2], z[t][3]);
    };
    // one phase: acc[t] += W-tile-rounds . B rounds; tile-round indices I0 .. I0 + R T - 1
    f32x4 accO[TO], accH[TH], accQ[QKV ? TQ : 1];
    auto run_phase = [&](auto i0c, auto rc, auto tc, auto &acc, const float *B, int ldb) __attribute__((always_inline)) {
        constexpr int I0 = decltype(i0c)::value, R = decltype(rc)::value, T = decltype(tc)::value;
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = {0.f, 0.f, 0.f, 0.f};
        x6_static_for<0, R>([&](auto rr) __attribute__((always_inline)) {
            constexpr int r = decltype(rr)::value;
            float4 bf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const float4 *>(B + lq * ldb + 4 * gq + 64 * r + 16 * j);
            x6_static_for<0, T>([&](auto tt) __attribute__((always_inline)) {
                constexpr int t = decltype(tt)::value, i = I0 + r * T + t;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i % LA][j].x, bf[j].x, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i % LA][j].y, bf[j].y, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i % LA][j].z, bf[j].z, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i % LA][j].w, bf[j].w, acc[t], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                tr_load(std::integral_constant<int, i + LA>{}); // the set just multiplied is free again
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    };
    using IC = std::integral_constant<int, 0>;
    float z[TO][4];
    // ---- y = LN2(LN1(x + ao W_o^T + b_o) + c)
    run_phase(IC{}, std::integral_constant<int, RD>{}, std::integral_constant<int, TO>{}, accO, bufA, LDA);
#pragma unroll
    for (int t = 0; t < TO; ++t) {
        const float4 bo = vec4(a.bo, nO + 16 * t + 4 * gq);
        const float4 rs = live ? res[t] : make_float4(0.f, 0.f, 0.f, 0.f);
        z[t][0] = accO[t][0] + bo.x + rs.x, z[t][1] = accO[t][1] + bo.y + rs.y, z[t][2] = accO[t][2] + bo.z + rs.z, z[t][3] = accO[t][3] + bo.w + rs.w;
    }
    layer_norm(z, a.g1, a.b1n, a.c);
    if (a.c) layer_norm(z, a.g2, a.b2n, nullptr);
    to_lds(z, bufA, LDA); // (every wave is past its last read of the ao tile: row_total's barriers)
    __syncthreads();
    // ---- h = relu(y W1^T + b1)
    run_phase(std::integral_constant<int, N0>{}, std::integral_constant<int, RD>{}, std::integral_constant<int, TH>{}, accH, bufA, LDA);
#pragma unroll
    for (int t = 0; t < TH; ++t) {
        const int n = wave * 16 * TH + 16 * t + 4 * gq;
        const float4 b1 = vec4(a.b1, n);
        *reinterpret_cast<float4 *>(bufH + lq * LDH + n) = make_float4(fmaxf(accH[t][0] + b1.x, 0.f), fmaxf(accH[t][1] + b1.y, 0.f),
                                                                       fmaxf(accH[t][2] + b1.z, 0.f), fmaxf(accH[t][3] + b1.w, 0.f));
    }
    __syncthreads();
    // ---- x' = LN3(y + h W2^T + b2)
    run_phase(std::integral_constant<int, N0 + N1>{}, std::integral_constant<int, RF>{}, std::integral_constant<int, TO>{}, accO, bufH, LDH);
#pragma unroll
    for (int t = 0; t < TO; ++t) {
        const float4 b2 = vec4(a.b2, nO + 16 * t + 4 * gq);
#pragma unroll
        for (int r = 0; r < 4; ++r) z[t][r] = accO[t][r] + (&b2.x)[r] + z[t][r];
    }
    layer_norm(z, a.g3, a.b3n, nullptr);
    if (live) {
#pragma unroll
        for (int t = 0; t < TO; ++t)
            *reinterpret_cast<float4 *>(a.Xo + (int64_t)mt * D + nO + 16 * t + 4 * gq) = make_float4(z[t][0], z[t][1], z[t][2], z[t][3]);
    }
    if constexpr (QKV) {
        // ---- qkv' = x' W_in^T + b_in
        to_lds(z, bufA, LDA); // (LN3's barriers: every wave is past FFN-1's reads of the y tile)
        __syncthreads();
        run_phase(std::integral_constant<int, N0 + N1 + N2>{}, std::integral_constant<int, RD>{}, std::integral_constant<int, TQ>{}, accQ, bufA, LDA);
        if (live) {
#pragma unroll
            for (int t = 0; t < TQ; ++t) {
                const int n = wave * 16 * TQ + 16 * t + 4 * gq;
                const float4 bi = vec4(a.bin, n);
                *reinterpret_cast<float4 *>(a.QKV + (int64_t)mt * (3 * D) + n) =
                    make_float4(accQ[t][0] + bi.x, accQ[t][1] + bi.y, accQ[t][2] + bi.z, accQ[t][3] + bi.w);
            }
        }
    }
}
static inline size_t small_wide_layer_floats(int d, int F) { return (size_t)d * d + 2 * (size_t)F * d; } // Wo | W1 | W2
static inline bool small_wide_shape(int d, int F) { return d == 256 && F == 256; }

// arguments of the embedding + layer-0 in-projection kernels of the latency paths
struct SmallEmbedArgs {
    const int64_t *seq;
    const float *E, *pe;
    const int32_t *tok_row, *m_dev;
    int rows, L;
    float sqrtd;
    int64_t n_item;
    float *X;               // [rows][128]
    const float *Wfin, *bin; // fragment-packed W_in of layer 0, its bias
    float *QKV;             // [rows][384]
    // packing plan of a single sequence computed in this kernel (k_embed_qkv_small16<true>; see k_plan_small)
    const int32_t *pos;
    int32_t *cnt, *off, *qrow, *tok_out, *padq, *mdev_out;
    const int64_t *user;
    const float *U, *uw, *ub;
    float *r_u;
    int ud;
    int64_t n_user;
    int32_t *step_pair;
};

// Packing plan of ONE sequence of L <= 256 tokens inside a 256-thread workgroup (see k_plan_small): thread t looks at
// token t; s_tok[i] / s_id[i] = window position / clamped item id of packed row i; returns the packed row count.
// Workgroup 0 also writes the plan arrays, r_u and the step hand-over for the kernels that follow.
__device__ __forceinline__ int small_plan_single(const SmallEmbedArgs &a, int *s_tok, int *s_id, int *s_wcnt, int *s_q) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int p = a.pos[0];
    p = p < 0 ? 0 : (p >= a.L ? a.L - 1 : p);
    const int64_t sv = tid < a.L ? a.seq[tid] : 0;
    const bool v = tid < a.L && (sv != 0 || tid == p); // plan_valid
    const unsigned long long bm = __ballot(v);
    if (lane == 0) s_wcnt[wave] = __popcll(bm);
    __syncthreads();
    int basei = 0;
    for (int w = 0; w < wave; ++w) basei += s_wcnt[w];
    const int M = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
    if (v) {
        const int idx = basei + __popcll(bm & ((1ull << lane) - 1ull));
        s_tok[idx] = tid;
        s_id[idx] = (int)(sv < 0 ? 0 : (sv > a.n_item ? a.n_item : sv));
        if (tid == p) {
            s_q[0] = idx;
            s_q[1] = (sv == 0) ? idx : -1; // the only pad a packed sequence can hold
        }
        if (blockIdx.x == 0) a.tok_out[idx] = tid;
    }
    __syncthreads();
    if (blockIdx.x == 0 && tid == 0) {
        a.cnt[0] = M, a.off[0] = 0, a.qrow[0] = s_q[0], a.padq[0] = s_q[1], a.mdev_out[0] = M;
        float acc = 0.f; // r_u = user_mask_layer(user_embedder(user)) (influentialRS.py:180), 0 without the user factor
        if (a.U) {
            int64_t u = a.user[0];
            if (u < 0) u = 0;
            if (u >= a.n_user) u = a.n_user - 1;
            const float *e = a.U + u * (int64_t)a.ud;
            for (int c = 0; c < a.ud; ++c) acc = __fmaf_rn(e[c], a.uw[c], acc);
            acc += a.ub[0];
        }
        a.r_u[0] = acc;
        if (a.step_pair) a.step_pair[0] = a.step_pair[1];
    }
    return M;
}

// Zero-padded fragment-packed copy of W[N][K] for k_block_small_any: [Np/16 tiles][Kp/16 k groups][64 lanes] float4.
__global__ void k_pack_frag16_any(const float *__restrict__ W, float *__restrict__ out, int N, int K, int Np, int Kp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x; // one float4 of the packed image
    if (i >= Np * Kp / 4) return;
    const int lane = i & 63, rest = i >> 6, KG = Kp >> 4;
    const int kg = rest % KG, nt = rest / KG;
    const int n = 16 * nt + (lane & 15), k = 16 * kg + 4 * (lane >> 4);
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (n < N && k + e < K) ? W[(size_t)n * K + k + e] : 0.f;
    reinterpret_cast<float4 *>(out)[i] = make_float4(v[0], v[1], v[2], v[3]);
}
// per-layer image: [Wo dp x dp | W1 Fp x dp | W2 dp x Fp | Win Qp x dp]
static inline bool small_any_shape(int d, int F) {
    const int Fp = (F + 15) & ~15, ksh = Fp > 128 ? 2 : Fp > 64 ? 1 : 0; // the k groups of W2 must split evenly into its chunks
    return !(d == 128 && F == 256) && d <= 96 && F <= 256 && ((Fp >> 4) & ((1 << ksh) - 1)) == 0;
}
static inline int small_any_dp(int d) { return d <= 32 ? 32 : d <= 64 ? 64 : 96; } // padded width: one of three instantiations
static inline size_t small_any_win_off(int d, int F) {
    const size_t dp = small_any_dp(d), Fp = (F + 15) & ~15;
    return dp * dp + 2 * Fp * dp;
}
static inline size_t small_any_layer_floats(int d, int F) {
    const size_t dp = small_any_dp(d), Qp = (3 * d + 15) & ~15;
    return small_any_win_off(d, F) + Qp * dp;
}

// The same layer tail for ANY small shape (d <= 96, F <= 256: the reference's CLI default d = 30, config 1's d = 64,
// the evaluator's d = 30 / F = 120), where a step is launch bound (four GEMM launches per layer for ~1 MFLOP of
// work).  16 tokens per workgroup, 4 waves, v_mfma_f32_16x16x4_f32 with zero-padded tiles: the weights come from
// zero-padded fragment-packed copies (row-major dword fragments measured 2.5x slower: 16 cache lines per wave load),
// activations sit in LDS padded to 16 columns; work items = (16-column output tile, k chunk) dealt round-robin to the waves -- the FFN's second GEMM
// (K = F) is split into up to 4 k chunks whose partial tiles are summed in a fixed order -- and the LayerNorms run on
// the LDS tile with 16 lanes per token.
#ifdef IRS_SMALL_TIMING
#define ASTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_small_t[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ASTAMP(i)
#endif
// DPT = padded width / 16 (2, 4 or 6) is a template parameter so that the fragment arrays have static shapes and the
// k loops of the three K = d GEMMs carry no run-time guards (the fully run-time version spent ~600 scalar branches).
template <int MAXI, int MAXG>
__device__ __forceinline__ void any_load(const float4 *W, int KG, int ksh, int gpi, int nit, int wave, int lane,
                                         float4 (&av)[MAXI][MAXG]) {
#pragma unroll
    for (int ii = 0; ii < MAXI; ++ii) {
        const int item = wave + 4 * ii, nt = item >> ksh, kc = item & ((1 << ksh) - 1);
        const float4 *wp = W + ((size_t)nt * KG + kc * gpi) * 64 + lane;
        if (ii < nit) { // wave-uniform
#pragma unroll
            for (int kg = 0; kg < MAXG; ++kg)
                if (kg < gpi) av[ii][kg] = wp[kg * 64];
        }
    }
}
template <int MAXI, int MAXG, class Emit>
__device__ __forceinline__ void any_gemm(int ksh, int gpi, int nit, int wave, const float4 (&av)[MAXI][MAXG], const float *brow,
                                         Emit emit) {
    typedef __attribute__((ext_vector_type(4))) float f32x4;
#pragma unroll
    for (int ii = 0; ii < MAXI; ++ii) {
        if (ii < nit) {
            const int item = wave + 4 * ii, nt = item >> ksh, kc = item & ((1 << ksh) - 1);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kg = 0; kg < MAXG; ++kg) {
                if (kg < gpi) {
                    const float4 bv = *reinterpret_cast<const float4 *>(brow + 16 * (kc * gpi + kg));
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ii][kg].x, bv.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ii][kg].y, bv.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ii][kg].z, bv.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ii][kg].w, bv.w, acc, 0, 0, 0);
                }
            }
            emit(nt, kc, acc);
        }
    }
}

template <bool QKV, int DPT>
__global__ void __launch_bounds__(256) k_block_small_any(SmallBlockArgs a, int d, int F) {
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    extern __shared__ __attribute__((aligned(16))) float sm_any[];
    constexpr int dp = 16 * DPT;
    const int Fp = (F + 15) & ~15, lda = dp + 4, ldh = Fp + 4;
    float *bufA = sm_any;            // [16][lda] GEMM input: ao -> y -> x'
    float *bufZ = bufA + 16 * lda;   // [16][lda] x + b_o (+ GEMM) -> y ; y + b2 (+ GEMM) -> x'
    float *bufH = bufZ + 16 * lda;   // [16][ldh] h
    float *bufP = bufH + 16 * ldh;   // [4][16][lda] partial tiles of the k-split GEMM
    // every per-column vector, staged once with the activation tile (in-kernel timing: read from global memory where
    // they are used, each LayerNorm / epilogue paid its own ~1 us round trip): 9 x [d] | b1 [F] | b_in [3d]
    float *vecs = bufP + 4 * 16 * lda;
    const float *vbo = vecs, *vg1 = vecs + d, *vb1n = vecs + 2 * d, *vc = vecs + 3 * d, *vg2 = vecs + 4 * d, *vb2n = vecs + 5 * d;
    const float *vb2 = vecs + 6 * d, *vg3 = vecs + 7 * d, *vb3n = vecs + 8 * d, *vb1 = vecs + 9 * d, *vbin = vecs + 9 * d + F;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 15, gq = lane >> 4;
    ASTAMP(0);
    const int M = a.m_dev ? min(a.M, a.m_dev[0]) : a.M;
    const int m0 = blockIdx.x * 16;
    if (m0 >= M) return;
    const int KSF = Fp > 128 ? 2 : Fp > 64 ? 1 : 0, KCF = 1 << KSF; // k chunks of the FFN's second GEMM: 4 / 2 / 1
    {
        // every global load of the fill is requested before the first LDS store: one round trip (two for the rows
        // behind a.xidx), not one per loop
        constexpr int NV = 4, NQ = 2; // 9 d <= 864, 3 d <= 288 values over 256 threads
        const float *src[9] = {a.bo, a.g1, a.b1n, a.c, a.g2, a.b2n, a.b2, a.g3, a.b3n};
        float fv[NV], fb1 = 0.f, fq[NQ], fao[DPT], fx[DPT];
        int xr[DPT];
#pragma unroll
        for (int u = 0; u < DPT; ++u) {
            const int i = tid + 256 * u, row = m0 + i / dp;
            xr[u] = (row < M && a.xidx) ? a.xidx[row] : row;
        }
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int i = tid + 256 * u, v = i / d;
            const float *p = src[0];
#pragma unroll
            for (int w = 1; w < 9; ++w) p = (v == w) ? src[w] : p;
            fv[u] = (i < 9 * d && p) ? p[i - v * d] : 0.f;
        }
        if (tid < F) fb1 = a.b1[tid];
#pragma unroll
        for (int u = 0; u < NQ; ++u) fq[u] = (QKV && tid + 256 * u < 3 * d) ? a.bin[tid + 256 * u] : 0.f;
#pragma unroll
        for (int u = 0; u < DPT; ++u) {
            const int i = tid + 256 * u, rr = i / dp, c = i - rr * dp, row = m0 + rr;
            const bool ok = row < M && c < d;
            fao[u] = ok ? a.AO[(int64_t)row * d + c] : 0.f;
            fx[u] = ok ? a.X[(int64_t)xr[u] * d + c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < NV; ++u)
            if (tid + 256 * u < 9 * d) vecs[tid + 256 * u] = fv[u];
        if (tid < F) vecs[9 * d + tid] = fb1;
        if constexpr (QKV) {
#pragma unroll
            for (int u = 0; u < NQ; ++u)
                if (tid + 256 * u < 3 * d) vecs[9 * d + F + tid + 256 * u] = fq[u];
        }
#pragma unroll
        for (int u = 0; u < DPT; ++u) {
            const int i = tid + 256 * u, rr = i / dp, c = i - rr * dp;
            bufA[rr * lda + c] = fao[u];
            bufZ[rr * lda + c] = fx[u]; // + b_o below, once the vectors are visible
        }
    }
    // acc(item) = W[16 nt .. +15][k range of chunk kc] . B[token][same k] on zero-padded, fragment-packed weights
    // (k_pack_frag16_any: float4 ((nt * Kp/16 + kg) * 64 + lane) = W[16nt + lq][16kg + 4gq .. +3], so a wave load is
    // 1 KB contiguous and needs no guards); B = buf[lq][same k] as one ds_read_b128; C register r <-> column
    // 16nt + 4gq + r of token lq.  A wave owns items wave, wave + 4, ... (item = nt * KC + kc, KC a power of two).
    // ALL the weight fragments of a wave's share of a GEMM are requested at once, before the barrier / LayerNorm in
    // front of the GEMM: weights do not depend on the activations.
    const float4 *Wo4 = reinterpret_cast<const float4 *>(a.Wf), *W14 = Wo4 + dp * dp / 4, *W24 = W14 + Fp * dp / 4;
    const float4 *Wq4 = reinterpret_cast<const float4 *>(a.Wfin);
    auto items_of = [&](int items) { return items > wave ? (items - wave + 3) >> 2 : 0; };
    constexpr int IO = (DPT + 3) / 4, I1 = 4, I2 = DPT, IQ = (3 * DPT + 3) / 4; // static bounds of a wave's items
    const int nit_o = items_of(DPT), nit_1 = items_of(Fp >> 4), nit_2 = items_of(DPT << KSF), gpi_2 = (Fp >> 4) >> KSF;
    const int nit_q = items_of((3 * d + 15) >> 4);
    const float *browA = bufA + lq * lda + 4 * gq, *browH = bufH + lq * ldh + 4 * gq;
    // LayerNorm of the 16 rows of bufZ in place (16 lanes per token: columns sub, sub + 16, ...)
    const int tk = tid >> 4, sub = tid & 15;
    auto group_sum = [&](float v) {
        return lanes_sum<15>(v);
    };
    auto layer_norm = [&](const float *g, const float *b, const float *add) {
        float *zr = bufZ + tk * lda;
        float s1 = 0.f;
        for (int c = sub; c < d; c += 16) s1 += zr[c];
        const float mu = group_sum(s1) / (float)d;
        float q = 0.f;
        for (int c = sub; c < d; c += 16) q += (zr[c] - mu) * (zr[c] - mu);
        const float rstd = 1.0f / sqrtf(group_sum(q) / (float)d + 1e-5f);
        for (int c = sub; c < d; c += 16) zr[c] = (zr[c] - mu) * rstd * g[c] + b[c] + (add ? add[c] : 0.f);
    };
    // ---- y = LN2(LN1(x + ao W_o^T + b_o) + c)
    float4 avo[IO][DPT], av1[I1][DPT], av2[I2][4], avq[IQ][DPT];
    any_load<IO, DPT>(Wo4, DPT, 0, DPT, nit_o, wave, lane, avo);
    __syncthreads(); // the tile fill above
    ASTAMP(1);
    any_gemm<IO, DPT>(0, DPT, nit_o, wave, avo, browA, [&](int nt, int, const f32x4 &acc) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = nt * 16 + 4 * gq + r;
            if (n < d) bufZ[lq * lda + n] += acc[r] + vbo[n];
        }
    });
    ASTAMP(2);
    any_load<I1, DPT>(W14, DPT, 0, DPT, nit_1, wave, lane, av1);
    __syncthreads();
    layer_norm(vg1, vb1n, vc);
    if (a.c) layer_norm(vg2, vb2n, nullptr);
    for (int c = sub; c < d; c += 16) bufA[tk * lda + c] = bufZ[tk * lda + c]; // columns d .. dp stay zero
    __syncthreads();
    ASTAMP(3);
    // ---- h = relu(y W1^T + b1)
    any_gemm<I1, DPT>(0, DPT, nit_1, wave, av1, browA, [&](int nt, int, const f32x4 &acc) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = nt * 16 + 4 * gq + r;
            bufH[lq * ldh + n] = n < F ? fmaxf(acc[r] + vb1[n], 0.f) : 0.f;
        }
    });
    ASTAMP(4);
    any_load<I2, 4>(W24, Fp >> 4, KSF, gpi_2, nit_2, wave, lane, av2);
    __syncthreads();
    // ---- x' = LN3(y + h W2^T + b2): k-split partial tiles, then a fixed-order sum
    any_gemm<I2, 4>(KSF, gpi_2, nit_2, wave, av2, browH, [&](int nt, int kc, const f32x4 &acc) {
#pragma unroll
        for (int r = 0; r < 4; ++r) bufP[(kc * 16 + lq) * lda + nt * 16 + 4 * gq + r] = acc[r];
    });
    ASTAMP(5);
    if constexpr (QKV) any_load<IQ, DPT>(Wq4, DPT, 0, DPT, nit_q, wave, lane, avq);
    __syncthreads();
    for (int c = sub; c < d; c += 16) {
        float v = bufZ[tk * lda + c] + vb2[c];
        for (int kc = 0; kc < KCF; ++kc) v += bufP[(kc * 16 + tk) * lda + c];
        bufZ[tk * lda + c] = v;
    }
    ASTAMP(6);
    layer_norm(vg3, vb3n, nullptr);
    ASTAMP(7);
    {
        const int row = m0 + tk;
        for (int c = sub; c < d; c += 16) {
            const float v = bufZ[tk * lda + c];
            bufA[tk * lda + c] = v;
            if (row < M) a.Xo[(int64_t)row * d + c] = v;
        }
    }
    if constexpr (!QKV) return;
    __syncthreads();
    ASTAMP(8);
    // ---- qkv' = x' W_in^T + b_in
    any_gemm<IQ, DPT>(0, DPT, nit_q, wave, avq, browA, [&](int nt, int, const f32x4 &acc) {
        const int row = m0 + lq;
        if (row < M) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = nt * 16 + 4 * gq + r;
                if (n < 3 * d) a.QKV[(int64_t)row * (3 * d) + n] = acc[r] + vbin[n];
            }
        }
    });
    ASTAMP(9);
}
// Embedding + layer 0's in-projection for the same shapes: x = E[seq] sqrt(d) + pe to global memory and to the
// LDS tile, then the QKV GEMM on the zero-padded packed W_in of layer 0 (one launch instead of two).
template <int DPT, bool PLAN>
__global__ void __launch_bounds__(256) k_embed_qkv_small_any(SmallEmbedArgs a, int d) {
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    constexpr int dp = 16 * DPT, lda = dp + 4, IQ = (3 * DPT + 3) / 4;
    __shared__ __attribute__((aligned(16))) float bufA[16 * lda];
    __shared__ float vbin[3 * dp];
    __shared__ int s_tok[256], s_id[256], s_wcnt[4], s_q[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 15, gq = lane >> 4;
    const int Qt = (3 * d + 15) >> 4; // 16-column tiles of the QKV output
    const int nit_q = Qt > wave ? (Qt - wave + 3) >> 2 : 0;
    float4 avq[IQ][DPT];
    any_load<IQ, DPT>(reinterpret_cast<const float4 *>(a.Wfin), DPT, 0, DPT, nit_q, wave, lane, avq);
    int M;
    if constexpr (PLAN) M = small_plan_single(a, s_tok, s_id, s_wcnt, s_q);
    else M = a.m_dev ? min(a.rows, a.m_dev[0]) : a.rows;
    const int m0 = blockIdx.x * 16;
    if (m0 >= M) return;
    for (int i = tid; i < 3 * d; i += 256) vbin[i] = a.bin[i];
    for (int i = tid; i < 16 * dp; i += 256) {
        const int rr = i / dp, c = i - rr * dp, row = m0 + rr;
        float v = 0.f;
        if (row < M && c < d) {
            const int orig = PLAN ? s_tok[row] : (a.tok_row ? a.tok_row[row] : row);
            int64_t id = PLAN ? (int64_t)s_id[row] : a.seq[orig];
            if (id < 0) id = 0;
            if (id > a.n_item) id = a.n_item;
            v = __fadd_rn(__fmul_rn(a.E[id * (int64_t)d + c], a.sqrtd), a.pe[(int64_t)(orig % a.L) * d + c]);
            a.X[(int64_t)row * d + c] = v;
        }
        bufA[rr * lda + c] = v;
    }
    __syncthreads();
    any_gemm<IQ, DPT>(0, DPT, nit_q, wave, avq, bufA + lq * lda + 4 * gq, [&](int nt, int, const f32x4 &acc) {
        const int row = m0 + lq;
        if (row < M) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = nt * 16 + 4 * gq + r;
                if (n < 3 * d) a.QKV[(int64_t)row * (3 * d) + n] = acc[r] + vbin[n];
            }
        }
    });
}
static inline size_t small_any_lds(int d, int F);
static void launch_small_any(bool qkv, int rows, int d, int F, const SmallBlockArgs &sb, hipStream_t s) {
    const dim3 grid((rows + 15) / 16);
    const size_t lds = small_any_lds(d, F);
#define A_(Q_, T_) hipLaunchKernelGGL((k_block_small_any<Q_, T_>), grid, dim3(256), lds, s, sb, d, F)
    switch (small_any_dp(d)) {
    case 32: if (qkv) A_(true, 2); else A_(false, 2); break;
    case 64: if (qkv) A_(true, 4); else A_(false, 4); break;
    default: if (qkv) A_(true, 6); else A_(false, 6); break;
    }
#undef A_
}
static inline size_t small_any_lds(int d, int F) {
    const int dp = small_any_dp(d), Fp = (F + 15) & ~15;
    return (size_t)(6 * 16 * (dp + 4) + 16 * (Fp + 4) + 12 * d + F) * sizeof(float);
}

// Embedding + layer 0's in-projection of the latency path, 16 packed tokens per workgroup: x = E[seq] sqrt(d) + pe
// goes to global memory (the residual of layer 0) and to LDS (the B operand); the in-projection runs on the
// fragment-packed W_in of layer 0, all 12 tile-rounds of a wave requested before the embedding rows are.

// PLAN = true (one sequence of L <= 256 tokens): every workgroup derives the packing plan itself -- thread t looks
// at token t, ballots and a four-entry scan give the packed index -- instead of reading it from a k_plan_small launch
// in front; workgroup 0 also writes the plan arrays, r_u and the step hand-over for the kernels that follow.
template <bool PLAN>
__global__ void __launch_bounds__(256) k_embed_qkv_small16(SmallEmbedArgs a) {
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    constexpr int D = 128, LDA = D + 4, MT = 1;
    __shared__ __attribute__((aligned(16))) float bufA[16 * LDA];
    __shared__ int s_tok[256], s_id[256], s_wcnt[4], s_q[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 15, gq = lane >> 4;
    float4 wq[2][6][4], vbin[6];
    W_LOAD(wq[0], 6, a.Wfin, wave * 96, D, 0);
    W_LOAD(wq[1], 6, a.Wfin, wave * 96, D, 1);
#pragma unroll
    for (int t = 0; t < 6; ++t) vbin[t] = *reinterpret_cast<const float4 *>(a.bin + wave * 96 + 16 * t + 4 * gq);
    __builtin_amdgcn_sched_barrier(0);
    int M;
    if constexpr (PLAN) {
        M = small_plan_single(a, s_tok, s_id, s_wcnt, s_q);
    } else
        M = a.m_dev ? min(a.rows, a.m_dev[0]) : a.rows;
    const int m0 = blockIdx.x * 16;
    if (m0 >= M) return;
    {
        const int rr = tid >> 4, c8 = tid & 15, row = m0 + rr;
        float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
        if (row < M) {
            const int orig = PLAN ? s_tok[row] : (a.tok_row ? a.tok_row[row] : row);
            int64_t id = PLAN ? (int64_t)s_id[row] : a.seq[orig];
            if (id < 0) id = 0;
            if (id > a.n_item) id = a.n_item;
            const float *e = a.E + id * (int64_t)D + 8 * c8;
            const float *pp = a.pe + (int64_t)(orig % a.L) * D + 8 * c8;
            const float4 e0 = *reinterpret_cast<const float4 *>(e), e1 = *reinterpret_cast<const float4 *>(e + 4);
            const float4 p0 = *reinterpret_cast<const float4 *>(pp), p1 = *reinterpret_cast<const float4 *>(pp + 4);
            v0 = make_float4(__fadd_rn(__fmul_rn(e0.x, a.sqrtd), p0.x), __fadd_rn(__fmul_rn(e0.y, a.sqrtd), p0.y),
                             __fadd_rn(__fmul_rn(e0.z, a.sqrtd), p0.z), __fadd_rn(__fmul_rn(e0.w, a.sqrtd), p0.w));
            v1 = make_float4(__fadd_rn(__fmul_rn(e1.x, a.sqrtd), p1.x), __fadd_rn(__fmul_rn(e1.y, a.sqrtd), p1.y),
                             __fadd_rn(__fmul_rn(e1.z, a.sqrtd), p1.z), __fadd_rn(__fmul_rn(e1.w, a.sqrtd), p1.w));
            *reinterpret_cast<float4 *>(a.X + (int64_t)row * D + 8 * c8) = v0;
            *reinterpret_cast<float4 *>(a.X + (int64_t)row * D + 8 * c8 + 4) = v1;
        }
        *reinterpret_cast<float4 *>(bufA + rr * LDA + 8 * c8) = v0;
        *reinterpret_cast<float4 *>(bufA + rr * LDA + 8 * c8 + 4) = v1;
    }
    __syncthreads();
    f32x4 acc[1][6] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f},
                        {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
    MMA_ROUND(acc, wq[0], 6, bufA, LDA, 0);
    MMA_ROUND(acc, wq[1], 6, bufA, LDA, 1);
    const int mt = m0 + lq;
    if (mt < M) {
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const int n = wave * 96 + 16 * t + 4 * gq;
            *reinterpret_cast<float4 *>(a.QKV + (int64_t)mt * (3 * D) + n) =
                make_float4(acc[0][t][0] + vbin[t].x, acc[0][t][1] + vbin[t].y, acc[0][t][2] + vbin[t].z, acc[0][t][3] + vbin[t].w);
        }
    }
}
#undef W_LOAD
#undef V_LOAD
#undef MMA_ROUND

// ------------------------------------------------------------------ attention on fp32 MFMA
// One workgroup per (head, sequence), 4 waves; K_h / V_h of the sequence in LDS.
// A wave owns 32-query blocks.  Per (query block, key block):
//   S^T[key][q] = K . Q^T            HDP/2 x v_mfma_f32_32x32x2_f32  (lane = query, regs = 16 keys)
//   online softmax on the 16 registers (row max/sum are register-local; the two
//   lane halves of a query are combined with one xor-32 shuffle)
//   O^T[c][q] += V^T . P^T           16 MFMAs per 32 head columns; the B operand IS the S^T
//   accumulator (register t of lane (q,kk) is key (t&3)+8(t>>2)+4kk of query q), so P never
//   leaves registers.
// MFMA k-slots of S^T: lane half kk supplies head column kk*HDP/2 + s in step s (contiguous
// per lane -> float4 loads of Q).  Causal structure: key blocks kb <= qb only; a per-block
// bitmask of masked keys (pads, keys >= L, the IRN target) lets fully masked blocks be skipped
// and clean blocks take a mask-free path.  The IRN mask's target column (key L-1, +1.0,
// visible to every query) seeds the online softmax with one VALU dot product per query.
template <int HDP, bool V4>
__global__ void __launch_bounds__(256) k_attn_mfma(const float *__restrict__ qkv, const int64_t *__restrict__ seq,
                                                   const float *__restrict__ r_u, float *__restrict__ out, int Lmax,
                                                   int d, int hd, int mask_mode, const int32_t *__restrict__ off,
                                                   const int32_t *__restrict__ cnt, const int32_t *__restrict__ tok_row) {
    constexpr int HH = HDP / 2;               // MFMA k-steps of S^T; head columns per lane half
    constexpr int VW = HDP < 32 ? 32 : HDP;   // V columns kept in LDS (zero padded)
    constexpr int CT = VW / 32;               // 32-column tiles of O^T
    constexpr int KLD = HDP + 1;              // padded K row stride (conflict-free column reads)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // packed decode: sequence b owns rows [off[b], off[b] + cnt[b]) of the packed token list (order preserving)
    const int L = cnt ? cnt[blockIdx.y] : Lmax;
    const int NB = (L + 31) / 32;
    // only the L real rows are kept (rows >= L are masked: reads clamp to row L-1), so that
    // L = 200, hd = 32 fits three workgroups per CU
    float *Vs = reinterpret_cast<float *>(smem);        // [L][VW]  (16-byte aligned rows)
    float *Ks = Vs + (size_t)Lmax * VW;                 // [L][KLD]
    unsigned int *padbits = reinterpret_cast<unsigned int *>(Ks + (size_t)Lmax * KLD); // [8]
    const int h = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, kk = lane >> 5;
    const int64_t base = off ? (int64_t)off[b] : (int64_t)b * Lmax;
    const int ld = 3 * d;
    const bool irn = (mask_mode == IRS_MASK_IRN);
    if (V4) {
        constexpr int C4 = HDP / 4;
        for (int idx = tid; idx < L * C4; idx += 256) {
            int j = idx / C4, c = (idx % C4) * 4;
            float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
            if (c < hd) {
                const float *row = qkv + (base + j) * ld + h * hd + c;
                kv = *reinterpret_cast<const float4 *>(row + d);
                vv = *reinterpret_cast<const float4 *>(row + 2 * d);
            }
            float *kd = Ks + (size_t)j * KLD + c;
            kd[0] = kv.x; kd[1] = kv.y; kd[2] = kv.z; kd[3] = kv.w;
            *reinterpret_cast<float4 *>(Vs + (size_t)j * VW + c) = vv;
        }
        if (VW > HDP)
            for (int idx = tid; idx < L * (VW - HDP); idx += 256) Vs[(size_t)(idx / (VW - HDP)) * VW + HDP + idx % (VW - HDP)] = 0.f;
    } else {
        for (int idx = tid; idx < L * VW; idx += 256) {
            int j = idx / VW, c = idx % VW;
            float kv = 0.f, vv = 0.f;
            if (c < hd) {
                const float *row = qkv + (base + j) * ld + h * hd + c;
                kv = row[d];
                vv = row[2 * d];
            }
            if (c < HDP) Ks[(size_t)j * KLD + c] = kv;
            Vs[idx] = vv;
        }
    }
    for (int kb = wave; kb < NB; kb += 4) { // masked-key bitmask of each key block
        int j = kb * 32 + lq;
        const int64_t jr = base + (j < L ? j : L - 1);
        bool masked = (j >= L) || (seq[tok_row ? (int64_t)tok_row[jr] : jr] == 0) || (irn && j == L - 1);
        unsigned long long bal = __ballot(masked);
        if (lane == 0) padbits[kb] = (unsigned int)bal;
    }
    __syncthreads();
    const float add_allowed = irn ? r_u[b] : 0.f;
    const float scale = 1.0f / sqrtf((float)hd);
    // the target (original position Lmax-1, never a pad when present) is the last row of the sequence
    const bool tgt_ok = irn && (seq[(int64_t)b * Lmax + Lmax - 1] != 0);

    for (int pass = 0; pass < 2; ++pass) {
        const int qb = pass == 0 ? wave : NB - 1 - wave;
        if (qb >= NB || (pass == 1 && qb < 4)) continue;
        const int qi = qb * 32 + lq; // this lane's query
        // Q fragments (B operand of S^T): Q[qi][kk*HH + s] * scale
        float qf[HH];
        {
            const float *qrow = qkv + (base + (qi < L ? qi : L - 1)) * ld + h * hd + kk * HH;
            if (V4) {
#pragma unroll
                for (int s4 = 0; s4 < HH / 4; ++s4) {
                    float4 t = (kk * HH + 4 * s4 < hd) ? *reinterpret_cast<const float4 *>(qrow + 4 * s4)
                                                       : make_float4(0.f, 0.f, 0.f, 0.f);
                    qf[4 * s4 + 0] = t.x * scale;
                    qf[4 * s4 + 1] = t.y * scale;
                    qf[4 * s4 + 2] = t.z * scale;
                    qf[4 * s4 + 3] = t.w * scale;
                }
            } else {
#pragma unroll
                for (int s2 = 0; s2 < HH; ++s2) qf[s2] = (kk * HH + s2 < hd) ? qrow[s2] * scale : 0.f;
            }
            if (qi >= L) {
#pragma unroll
                for (int s2 = 0; s2 < HH; ++s2) qf[s2] = 0.f;
            }
        }
        float m = -INFINITY, l = 0.f;
        f32x16 o[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[ct][r] = 0.f;
        if (tgt_ok) { // target column: s = q . K[L-1] + 1.0, p = 1
            float part = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < HH; ++s2) part = __fmaf_rn(qf[s2], Ks[(size_t)(L - 1) * KLD + kk * HH + s2], part);
            float st = lanes_sum<32>(part) + 1.0f;
            m = st;
            l = (kk == 0) ? 1.f : 0.f; // halves are summed at the end
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[ct][r] = Vs[(size_t)(L - 1) * VW + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk];
        }
        for (int kb = 0; kb <= qb; ++kb) {
            const unsigned int pm = padbits[kb];
            if (pm == 0xFFFFFFFFu) continue; // nothing visible in this key block (wave-uniform)
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
            const float *kp = Ks + (size_t)min(kb * 32 + lq, L - 1) * KLD + kk * HH;
#pragma unroll
            for (int s2 = 0; s2 < HH; ++s2) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[s2], qf[s2], sacc, 0, 0, 0);
            float mx = -INFINITY;
            if (pm == 0u && kb < qb) { // clean off-diagonal block: no masking
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    sacc[r] += add_allowed;
                    mx = fmaxf(mx, sacc[r]);
                }
            } else {
                const unsigned int pmk = pm >> (4 * kk);
                const int qlim = (kb < qb) ? 64 : lq - 4 * kk; // key index (within block, minus 4kk) must be <= qlim
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ki = (r & 3) + 8 * (r >> 2); // + 4kk = key index within the block
                    bool ok = !((pmk >> ki) & 1u) && (ki <= qlim);
                    float v = ok ? sacc[r] + add_allowed : -INFINITY;
                    sacc[r] = v;
                    mx = fmaxf(mx, v);
                }
            }
            mx = lanes_max<32>(mx);
            const float mn = fmaxf(m, mx);
            const bool dead = (mn == -INFINITY);
            const float alpha = dead ? 1.f : __expf(m - mn);
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float pv = dead ? 0.f : __expf(sacc[r] - mn);
                sacc[r] = pv;
                ps += pv;
            }
            l = l * alpha + ps;
            m = mn;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
                for (int r = 0; r < 16; ++r) o[ct][r] *= alpha;
                const float *vp = Vs + ct * 32 + lq; // A: V^T[c = lq][key]
                const int key0 = kb * 32 + 4 * kk;
#pragma unroll
                for (int t = 0; t < 16; ++t)
                    o[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[(size_t)min(key0 + (t & 3) + 8 * (t >> 2), L - 1) * VW], sacc[t],
                                                                 o[ct], 0, 0, 0);
            }
        }
        const float lt = lanes_sum<32>(l);
        const float inv = 1.0f / lt; // 0 (fully masked) -> inf, 0 * inf = NaN like torch
        if (qi < L) {
            float *orow = out + (base + qi) * d + h * hd;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = ct * 32 + 8 * g + 4 * kk;
                    if (V4) {
                        if (c < hd)
                            *reinterpret_cast<float4 *>(orow + c) = make_float4(o[ct][4 * g] * inv, o[ct][4 * g + 1] * inv,
                                                                                 o[ct][4 * g + 2] * inv, o[ct][4 * g + 3] * inv);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (c + e < hd) orow[c + e] = o[ct][4 * g + e] * inv;
                    }
                }
        }
    }
}

// ------------------------------------------------------------------ attention, head dim 32, 16-query blocks
// Same mathematics and masks as k_attn_mfma, restructured around v_mfma_f32_16x16x4_f32 for the throughput
// shapes: a causal sequence of L tokens has only ceil(L/32) 32-query blocks, too few and too unequal to keep
// four waves busy (L = 132: costs 1..5 key blocks -> the longest wave does 6 of 15), and half of every
// diagonal 32x32 block is masked.  With 16-query blocks x 16-key tiles the causal waste halves and a greedy
// longest-first assignment of the blocks to the waves balances to ~90 %.
//   S^T[key][q] = K . Q^T : lane (q = lane%16, gq = lane/16) holds Q[q][8gq .. 8gq+7] (B operand) and reads
//                 K[key][8gq .. 8gq+7] as two ds_read_b128 (A operand); C register r <-> key 4gq + r
//   O^T[c][q] += V^T . P^T: the S accumulator is the B operand (register j <-> key 4gq + j); the A operand
//                 V^T[c][4gq .. 4gq+3] is ONE ds_read_b128 from the transposed V image
// A query block keeps ALL its score tiles in registers (<= MAXT tiles of 4 registers): the softmax is two-pass
// (one max / shuffle pair per query block, no running rescale), the score tiles are independent MFMA chains and
// P.V runs on four accumulators -- the online form's per-tile shuffle + rescale chain was the critical path.
// LDS: K [L][32] with the 16-byte chunk index XOR (key&7)^((key>>3)&1), V^T [32][S] with S = 8 mod 16 floats --
// both conflict-free for the four 16-lane groups ds_read_b128 is serviced in.  Keys in [L, 16 ceil(L/16)) of
// the last tile carry p = 0; their K rows are zero-filled, their V^T columns are zero-filled up to S (a column
// index >= S aliases the next row's first keys: finite values times 0).
// Sum / maximum over the four lanes that share lane & 15 (l, l ^ 16, l ^ 32, l ^ 48), every lane getting the result:
// gfx950's v_permlane16_swap / v_permlane32_swap (vector ALU) instead of two ds_bpermute round trips through the LDS
// crossbar, each awaited at once.  With both operands the same register, the swap leaves {rows 0,0,2,2} and
// {rows 1,1,3,3} (16-lane rows), resp. {low half twice} and {high half twice}: their sum is x[l] + x[l ^ 16], resp.
// s[l] + s[l ^ 32], in every lane -- the same two additions as the shuffle form, so the bits are the same.
__device__ __forceinline__ float quad16_sum(float v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float quad16_max(float v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float s = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

__device__ __forceinline__ int attn16_vstride(int Lmax) {
    int S = (Lmax + 7) & ~7;
    if ((S & 15) != 8) S += 8;
    return S;
}

// FAST (packed sequences: the plan guarantees at most ONE masked token below the diagonal, index padq[b]): an
// off-diagonal tile is then unmasked unless it holds that token, so the per-pair mask words, the do/skip logic
// and the live-tile bits of the general form reduce to two comparisons of the pair index with the block index.
// The general form executed 5.3 scalar and 5.9 vector instructions and 1.2 branches per MFMA (PMC, round 2).
#ifdef ATTN_STAMP // (lab: tools/attn_lab.hip) per wave: start, loads issued, LDS filled, end; blocks done
__device__ unsigned long long g_attn_stamp[8 * 65536];
#define ATTN_T(i_) do { if (lane == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    g_attn_stamp[((size_t)((blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) & 65535) * 8 + (i_)] = t_; } } while (0)
#else
#define ATTN_T(i_)
#endif
// DMA (round 4; the production form): K and V of the (sequence, head) go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4), 8
// whole 128-byte key rows per instruction, straight from the row-major q | k | v rows the layer kernel writes -- no VGPR
// staging, no ds_write pass, no transposed copy.  The DMA writes LDS lane-linearly, so the two swizzles live on the SOURCE
// side (lane (row jl, chunk p) fetches the row's chunk p ^ sw): K keeps its conflict-free ds_read_b128 image (sw = (key & 7)
// ^ ((key >> 3) & 1)); V stays ROW-major [key][32] with the two 64-byte halves of a row exchanged on keys with bit 2 set
// (sw = 4 ((key >> 2) & 1)): the O^T += V^T P^T step reads its A operand V[4 gq + j][16 ct + lq] as four ds_read_b32 (two
// ds_read2_b32), and a 32-lane half (lane groups gq and gq + 1: keys 4 apart) then covers 32 distinct banks.  The register
// form's transposed V^T image cost ~1K cycles of 8-way conflicted ds_write_b32 per wave and fill, its staging loads'
// issue dominated a fill (39 % of a wave's life, profiles/r03/attn16_lab_stamps.txt), and PMC showed an LDS conflict
// ratio of 0.47.  Keys in [L, L16) re-read row L - 1 (finite values times p = 0).
// A score tile is consumed behind wave-uniform branches (pad in this tile? diagonal tile?).  The compiler pads the distance between a
// matrix instruction and the first vector instruction that reads its result only along the fall-through path; on a taken edge the
// reader may come too early -- the hardware does not interlock it, and whether it bites depends on how the wave was arbitrated
// (round 5: NaN rows in a lab form of the sequence-resident attention; the kernels below have the same shape and get the same
// explicit wait: 12 wait states behind the tile's last matrix instruction, then the tile counts as written here).
#define ATTN_MFMA_LANDED(a_) asm volatile("s_nop 7\n\ts_nop 3" : "+v"(a_));
#define ATTN_MFMA_LANDED2(a_, b_) asm volatile("s_nop 7\n\ts_nop 3" : "+v"(a_), "+v"(b_));
typedef __attribute__((address_space(3))) void attn_lds_void;
typedef const __attribute__((address_space(1))) void attn_glb_void;
// NW = waves per workgroup (4).  Measured and not kept (tools/attn_lab, profiles/r04/attn_lab_variants_r04.txt): NW = 8 -- the
// same LDS images serving twice the waves, two workgroups per CU, four waves per SIMD -- is 45 % SLOWER (the block phase is
// MFMA-issue bound at three waves per SIMD already: 94 % of the pipe's cycles while three waves are in their blocks); touching
// the K / V rows of the workgroup 256 or 768 dispatch slots ahead with dropped loads (an L2 prefetch) is 4 % slower.
// Also measured and not kept: ONE workgroup for the four heads of a short sequence (wave = head, the other three workgroups
// of the group returning at once): 227 vs 129 us on a batch of 40-token sequences -- a launch of short sequences is bound by
// the rate at which workgroups are DISPATCHED (~8 ns per workgroup chip-wide: 16384 workgroups = 129 us whatever they do), and
// workgroups that return at once are dispatched all the same.  What that regime needs is fewer dispatches: a persistent
// grid over a work list built by the plan kernel (profiles/r04/README.md).
template <int MAXT, bool FAST, bool DMA = false, int NW = 4>
__global__ void __launch_bounds__(64 * NW, NW == 8 ? 4 : 3) k_attn16(const float *__restrict__ qkv, const int64_t *__restrict__ seq,
                                                const float *__restrict__ r_u, float *__restrict__ out, int Lmax, int d,
                                                int mask_mode, const int32_t *__restrict__ off,
                                                const int32_t *__restrict__ cnt, const int32_t *__restrict__ padq,
                                                int out_frag, int H) {
    static_assert(NW == 4 || DMA, "the register-staged fill is written for four waves");
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    constexpr int HD = 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S = attn16_vstride(Lmax);
    const int Lcap16 = (Lmax + 15) & ~15;
    float *const Vt0 = reinterpret_cast<float *>(smem);             // [32][S]   (DMA: V [Lmax rounded to 16][32], halves swizzled)
    float *const Ks0 = Vt0 + (DMA ? Lcap16 * HD : 32 * S);          // [Lmax rounded to 16][32], swizzled chunks
    unsigned int *padbits = reinterpret_cast<unsigned int *>(Ks0 + (size_t)Lcap16 * HD); // [ceil(L/32)] (+ the work item at word 12)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 15, gq = lane >> 4;
    const int ld = 3 * d;
    const bool irn = (mask_mode == IRS_MASK_IRN);
    const int h = blockIdx.x, b = blockIdx.y;
    {
    const int L = cnt ? cnt[b] : Lmax;
    const int64_t base = off ? (int64_t)off[b] : (int64_t)b * Lmax;
    const int L16 = (L + 15) & ~15, NB16 = L16 >> 4;
    float *Vt = Vt0, *Ks = Ks0;
    // gridDim.z > 1 (few sequences, the latency path): the query blocks of a (sequence, head) are dealt one per wave
    // over 4 gridDim.z waves, largest first -- a wave's dependent chain is then one block instead of three, on
    // three times as many CUs.  Workgroup z owns blocks NB16-1-4z .. NB16-4-4z; each stages the whole K / V of the
    // (sequence, head) -- the loads of a fill are all in flight together, and the IRN target column needs key L-1.
    const int zsplit = blockIdx.z;
    if (gridDim.z > 1 && 4 * zsplit >= NB16) return;
    if (L <= 0) return; // (a plan never yields an empty sequence: the consumed row always counts)
    ATTN_T(0);
    if constexpr (DMA) {
        const int jl = lane >> 3, p = lane & 7;
        for (int i = wave; i < (L16 >> 3); i += NW) { // 8 key rows per instruction; the LDS destination is wave-uniform
            const int j = 8 * i + jl;
            const float *row = qkv + (base + (j < L ? j : L - 1)) * ld + h * HD;
            const int swk = (j & 7) ^ ((j >> 3) & 1), swv = ((j >> 2) & 1) << 2;
            __builtin_amdgcn_global_load_lds((attn_glb_void *)(row + d + 4 * (p ^ swk)), (attn_lds_void *)(Ks + i * 256), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((attn_glb_void *)(row + 2 * d + 4 * (p ^ swv)), (attn_lds_void *)(Vt + i * 256), 16, 0, 0);
        }
    }
    // K / V of this (sequence, head) -> LDS.  8 consecutive lanes take the 8 16-byte chunks of one key row, so a
    // load instruction covers 8 whole 128-byte K_h (V_h) slices -- in-kernel timing showed the ISSUE of these
    // loads, not their latency, dominating the fill when every lane touched a different row.  K's ds_write_b128
    // stays conflict-free; the transposed V^T ds_write_b32 are 8-way conflicted (a row stride S = 8 mod 16 puts
    // all 8 chunks of a key on one bank), ~1K cycles per wave once, cheaper than the slow loads were.  ALL global
    // loads are issued before the first LDS store (one memory round trip per workgroup).
    float4 kv[DMA ? 1 : MAXT / 2], vv[DMA ? 1 : MAXT / 2];
    if constexpr (!DMA) {
        const int jl = tid >> 3, c4 = tid & 7; // 8 lanes per key row: 128-byte contiguous K_h / V_h slices
#pragma unroll
        for (int it = 0; it < MAXT / 2; ++it) {
            const int j = jl + 32 * it;
            kv[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            vv[it] = kv[it];
            if (j < L) {
                const float *row = qkv + (base + j) * ld + h * HD + 4 * c4;
                kv[it] = *reinterpret_cast<const float4 *>(row + d);
                vv[it] = *reinterpret_cast<const float4 *>(row + 2 * d);
            }
        }
    }
    ATTN_T(1);
    // (the assignment below and the first Q request run while the K / V rows are in flight)
    // longest-first assignment of the 16-query blocks (block qb costs qb + 1 key tiles) to the 4 waves
    unsigned int mine = 0;
    if (gridDim.z > 1) {
        const int v = 4 * zsplit + wave;
        if (v < NB16) mine = 1u << (NB16 - 1 - v);
    } else if constexpr (NW == 4) { // the closed form of the greedy deal below (k_attn16h; tests/test_host_logic.py holds the identity)
        mine = __builtin_bitreverse32((0x01010101u << wave) | (0x01010101u << (7 - wave))) >> (32 - NB16);
    } else {
        int load[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) load[w] = 0;
        for (int qb = NB16 - 1; qb >= 0; --qb) {
            int w = 0;
#pragma unroll
            for (int v = 1; v < NW; ++v)
                if (load[v] < load[w]) w = v;
#pragma unroll
            for (int v = 0; v < NW; ++v) // (static indices: a run-time index would send the array to scratch)
                if (v == w) load[v] += qb + 1;
            if (w == wave) mine |= 1u << qb;
        }
    }
    // Q of a query block: lane (q, gq) holds Q[q][8gq .. 8gq+7]; the next block's rows are requested one block
    // ahead (the first one before the K / V fill) so their latency is never exposed
    auto load_q = [&](int qb, float4 &t0, float4 &t1) {
        const int qi = qb * 16 + lq;
        const float *qrow = qkv + (base + (qi < L ? qi : L - 1)) * ld + h * HD + 8 * gq;
        t0 = *reinterpret_cast<const float4 *>(qrow);
        t1 = *reinterpret_cast<const float4 *>(qrow + 4);
    };
    float4 qn0 = make_float4(0.f, 0.f, 0.f, 0.f), qn1 = qn0;
    int qb_next = mine ? 31 - __builtin_clz(mine) : -1;
    if (qb_next >= 0) load_q(qb_next, qn0, qn1);
    // masked-key bitmask of each 32-key block.  A packed sequence holds no pads except possibly its pos token
    // (index padq[b], recorded by the plan): no global loads on that path.
    const int pq = padq ? padq[b] : -1;
    for (int kb = wave; kb < (L + 31) / 32; kb += NW) {
        const int j = kb * 32 + (lane & 31);
        bool masked = (j >= L) || (irn && j == L - 1);
        if (padq) masked = masked || (j == pq);
        else masked = masked || (seq[base + (j < L ? j : L - 1)] == 0);
        const unsigned long long bal = __ballot(masked);
        if (lane == 0) padbits[kb] = (unsigned int)bal;
    }
    if constexpr (DMA) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's DMA pieces have landed (the barrier below publishes them)
    } else {
        const int jl = tid >> 3, c4 = tid & 7;
#pragma unroll
        for (int it = 0; it < MAXT / 2; ++it) {
            const int j = jl + 32 * it;
            if (j < L16) {
                *reinterpret_cast<float4 *>(Ks + j * HD + ((c4 ^ (j & 7) ^ ((j >> 3) & 1)) << 2)) = kv[it];
                if (j < S) {
                    Vt[(4 * c4 + 0) * S + j] = vv[it].x;
                    Vt[(4 * c4 + 1) * S + j] = vv[it].y;
                    Vt[(4 * c4 + 2) * S + j] = vv[it].z;
                    Vt[(4 * c4 + 3) * S + j] = vv[it].w;
                }
            }
        }
    }
    ATTN_T(2);
    __syncthreads();
    ATTN_T(3);
    // Scores are kept in the log2 domain (Q is pre-scaled by log2(e)/sqrt(hd), p = exp2(s - m)), and the IRN
    // mask's "+ r_u on every allowed key" is applied as "- r_u on the target column" instead (softmax is
    // shift-invariant over the unmasked keys): two VALU operations per score fewer.
    const float LOG2E = 1.4426950408889634f;
    const float tgt_add = irn ? (1.0f - r_u[b]) * LOG2E : 0.f;
    const float scale = LOG2E / sqrtf((float)HD);
    const bool tgt_ok = irn && (seq[(int64_t)b * Lmax + Lmax - 1] != 0);

    while (qb_next >= 0) {
        const int qb = qb_next;
        mine &= ~(1u << qb);
        const int qi = qb * 16 + lq; // this lane's query
        float qf[8];
        {
            const float4 t0 = qn0, t1 = qn1;
            qb_next = mine ? 31 - __builtin_clz(mine) : -1;
            if (qb_next >= 0) load_q(qb_next, qn0, qn1);
            const float sc = qi < L ? scale : 0.f;
            qf[0] = t0.x * sc, qf[1] = t0.y * sc, qf[2] = t0.z * sc, qf[3] = t0.w * sc;
            qf[4] = t1.x * sc, qf[5] = t1.y * sc, qf[6] = t1.z * sc, qf[7] = t1.w * sc;
        }
        // ---- pass 1: every visible score tile of this query block, masked, with the running maximum
        f32x4 sacc[MAXT];
        float mx = -INFINITY;
        auto score_tile = [&](int kt, f32x4 &sa) {
            const int key = kt * 16 + lq;
            const float *kr = Ks + key * HD;
            const int sw = (key & 7) ^ ((key >> 3) & 1);
            const float4 k0 = *reinterpret_cast<const float4 *>(kr + (((2 * gq) ^ sw) << 2));
            const float4 k1 = *reinterpret_cast<const float4 *>(kr + (((2 * gq + 1) ^ sw) << 2));
            sa = {0.f, 0.f, 0.f, 0.f};
            sa = __builtin_amdgcn_mfma_f32_16x16x4f32(k0.x, qf[0], sa, 0, 0, 0);
            sa = __builtin_amdgcn_mfma_f32_16x16x4f32(k0.y, qf[1], sa, 0, 0, 0);
            sa = __builtin_amdgcn_mfma_f32_16x16x4f32(k0.z, qf[2], sa, 0, 0, 0);
            sa = __builtin_amdgcn_mfma_f32_16x16x4f32(k0.w, qf[3], sa, 0, 0, 0);
            sa = __builtin_amdgcn_mfma_f32_16x16x4f32(k1.x, qf[4], sa, 0, 0, 0);
            sa = __builtin_amdgcn_mfma_f32_16x16x4f32(k1.y, qf[5], sa, 0, 0, 0);
            sa = __builtin_amdgcn_mfma_f32_16x16x4f32(k1.z, qf[6], sa, 0, 0, 0);
            sa = __builtin_amdgcn_mfma_f32_16x16x4f32(k1.w, qf[7], sa, 0, 0, 0);
        };
        auto mask_tile = [&](int kt, unsigned int pm, f32x4 &sa) {
            ATTN_MFMA_LANDED(sa)
            if (pm == 0u && kt < qb) { // clean off-diagonal tile: no masking
                mx = fmaxf(fmaxf(mx, sa[0]), fmaxf(sa[1], fmaxf(sa[2], sa[3])));
            } else {
                const unsigned int pmk = pm >> (4 * gq);
                const int qlim = (kt < qb) ? 64 : lq - 4 * gq; // key index within the tile, minus 4gq, must be <= qlim
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = !((pmk >> r) & 1u) && (r <= qlim);
                    const float v = ok ? sa[r] : -INFINITY;
                    sa[r] = v;
                    mx = fmaxf(mx, v);
                }
            }
        };
        unsigned int live = 0; // tiles with at least one unmasked key (wave-uniform)
        if (FAST) {
            const int pq_pair = pq >= 0 ? pq >> 5 : -1; // tile pair / lane group / register holding the one masked token
            auto pad_fix = [&](int kt, f32x4 &sa) {
                if ((pq >> 4) == kt) {
                    const bool mine = ((pq >> 2) & 3) == gq;
#pragma unroll
                    for (int r = 0; r < 4; ++r) sa[r] = (mine && (pq & 3) == r) ? -INFINITY : sa[r];
                }
            };
            const unsigned int pm_diag = (padbits[qb >> 1] >> (16 * (qb & 1))) & 0xFFFFu;
            // (pairs in groups of two under one guard: a block of few tiles does not walk every pair's own comparisons)
#pragma unroll
            for (int kg = 0; kg < MAXT / 4; ++kg) {
            if (4 * kg <= qb) {
#pragma unroll
            for (int kp = 2 * kg; kp < 2 * kg + 2; ++kp) {
                const int k0t = 2 * kp, k1t = 2 * kp + 1;
                if (k1t <= qb) { // two independent MFMA chains; tile k0t lies below the diagonal
                    score_tile(k0t, sacc[k0t]);
                    score_tile(k1t, sacc[k1t]);
                    ATTN_MFMA_LANDED2(sacc[k0t], sacc[k1t])
                    if (kp == pq_pair) {
                        pad_fix(k0t, sacc[k0t]);
                        if (k1t < qb) pad_fix(k1t, sacc[k1t]);
                    }
                    mx = fmaxf(fmaxf(mx, sacc[k0t][0]), fmaxf(sacc[k0t][1], fmaxf(sacc[k0t][2], sacc[k0t][3])));
                    if (k1t < qb) mx = fmaxf(fmaxf(mx, sacc[k1t][0]), fmaxf(sacc[k1t][1], fmaxf(sacc[k1t][2], sacc[k1t][3])));
                    else mask_tile(k1t, pm_diag, sacc[k1t]);
                } else if (k0t == qb) {
                    score_tile(k0t, sacc[k0t]);
                    mask_tile(k0t, pm_diag, sacc[k0t]);
                }
            }
            }
            }
            live = (2u << qb) - 1u;
        } else {
#pragma unroll
        for (int kp = 0; kp < MAXT / 2; ++kp) {
            const int k0t = 2 * kp, k1t = 2 * kp + 1;
            const unsigned int pw = (k0t <= qb) ? padbits[kp] : 0xFFFFFFFFu;
            const unsigned int pm0 = pw & 0xFFFFu, pm1 = pw >> 16;
            const bool do0 = pm0 != 0xFFFFu, do1 = (k1t <= qb) && pm1 != 0xFFFFu;
            if (do0 && do1) { // two independent MFMA chains
                score_tile(k0t, sacc[k0t]);
                score_tile(k1t, sacc[k1t]);
                mask_tile(k0t, pm0, sacc[k0t]);
                mask_tile(k1t, pm1, sacc[k1t]);
                live |= 3u << k0t;
            } else if (do0) {
                score_tile(k0t, sacc[k0t]);
                mask_tile(k0t, pm0, sacc[k0t]);
                live |= 1u << k0t;
            } else if (do1) {
                score_tile(k1t, sacc[k1t]);
                mask_tile(k1t, pm1, sacc[k1t]);
                live |= 2u << k0t;
            }
        }
        }
        // ---- the IRN target column (key L-1, +1.0, visible to every query): s = q . K[L-1] + 1.0
        float st = -INFINITY;
        if (tgt_ok) {
            const int jt = L - 1;
            const float *kr = Ks + jt * HD;
            const int sw = (jt & 7) ^ ((jt >> 3) & 1);
            const float4 k0 = *reinterpret_cast<const float4 *>(kr + (((2 * gq) ^ sw) << 2));
            const float4 k1 = *reinterpret_cast<const float4 *>(kr + (((2 * gq + 1) ^ sw) << 2));
            float part = qf[0] * k0.x;
            part = __fmaf_rn(qf[1], k0.y, part);
            part = __fmaf_rn(qf[2], k0.z, part);
            part = __fmaf_rn(qf[3], k0.w, part);
            part = __fmaf_rn(qf[4], k1.x, part);
            part = __fmaf_rn(qf[5], k1.y, part);
            part = __fmaf_rn(qf[6], k1.z, part);
            part = __fmaf_rn(qf[7], k1.w, part);
            st = quad16_sum(part) + tgt_add;
        }
        mx = quad16_max(mx);
        float m = fmaxf(mx, st);
        if (m == -INFINITY) m = 0.f; // nothing visible: every exp2(-inf - 0) is 0, l = 0 -> NaN row like torch
        // ---- pass 2: p = exp(s - m), O^T += V^T P^T on four accumulators (tile parity x column tile)
        float l = 0.f;
        f32x4 o[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[i][ct][r] = 0.f;
        if (tgt_ok) {
            const float pt = __builtin_amdgcn_exp2f(st - m);
            l = (gq == 0) ? pt : 0.f; // the four lanes of a query are summed at the end
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                if constexpr (DMA) {
                    const int jt = L - 1;
                    const float4 vt4 = *reinterpret_cast<const float4 *>(Vt + jt * HD + (((4 * ct + gq) ^ (((jt >> 2) & 1) << 2)) << 2));
                    o[0][ct][0] = pt * vt4.x, o[0][ct][1] = pt * vt4.y, o[0][ct][2] = pt * vt4.z, o[0][ct][3] = pt * vt4.w;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[0][ct][r] = pt * Vt[(16 * ct + 4 * gq + r) * S + L - 1];
                }
            }
        }
#pragma unroll
        for (int kg = 0; kg < MAXT / 4; ++kg) {
        if (!FAST || 4 * kg <= qb) { // FAST: tiles in groups of four under one guard
#pragma unroll
        for (int kt = 4 * kg; kt < 4 * kg + 4; ++kt) {
            if (FAST ? (kt <= qb) : ((live >> kt) & 1u) != 0u) { // wave-uniform; tiles beyond qb are never live
                f32x4 pa;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(sacc[kt][r] - m);
                    pa[r] = pv;
                    l += pv;
                }
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    float4 v4;
                    if constexpr (DMA) { // V[kt 16 + 4 gq + j][16 ct + lq], j = 0 .. 3 (rows 128 bytes apart)
                        const float *vb = Vt + (kt * 16 + 4 * gq) * HD + ((16 * ct + lq) ^ ((gq & 1) << 4));
                        v4 = make_float4(vb[0], vb[HD], vb[2 * HD], vb[3 * HD]);
                    } else
                        v4 = *reinterpret_cast<const float4 *>(Vt + (16 * ct + lq) * S + kt * 16 + 4 * gq);
                    f32x4 oo = o[kt & 1][ct];
                    oo = __builtin_amdgcn_mfma_f32_16x16x4f32(v4.x, pa[0], oo, 0, 0, 0);
                    oo = __builtin_amdgcn_mfma_f32_16x16x4f32(v4.y, pa[1], oo, 0, 0, 0);
                    oo = __builtin_amdgcn_mfma_f32_16x16x4f32(v4.z, pa[2], oo, 0, 0, 0);
                    oo = __builtin_amdgcn_mfma_f32_16x16x4f32(v4.w, pa[3], oo, 0, 0, 0);
                    o[kt & 1][ct] = oo;
                }
            }
        }
        }
        }
        const float lt = quad16_sum(l);
        const float inv = 1.0f / lt; // 0 (fully masked) -> inf, 0 * inf = NaN like torch
        if (qi < L) {
            if (out_frag) { // fragment-major image (d = 128: column block tn = head): the fused block kernel's B operand
                const int64_t tk = base + qi;
                float4 *of = reinterpret_cast<float4 *>(out) + ((size_t)(tk >> 5) * H + h) * 4 * 64 + (gq & 1) * 32 + (tk & 31);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    of[(2 * ct + (gq >> 1)) * 64] =
                        make_float4((o[0][ct][0] + o[1][ct][0]) * inv, (o[0][ct][1] + o[1][ct][1]) * inv,
                                    (o[0][ct][2] + o[1][ct][2]) * inv, (o[0][ct][3] + o[1][ct][3]) * inv);
            } else {
                float *orow = out + (base + qi) * d + h * HD + 4 * gq;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    *reinterpret_cast<float4 *>(orow + 16 * ct) =
                        make_float4((o[0][ct][0] + o[1][ct][0]) * inv, (o[0][ct][1] + o[1][ct][1]) * inv,
                                    (o[0][ct][2] + o[1][ct][2]) * inv, (o[0][ct][3] + o[1][ct][3]) * inv);
            }
        }
    }
    ATTN_T(4);
    } // work items
}

// ------------------------------------------------------------------ attention, head dim 32, split-float16 MFMAs (round 4)
// k_attn16 with O^T += V^T P^T on the 16-bit matrix pipe the way the layer kernel's IRS_GEMM_H3 mode multiplies: two FLOAT16
// planes per float32 operand (h = f16(x), l = f16(x - h): 22 of 24 significand bits), three plane products (h.l, l.h, h.h)
// on v_mfma_f32_16x16x32_f16 -- a pair of key tiles costs 6 matrix instructions of 16 cycles instead of 16 of 32.  The SCORES
// stay on the exact float32 chain (v_mfma_f32_16x16x4_f32 on a float32 K image, as in k_attn16): an error in a score is
// exponentiated, and a first form with K and q on float16 planes as well -- 26-37 % faster in the lab -- was 1e-3 off on rows
// whose logits are large (layer 0 of the synthetic models: |q||k| ~ 2000, so 2^-22 |q||k| is 5e-4 in the exponent); an error
// in p or V is not amplified.  (Round 5 measured the three-product float16 form of the scores here again, K planes written by the
// layer kernel's tail like V: the 6-layer decode of 4096 users 1.5 % faster only -- at three workgroups per CU this kernel is
// bound by vector-instruction issue, not by the matrix pipe -- with the rows' maximum distance to the float32-MFMA kernels 8.3e-5
// instead of 2.7e-5 and 2.7e-4 on a full decode's last-layer rows (test_throughput_shape_decode_matches_small_batches_and_oracle):
// not kept.  The sequence-resident kernel, whose attention phase WAS matrix-pipe heavy at two waves per SIMD, keeps it: -9 %.)
// What made the bf16 form of round 3 (k_attn16x) lose is gone:
//   * V arrives ALREADY SPLIT: the layer kernel's q | k | v tail (k_block_x6, kv_planes) writes each (token, head) V
//     row as [32 f16 h | 32 f16 l] -- the same 128 bytes as 32 floats -- so the fill is pure LDS-DMA (16 key rows x 64 B of
//     one plane per instruction, chunks swizzled on the source side) and costs no vector instruction;
//   * two planes are 256 B per key, the float32 images' footprint: three workgroups per CU as before;
//   * p = exp2(s - m) is split into two planes (3 vector instructions per value: v_cvt_pk_f16_f32, back, subtract, again).
// Everything else is k_attn16x's structure: [plane][key][64 B] images, V in front of K, V^T operands by the transposing
// ds_read_b64_tr_b16, the two score tiles of a pair side by side as the 32-key B operand, two-pass softmax, LPT blocks.
typedef __attribute__((ext_vector_type(4))) short attn_s16x4; // operand type of the transposing LDS read
__device__ __forceinline__ void attn_split8h(const float (&v)[8], x6_f16x8 (&P)[2]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const _Float16 h = (_Float16)v[j];
        P[0][j] = h;
        P[1][j] = (_Float16)(v[j] - (float)h);
    }
}
template <int MAXT, bool FAST, int NW>
__global__ void __launch_bounds__(64 * NW, 3) k_attn16h(const float *__restrict__ qkv, const int64_t *__restrict__ seq,
                                                       const float *__restrict__ r_u, float *__restrict__ out, int Lmax, int d,
                                                       int mask_mode, const int32_t *__restrict__ off,
                                                       const int32_t *__restrict__ cnt, const int32_t *__restrict__ padq,
                                                       int out_frag) {
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    constexpr int HD = 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // every per-sequence scalar of the kernel in ONE batch of loads (packed form: all the pointers exist).  The form that read
    // each where it was first used -- "cnt if present", "off if present", the pad index in front of the mask words, r_u and the
    // target id behind the fill barrier -- was five dependent scalar round trips per wave, two of them behind the barrier.
    const bool irn = (mask_mode == IRS_MASK_IRN);
    int L_, pq_ = -1;
    int64_t base_;
    float ru_ = 0.f;
    const int64_t last_id = seq[(int64_t)blockIdx.y * Lmax + Lmax - 1];
    if constexpr (FAST) {
        L_ = cnt[blockIdx.y], base_ = (int64_t)off[blockIdx.y], pq_ = padq[blockIdx.y];
        ru_ = (irn ? r_u : reinterpret_cast<const float *>(cnt))[blockIdx.y];
    } else {
        L_ = cnt ? cnt[blockIdx.y] : Lmax;
        base_ = off ? (int64_t)off[blockIdx.y] : (int64_t)blockIdx.y * Lmax;
        if (padq) pq_ = padq[blockIdx.y];
        if (irn) ru_ = r_u[blockIdx.y];
    }
    const int L = L_;
    const int L16max = (Lmax + 15) & ~15;
    const int PL = L16max * 64;                 // bytes of one plane image
    char *Vp = smem;                            // V: [2 planes][L16][64 B]
    float *Ks = reinterpret_cast<float *>(smem + 2 * PL); // K: float32 [L16][32], chunk-swizzled like k_attn16's (256 B per key in all)
    unsigned int *padbits = reinterpret_cast<unsigned int *>(smem + 4 * PL); // [ceil(L/32)]
    const int h = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 15, gq = lane >> 4;
    const int64_t base = base_;
    const int ld = 3 * d;
    const int L16 = (L + 15) & ~15, NB16 = L16 >> 4;
    if (L <= 0) return;
    // Fill by LDS-DMA.  K: float32 rows, 8 key rows x 128 B per instruction, chunk p of key j from source chunk p ^ (j & 7) ^
    // ((j >> 3) & 1) (k_attn16's conflict-free ds_read_b128 image).  V: the two float16 planes the layer kernel wrote, 16 key
    // rows x 64 B of ONE plane per instruction (4 lanes per row), chunks swizzled by ((key >> 2) & 1) << 1 for the transposing
    // reads.  Keys in [L, L16) re-read row L - 1 (finite values times p = 0); the pair partner of an odd last tile is not
    // read at all (its operand half is zero).
    {
        const char *qb8 = reinterpret_cast<const char *>(qkv);
        const int jl8 = lane >> 3, p8 = lane & 7;
        for (int i = wave; i < (L16 >> 3); i += NW) {
            const int j = 8 * i + jl8;
            const float *row = qkv + (base + (j < L ? j : L - 1)) * ld + h * HD;
            const int swk = (j & 7) ^ ((j >> 3) & 1);
            __builtin_amdgcn_global_load_lds((attn_glb_void *)(row + d + 4 * (p8 ^ swk)), (attn_lds_void *)(Ks + i * 256), 16, 0, 0);
        }
        const int jl = lane >> 2, c = lane & 3;
        for (int i = wave; i < NB16; i += NW) {
            const int j = 16 * i + jl;
            const char *row = qb8 + ((base + (j < L ? j : L - 1)) * ld) * 4 + h * 128;
            const int swv = ((j >> 2) & 1) << 1;
#pragma unroll
            for (int p = 0; p < 2; ++p)
                __builtin_amdgcn_global_load_lds((attn_glb_void *)(row + 2 * d * 4 + p * 64 + ((c ^ swv) << 4)),
                                                 (attn_lds_void *)(Vp + p * PL + i * 1024), 16, 0, 0);
        }
    }
    // longest-first assignment of the 16-query blocks (block qb costs qb + 1 key tiles) to the four waves.  With costs n, n - 1,
    // .. the greedy rule (next block to the least loaded wave, ties to the lowest wave) deals the blocks 0 1 2 3 3 2 1 0 0 1 ..
    // in descending order: the i-th largest block goes to wave i & 7 (< 4) or 7 - (i & 7) -- a constant bit pattern over i,
    // reversed into block order (the loop form was ~150 instructions per wave; tests/test_host_logic.py checks the identity).
    static_assert(NW == 4, "the closed form of the longest-first deal");
    unsigned int mine = __builtin_bitreverse32((0x01010101u << wave) | (0x01010101u << (7 - wave))) >> (32 - NB16);
    auto load_q = [&](int qb, float4 &t0, float4 &t1) {
        const int qi = qb * 16 + lq;
        const float *qrow = qkv + (base + (qi < L ? qi : L - 1)) * ld + h * HD + 8 * gq;
        t0 = *reinterpret_cast<const float4 *>(qrow);
        t1 = *reinterpret_cast<const float4 *>(qrow + 4);
    };
    float4 qn0 = make_float4(0.f, 0.f, 0.f, 0.f), qn1 = qn0;
    int qb_next = mine ? 31 - __builtin_clz(mine) : -1;
    if (qb_next >= 0) load_q(qb_next, qn0, qn1);
    const int pq = pq_;
    for (int kb = wave; kb < (L + 31) / 32; kb += NW) {
        const int j = kb * 32 + (lane & 31);
        bool masked = (j >= L) || (irn && j == L - 1);
        if (padq) masked = masked || (j == pq);
        else masked = masked || (seq[base + (j < L ? j : L - 1)] == 0);
        const unsigned long long bal = __ballot(masked);
        if (lane == 0) padbits[kb] = (unsigned int)bal;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's DMA pieces have landed (the barrier publishes them)
    __syncthreads();
    const float LOG2E = 1.4426950408889634f;
    const float tgt_add = irn ? (1.0f - ru_) * LOG2E : 0.f;
    const float scale = LOG2E / sqrtf((float)HD);
    const bool tgt_ok = irn && (last_id != 0);
    // this lane's addresses: K row read (key = 16 kt + lq, chunk gq), V transposed read (block row (lane & 15) >> 2, columns
    // 16 ct + 4 (lane & 3) ..)
    const int tq = (lane & 15) >> 2, tp = lane & 3;

    while (qb_next >= 0) {
        const int qb = qb_next;
        mine &= ~(1u << qb);
        const int qi = qb * 16 + lq;
        float qf[8];
        {
            const float4 t0 = qn0, t1 = qn1;
            qb_next = mine ? 31 - __builtin_clz(mine) : -1;
            if (qb_next >= 0) load_q(qb_next, qn0, qn1);
            const float sc = qi < L ? scale : 0.f;
            qf[0] = t0.x * sc, qf[1] = t0.y * sc, qf[2] = t0.z * sc, qf[3] = t0.w * sc;
            qf[4] = t1.x * sc, qf[5] = t1.y * sc, qf[6] = t1.z * sc, qf[7] = t1.w * sc;
        }
        f32x4 sacc[MAXT];
        float mx = -INFINITY;
        auto score_tile = [&](int kt, f32x4 &sa) __attribute__((always_inline)) {
            const int key = kt * 16 + lq;
            const float *kr = Ks + key * HD;
            const int sw = (key & 7) ^ ((key >> 3) & 1);
            const float4 k0 = *reinterpret_cast<const float4 *>(kr + (((2 * gq) ^ sw) << 2));
            const float4 k1 = *reinterpret_cast<const float4 *>(kr + (((2 * gq + 1) ^ sw) << 2));
            sa = {0.f, 0.f, 0.f, 0.f};
            sa = __builtin_amdgcn_mfma_f32_16x16x4f32(k0.x, qf[0], sa, 0, 0, 0);
            sa = __builtin_amdgcn_mfma_f32_16x16x4f32(k0.y, qf[1], sa, 0, 0, 0);
            sa = __builtin_amdgcn_mfma_f32_16x16x4f32(k0.z, qf[2], sa, 0, 0, 0);
            sa = __builtin_amdgcn_mfma_f32_16x16x4f32(k0.w, qf[3], sa, 0, 0, 0);
            sa = __builtin_amdgcn_mfma_f32_16x16x4f32(k1.x, qf[4], sa, 0, 0, 0);
            sa = __builtin_amdgcn_mfma_f32_16x16x4f32(k1.y, qf[5], sa, 0, 0, 0);
            sa = __builtin_amdgcn_mfma_f32_16x16x4f32(k1.z, qf[6], sa, 0, 0, 0);
            sa = __builtin_amdgcn_mfma_f32_16x16x4f32(k1.w, qf[7], sa, 0, 0, 0);
        };
        auto mask_tile = [&](int kt, unsigned int pm, f32x4 &sa) __attribute__((always_inline)) {
            ATTN_MFMA_LANDED(sa)
            if (pm == 0u && kt < qb) {
                mx = fmaxf(fmaxf(mx, sa[0]), fmaxf(sa[1], fmaxf(sa[2], sa[3])));
            } else {
                const unsigned int pmk = pm >> (4 * gq);
                const int qlim = (kt < qb) ? 64 : lq - 4 * gq;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = !((pmk >> r) & 1u) && (r <= qlim);
                    const float v = ok ? sa[r] : -INFINITY;
                    sa[r] = v;
                    mx = fmaxf(mx, v);
                }
            }
        };
        unsigned int live = 0;
        if (FAST) {
            const int pq_pair = pq >= 0 ? pq >> 5 : -1;
            auto pad_fix = [&](int kt, f32x4 &sa) __attribute__((always_inline)) {
                if ((pq >> 4) == kt) {
                    const bool mine_ = ((pq >> 2) & 3) == gq;
#pragma unroll
                    for (int r = 0; r < 4; ++r) sa[r] = (mine_ && (pq & 3) == r) ? -INFINITY : sa[r];
                }
            };
            const unsigned int pm_diag = (padbits[qb >> 1] >> (16 * (qb & 1))) & 0xFFFFu;
#pragma unroll
            for (int kg = 0; kg < MAXT / 4; ++kg) {
                if (4 * kg <= qb) {
#pragma unroll
                    for (int kp = 2 * kg; kp < 2 * kg + 2; ++kp) {
                        const int k0t = 2 * kp, k1t = 2 * kp + 1;
                        if (k1t <= qb) {
                            score_tile(k0t, sacc[k0t]);
                            score_tile(k1t, sacc[k1t]);
                            ATTN_MFMA_LANDED2(sacc[k0t], sacc[k1t])
                            if (kp == pq_pair) {
                                pad_fix(k0t, sacc[k0t]);
                                if (k1t < qb) pad_fix(k1t, sacc[k1t]);
                            }
                            mx = fmaxf(fmaxf(mx, sacc[k0t][0]), fmaxf(sacc[k0t][1], fmaxf(sacc[k0t][2], sacc[k0t][3])));
                            if (k1t < qb) mx = fmaxf(fmaxf(mx, sacc[k1t][0]), fmaxf(sacc[k1t][1], fmaxf(sacc[k1t][2], sacc[k1t][3])));
                            else mask_tile(k1t, pm_diag, sacc[k1t]);
                        } else if (k0t == qb) {
                            score_tile(k0t, sacc[k0t]);
                            mask_tile(k0t, pm_diag, sacc[k0t]);
                        }
                    }
                }
            }
            live = (2u << qb) - 1u;
        } else {
#pragma unroll
            for (int kp = 0; kp < MAXT / 2; ++kp) {
                const int k0t = 2 * kp, k1t = 2 * kp + 1;
                const unsigned int pw = (k0t <= qb) ? padbits[kp] : 0xFFFFFFFFu;
                const unsigned int pm0 = pw & 0xFFFFu, pm1 = pw >> 16;
                const bool do0 = pm0 != 0xFFFFu, do1 = (k1t <= qb) && pm1 != 0xFFFFu;
                if (do0) {
                    score_tile(k0t, sacc[k0t]);
                    mask_tile(k0t, pm0, sacc[k0t]);
                    live |= 1u << k0t;
                }
                if (do1) {
                    score_tile(k1t, sacc[k1t]);
                    mask_tile(k1t, pm1, sacc[k1t]);
                    live |= 2u << k0t;
                }
            }
        }
        // ---- the IRN target column (key L-1, +1.0, visible to every query): float32 arithmetic on the summed planes
        float st = -INFINITY;
        float vt[8]; // V[L-1][16 ct + 4 gq + r], ct = 0, 1
        if (tgt_ok) {
            const int jt = L - 1;
            const float *kr = Ks + jt * HD;
            const int sw = (jt & 7) ^ ((jt >> 3) & 1);
            const float4 k0 = *reinterpret_cast<const float4 *>(kr + (((2 * gq) ^ sw) << 2));
            const float4 k1 = *reinterpret_cast<const float4 *>(kr + (((2 * gq + 1) ^ sw) << 2));
            float part = qf[0] * k0.x;
            part = __fmaf_rn(qf[1], k0.y, part);
            part = __fmaf_rn(qf[2], k0.z, part);
            part = __fmaf_rn(qf[3], k0.w, part);
            part = __fmaf_rn(qf[4], k1.x, part);
            part = __fmaf_rn(qf[5], k1.y, part);
            part = __fmaf_rn(qf[6], k1.z, part);
            part = __fmaf_rn(qf[7], k1.w, part);
            st = quad16_sum(part) + tgt_add;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) { // columns 16 ct + 4 gq .. + 3: chunk 2 ct + (gq >> 1), half gq & 1
                typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
                const char *vr = Vp + jt * 64 + ((((2 * ct + (gq >> 1)) ^ (((jt >> 2) & 1) << 1))) << 4) + 8 * (gq & 1);
                const f16x4 b0 = *reinterpret_cast<const f16x4 *>(vr), b1 = *reinterpret_cast<const f16x4 *>(vr + PL);
#pragma unroll
                for (int r = 0; r < 4; ++r) vt[4 * ct + r] = (float)b0[r] + (float)b1[r];
            }
        }
        mx = quad16_max(mx);
        float m = fmaxf(mx, st);
        if (m == -INFINITY) m = 0.f;
        // ---- pass 2: p = exp2(s - m); O^T += V^T P^T, one 32-key MFMA step per pair of tiles
        float l = 0.f;
        f32x4 o[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[i][ct][r] = 0.f;
        float ot[8]; // the target column's contribution (float32, added at the end)
#pragma unroll
        for (int e = 0; e < 8; ++e) ot[e] = 0.f;
        if (tgt_ok) {
            const float pt = __builtin_amdgcn_exp2f(st - m);
            l = (gq == 0) ? pt : 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) ot[e] = pt * vt[e];
        }
#pragma unroll
        for (int kp = 0; kp < MAXT / 2; ++kp) {
            const int k0t = 2 * kp, k1t = 2 * kp + 1;
            const bool on0 = FAST ? (k0t <= qb) : (((live >> k0t) & 1u) != 0u);
            const bool on1 = FAST ? (k1t <= qb) : (((live >> k1t) & 1u) != 0u);
            if (on0 || on1) { // wave-uniform
                float pa[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pa[r] = on0 ? __builtin_amdgcn_exp2f(sacc[k0t][r] - m) : 0.f;
                    pa[4 + r] = on1 ? __builtin_amdgcn_exp2f(sacc[k1t][r] - m) : 0.f;
                    l += pa[r] + pa[4 + r];
                }
                x6_f16x8 P[2];
                attn_split8h(pa, P);
                const int r0 = k0t * 16 + 4 * gq + tq, r1 = r0 + 16;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const int o0 = r0 * 64 + ((((2 * ct + (tp >> 1)) ^ (((r0 >> 2) & 1) << 1))) << 4) + 8 * (tp & 1);
                    const int o1 = r1 * 64 + ((((2 * ct + (tp >> 1)) ^ (((r1 >> 2) & 1) << 1))) << 4) + 8 * (tp & 1);
                    x6_f16x8 V[2];
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const attn_s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) attn_s16x4 *)(Vp + p * PL + o0));
                        attn_s16x4 a1 = {0, 0, 0, 0}; // (wave-uniform: a pair whose second tile does not exist reads nothing past the image)
                        if (on1) a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) attn_s16x4 *)(Vp + p * PL + o1));
                        typedef __attribute__((ext_vector_type(8))) short s16x8;
                        const s16x8 both = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                        V[p] = __builtin_bit_cast(x6_f16x8, both);
                    }
                    f32x4 oo = o[kp & 1][ct];
                    oo = __builtin_amdgcn_mfma_f32_16x16x32_f16(V[0], P[1], oo, 0, 0, 0);
                    oo = __builtin_amdgcn_mfma_f32_16x16x32_f16(V[1], P[0], oo, 0, 0, 0);
                    oo = __builtin_amdgcn_mfma_f32_16x16x32_f16(V[0], P[0], oo, 0, 0, 0);
                    o[kp & 1][ct] = oo;
                }
            }
        }
        const float lt = quad16_sum(l);
        const float inv = 1.0f / lt;
        if (qi < L) {
            if (out_frag) {
                const int64_t tk = base + qi;
                float4 *of = reinterpret_cast<float4 *>(out) + ((size_t)(tk >> 5) * gridDim.x + h) * 4 * 64 + (gq & 1) * 32 + (tk & 31);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    of[(2 * ct + (gq >> 1)) * 64] =
                        make_float4(((o[0][ct][0] + o[1][ct][0]) + ot[4 * ct + 0]) * inv, ((o[0][ct][1] + o[1][ct][1]) + ot[4 * ct + 1]) * inv,
                                    ((o[0][ct][2] + o[1][ct][2]) + ot[4 * ct + 2]) * inv, ((o[0][ct][3] + o[1][ct][3]) + ot[4 * ct + 3]) * inv);
            } else {
                float *orow = out + (base + qi) * d + h * HD + 4 * gq;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    *reinterpret_cast<float4 *>(orow + 16 * ct) =
                        make_float4(((o[0][ct][0] + o[1][ct][0]) + ot[4 * ct + 0]) * inv, ((o[0][ct][1] + o[1][ct][1]) + ot[4 * ct + 1]) * inv,
                                    ((o[0][ct][2] + o[1][ct][2]) + ot[4 * ct + 2]) * inv, ((o[0][ct][3] + o[1][ct][3]) + ot[4 * ct + 3]) * inv);
            }
        }
    }
}

// ------------------------------------------------------------------ one query block of the sequence-resident kernel (round 5)
// k_attn16h's block body (FAST form: a packed sequence, at most one pad below the diagonal; two-pass softmax with every score
// tile of the block in registers, O^T += V^T P^T on float16 plane pairs through the transposing LDS read) with the SCORES too on
// exact float16 plane products (three v_mfma_f32_16x16x32_f16 per tile instead of eight 16x16x4_f32), on images the layer
// kernel's own waves wrote: Ks = [key][2 planes x 32 float16] chunk-swizzled, Vp = [2 planes][PL bytes] of [key][64 B], both
// based at the sequence's first row.  The query rows come from the wave's q scratch ([16 rows][36] float32), the normalised
// output goes to the fragment-major scratch tile (`of` = the tile's float4 base + 16 blk: lane (lq, gq) writes token column lq);
// query rows beyond the sequence store zeros (the layer body multiplies whole tiles: they must stay finite).
__device__ __forceinline__ void seq_attn_block(const float *Ks, const char *Vp, int PL, int L, int qb, bool irn, float tgt_add, bool tgt_ok,
                                               int pq, const float *qscr, float4 *of, bool store) {
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    constexpr int HD = 32, MAXT = 16;
    int lane = threadIdx.x & 63;
    // (laundered: the lane-derived masks, offsets and predicates below are the same for every head and layer; hoisted out of the
    //  caller's loops they cost ~50 registers held -- spilled -- for the whole kernel instead of ~30 instructions per call)
    asm volatile("" : "+v"(lane));
    const int lq = lane & 15, gq = lane >> 4;
    const int tq = (lane & 15) >> 2, tp = lane & 3;
    float4 *dst = of + (gq & 1) * 32 + lq;
    if (16 * qb >= L) { // the whole block lies beyond the sequence
        if (store) {
            dst[(gq >> 1) * 64] = make_float4(0.f, 0.f, 0.f, 0.f);
            dst[(2 + (gq >> 1)) * 64] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    const float LOG2E = 1.4426950408889634f;
    const float scale = LOG2E / sqrtf((float)HD);
    const int qi = qb * 16 + lq;
    float qf[8];
    {
        const float4 t0 = *reinterpret_cast<const float4 *>(qscr + lq * 36 + 8 * gq), t1 = *reinterpret_cast<const float4 *>(qscr + lq * 36 + 8 * gq + 4);
        const float sc = qi < L ? scale * SEQ_KQ_SCALE : 0.f;
        qf[0] = t0.x * sc, qf[1] = t0.y * sc, qf[2] = t0.z * sc, qf[3] = t0.w * sc;
        qf[4] = t1.x * sc, qf[5] = t1.y * sc, qf[6] = t1.z * sc, qf[7] = t1.w * sc;
    }
    // The score tiles are consumed behind wave-uniform BRANCHES (does this pair hold the pad?  is the tile the diagonal one?).  The
    // compiler pads the distance between a matrix instruction and the first vector instruction that reads its result with s_nop --
    // along the fall-through path; on the TAKEN edge of such a branch a lab form of this function (two passes over the tiles with
    // the scores recomputed: runtime loops instead of 16 unrolled positions) was left with 2 instructions behind a 16-cycle MFMA.
    // The hardware does not interlock that: the reader got the register's old contents, the row maximum came out too small, exp2
    // overflowed -- NaN rows in a third of the three-block sequences (profiles/r05/README.md).  The statement below reads and
    // "writes" the tiles behind 12 wait states of its own: every path to a consumer is long enough whatever the block layout.
    // (That two-pass form, for the record: 64 registers fewer, 15 % less code, bit-identical rows; 5 % FASTER on 16-token sequences,
    //  1.5 - 2.5 % SLOWER on the bench's windows and on 208-token sequences -- a block's time is its vector + matrix instruction
    //  count at 4 cycles of its SIMD each, two waves sharing the SIMD, and recomputing adds three matrix instructions per tile.)
#define SEQ_MFMA_LANDED(a_, b_) asm volatile("s_nop 7\n\ts_nop 3" : "+v"(a_), "+v"(b_));
    f32x4 sacc[MAXT];
    float mx = -INFINITY;
    // scores on exact float16 plane products (round 5): K arrives as two planes (the front's epilogue splits each key row
    // once), q is split here once per block; Kl.qh + Kh.ql + Kh.qh on v_mfma_f32_16x16x32_f16 (the whole head dim in one
    // instruction: 3 x 16 cycles per tile instead of 8 x 32 on the float32 pipe).  Lane (lq, gq): key 16 kt + lq, channels 8 gq ..
    x6_f16x8 Q[2];
    attn_split8h(qf, Q);
    const char *Kc = reinterpret_cast<const char *>(Ks);
    auto score_tile = [&](int kt, f32x4 &sa) __attribute__((always_inline)) {
        const int key = kt * 16 + lq;
        const char *kr = Kc + key * 128;
        const int sw = (key >> 1) & 7;
        const x6_f16x8 kh = *reinterpret_cast<const x6_f16x8 *>(kr + ((gq ^ sw) << 4));
        const x6_f16x8 kl = *reinterpret_cast<const x6_f16x8 *>(kr + (((4 + gq) ^ sw) << 4));
        sa = {0.f, 0.f, 0.f, 0.f};
        sa = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl, Q[0], sa, 0, 0, 0);
        sa = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, Q[1], sa, 0, 0, 0);
        sa = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, Q[0], sa, 0, 0, 0);
        if (SEQ_KQ_SCALE != 1.0f) sa *= 1.0f / (SEQ_KQ_SCALE * SEQ_KQ_SCALE);
    };
    // masked keys of the diagonal tile: beyond the sequence, the IRN target column (added separately), the one pad
    unsigned int pm_diag = 0;
    {
        const int k0_ = 16 * qb;
        if (L - k0_ < 16) pm_diag = (0xFFFFu << (L - k0_)) & 0xFFFFu;
        if (irn && L - 1 >= k0_ && L - 1 < k0_ + 16) pm_diag |= 1u << (L - 1 - k0_);
        if (pq >= k0_ && pq < k0_ + 16) pm_diag |= 1u << (pq - k0_);
    }
    auto mask_diag = [&](int kt, f32x4 &sa) __attribute__((always_inline)) {
        const unsigned int pmk = pm_diag >> (4 * gq);
        const int qlim = lq - 4 * gq;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = !((pmk >> r) & 1u) && (r <= qlim);
            const float v = ok ? sa[r] : -INFINITY;
            sa[r] = v;
            mx = fmaxf(mx, v);
        }
    };
    const int pq_pair = pq >= 0 ? pq >> 5 : -1;
    auto pad_fix = [&](int kt, f32x4 &sa) __attribute__((always_inline)) {
        if ((pq >> 4) == kt) {
            const bool mine_ = ((pq >> 2) & 3) == gq;
#pragma unroll
            for (int r = 0; r < 4; ++r) sa[r] = (mine_ && (pq & 3) == r) ? -INFINITY : sa[r];
        }
    };
#pragma unroll
    for (int kg = 0; kg < MAXT / 4; ++kg) {
        if (4 * kg <= qb) {
#pragma unroll
            for (int kp = 2 * kg; kp < 2 * kg + 2; ++kp) {
                const int k0t = 2 * kp, k1t = 2 * kp + 1;
                if (k1t <= qb) {
                    score_tile(k0t, sacc[k0t]);
                    score_tile(k1t, sacc[k1t]);
                    SEQ_MFMA_LANDED(sacc[k0t], sacc[k1t])
                    if (kp == pq_pair) {
                        pad_fix(k0t, sacc[k0t]);
                        if (k1t < qb) pad_fix(k1t, sacc[k1t]);
                    }
                    mx = fmaxf(fmaxf(mx, sacc[k0t][0]), fmaxf(sacc[k0t][1], fmaxf(sacc[k0t][2], sacc[k0t][3])));
                    if (k1t < qb) mx = fmaxf(fmaxf(mx, sacc[k1t][0]), fmaxf(sacc[k1t][1], fmaxf(sacc[k1t][2], sacc[k1t][3])));
                    else mask_diag(k1t, sacc[k1t]);
                } else if (k0t == qb) {
                    score_tile(k0t, sacc[k0t]);
                    SEQ_MFMA_LANDED(sacc[k0t], sacc[k0t])
                    mask_diag(k0t, sacc[k0t]);
                }
            }
        }
    }
    // ---- the IRN target column (key L - 1, +1.0 where every other visible key carries +r_u, visible to every query)
    float st = -INFINITY;
    float vt[8];
    if (tgt_ok) {
        const int jt = L - 1;
        const char *kr = Kc + jt * 128;
        const int sw = (jt >> 1) & 7;
        const x6_f16x8 th = *reinterpret_cast<const x6_f16x8 *>(kr + ((gq ^ sw) << 4));
        const x6_f16x8 tl = *reinterpret_cast<const x6_f16x8 *>(kr + (((4 + gq) ^ sw) << 4));
        const float4 k0 = make_float4((float)th[0] + (float)tl[0], (float)th[1] + (float)tl[1], (float)th[2] + (float)tl[2], (float)th[3] + (float)tl[3]);
        const float4 k1 = make_float4((float)th[4] + (float)tl[4], (float)th[5] + (float)tl[5], (float)th[6] + (float)tl[6], (float)th[7] + (float)tl[7]);
        float part = qf[0] * k0.x;
        part = __fmaf_rn(qf[1], k0.y, part);
        part = __fmaf_rn(qf[2], k0.z, part);
        part = __fmaf_rn(qf[3], k0.w, part);
        part = __fmaf_rn(qf[4], k1.x, part);
        part = __fmaf_rn(qf[5], k1.y, part);
        part = __fmaf_rn(qf[6], k1.z, part);
        part = __fmaf_rn(qf[7], k1.w, part);
        st = quad16_sum(part) * (1.0f / (SEQ_KQ_SCALE * SEQ_KQ_SCALE)) + tgt_add;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
            const char *vr = Vp + jt * 64 + ((((2 * ct + (gq >> 1)) ^ (((jt >> 2) & 1) << 1))) << 4) + 8 * (gq & 1);
            const f16x4 b0 = *reinterpret_cast<const f16x4 *>(vr), b1 = *reinterpret_cast<const f16x4 *>(vr + PL);
#pragma unroll
            for (int r = 0; r < 4; ++r) vt[4 * ct + r] = (float)b0[r] + (float)b1[r];
        }
    }
    mx = quad16_max(mx);
    float m = fmaxf(mx, st);
    if (m == -INFINITY) m = 0.f;
    float l = 0.f;
    f32x4 o[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[i][ct][r] = 0.f;
    float ot[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) ot[e] = 0.f;
    if (tgt_ok) {
        const float pt = __builtin_amdgcn_exp2f(st - m);
        l = (gq == 0) ? pt : 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) ot[e] = pt * vt[e];
    }
#pragma unroll
    for (int kp = 0; kp < MAXT / 2; ++kp) {
        const int k0t = 2 * kp, k1t = 2 * kp + 1;
        const bool on0 = k0t <= qb, on1 = k1t <= qb;
        if (on0 || on1) { // wave-uniform
            float pa[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pa[r] = on0 ? __builtin_amdgcn_exp2f(sacc[k0t][r] - m) : 0.f;
                pa[4 + r] = on1 ? __builtin_amdgcn_exp2f(sacc[k1t][r] - m) : 0.f;
                l += pa[r] + pa[4 + r];
            }
            x6_f16x8 P[2];
            attn_split8h(pa, P);
            const int r0 = k0t * 16 + 4 * gq + tq, r1 = r0 + 16;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int o0 = r0 * 64 + ((((2 * ct + (tp >> 1)) ^ (((r0 >> 2) & 1) << 1))) << 4) + 8 * (tp & 1);
                const int o1 = r1 * 64 + ((((2 * ct + (tp >> 1)) ^ (((r1 >> 2) & 1) << 1))) << 4) + 8 * (tp & 1);
                x6_f16x8 V[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const attn_s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) attn_s16x4 *)(Vp + p * PL + o0));
                    attn_s16x4 a1 = {0, 0, 0, 0};
                    if (on1) a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) attn_s16x4 *)(Vp + p * PL + o1));
                    typedef __attribute__((ext_vector_type(8))) short s16x8;
                    const s16x8 both = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                    V[p] = __builtin_bit_cast(x6_f16x8, both);
                }
                f32x4 oo = o[kp & 1][ct];
                oo = __builtin_amdgcn_mfma_f32_16x16x32_f16(V[0], P[1], oo, 0, 0, 0);
                oo = __builtin_amdgcn_mfma_f32_16x16x32_f16(V[1], P[0], oo, 0, 0, 0);
                oo = __builtin_amdgcn_mfma_f32_16x16x32_f16(V[0], P[0], oo, 0, 0, 0);
                o[kp & 1][ct] = oo;
            }
        }
    }
    const float lt = quad16_sum(l);
    const float inv = qi < L ? 1.0f / lt : 0.f;
    if (store) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            float4 v = make_float4(((o[0][ct][0] + o[1][ct][0]) + ot[4 * ct + 0]) * inv, ((o[0][ct][1] + o[1][ct][1]) + ot[4 * ct + 1]) * inv,
                                   ((o[0][ct][2] + o[1][ct][2]) + ot[4 * ct + 2]) * inv, ((o[0][ct][3] + o[1][ct][3]) + ot[4 * ct + 3]) * inv);
            if (qi >= L) v = make_float4(0.f, 0.f, 0.f, 0.f);
            dst[(2 * ct + (gq >> 1)) * 64] = v;
        }
    }
#undef SEQ_MFMA_LANDED
}

// ------------------------------------------------------------------ single-query attention (last layer, rows-only decode)
// Only row pos[b] of the last layer is consumed by the scoring step (the reference computes all L
// rows and uses output[index][history_end_pos], influentialRS.py:374,421).  One wave per (sequence,
// head): lanes over keys for the scores, lanes over head columns for P.V.  Same mask semantics as
// k_attn_mfma.  out_rows[b][h*hd + c].
__global__ void __launch_bounds__(64) k_attn_row(const float *__restrict__ qkv, const int64_t *__restrict__ seq,
                                                 const float *__restrict__ r_u, const int32_t *__restrict__ pos,
                                                 float *__restrict__ out_rows, int Lmax, int d, int hd, int mask_mode,
                                                 const int32_t *__restrict__ off, const int32_t *__restrict__ cnt,
                                                 const int32_t *__restrict__ tok_row, const int32_t *__restrict__ qrow,
                                                 const float *__restrict__ qrows) {
    __shared__ float p_s[256];
    __shared__ float q_s[64];
    const int h = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const int L = cnt ? cnt[b] : Lmax;
    const int64_t base = off ? (int64_t)off[b] : (int64_t)b * Lmax;
    const int ld = 3 * d;
    int i = qrow ? qrow[b] - (int)base : pos[b]; // query row within the sequence
    if (i < 0) i = 0;
    if (i >= L) i = L - 1;
    const bool has_tgt = (seq[(int64_t)b * Lmax + Lmax - 1] != 0);
    const bool irn = (mask_mode == IRS_MASK_IRN);
    const float add_allowed = irn ? r_u[b] : 0.f;
    const float scale = 1.0f / sqrtf((float)hd);
    // qrows: the queries of the consumed rows computed separately (the producer wrote k | v only)
    if (lane < hd) q_s[lane] = (qrows ? qrows[(int64_t)b * d + h * hd + lane] : qkv[(base + i) * ld + h * hd + lane]) * scale;
    __syncthreads();
    float sc[4];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int j = lane + 64 * t;
        float s = -INFINITY;
        if (j < L) {
            const bool is_tgt = irn && has_tgt && (j == L - 1);
            const int64_t jr = base + j;
            const bool ok = (seq[tok_row ? (int64_t)tok_row[jr] : jr] != 0) && (is_tgt || j <= i);
            if (ok) {
                const float *kr = qkv + (base + j) * ld + d + h * hd;
                float acc = 0.f;
                if ((hd & 3) == 0 && (d & 3) == 0 && ((((uintptr_t)qkv) & 15) == 0)) {
                    for (int c = 0; c < hd; c += 4) {
                        const float4 k4 = *reinterpret_cast<const float4 *>(kr + c);
                        acc = __fmaf_rn(q_s[c], k4.x, acc);
                        acc = __fmaf_rn(q_s[c + 1], k4.y, acc);
                        acc = __fmaf_rn(q_s[c + 2], k4.z, acc);
                        acc = __fmaf_rn(q_s[c + 3], k4.w, acc);
                    }
                } else {
                    for (int c = 0; c < hd; ++c) acc = __fmaf_rn(q_s[c], kr[c], acc);
                }
                s = acc + (is_tgt ? 1.0f : add_allowed);
            }
        }
        sc[t] = s;
        mx = fmaxf(mx, s);
    }
    mx = lanes_max<63>(mx);
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int j = lane + 64 * t;
        float pv = (sc[t] == -INFINITY) ? 0.f : __expf(sc[t] - mx);
        p_s[j] = pv;
        sum += pv;
    }
    sum = wave_sum(sum);
    __syncthreads();
    // P.V: lane = (key parity group, column); independent coalesced loads, two partial sums combined at the end
    {
        const int c = lane & 31, half = lane >> 5;
        const int nc = hd > 32 ? 2 : 1; // hd <= 64: up to two column passes
        for (int cp = 0; cp < nc; ++cp) {
            const int col = cp * 32 + c;
            float o = 0.f;
            if (col < hd) {
                const float *vcol = qkv + base * ld + 2 * d + h * hd + col;
                for (int j = half; j < L; j += 2) o = __fmaf_rn(p_s[j], vcol[(int64_t)j * ld], o);
            }
            o = lanes_sum<32>(o);
            if (half == 0 && col < hd) out_rows[(int64_t)b * d + h * hd + col] = o / sum; // sum == 0 -> NaN like torch
        }
    }
}

// The same row for head dim 32 and L <= 256 on four waves: 8 lanes per key (one 16-byte chunk of K_h and of V_h
// each), every load of the workgroup requested before the first use -- one memory round trip instead of the ~10
// dependent ones of the one-wave loop above (14 -> 5 us on the single-user path, where every launch starts cold).
// Packed sequences only: padq is the plan's record of the one pad a packed sequence may hold (no seq lookups).
__global__ void __launch_bounds__(256) k_attn_row32(const float *__restrict__ qkv, const int64_t *__restrict__ seq,
                                                    const float *__restrict__ r_u, float *__restrict__ out_rows, int Lmax, int d, int mask_mode,
                                                    const int32_t *__restrict__ off, const int32_t *__restrict__ cnt,
                                                    const int32_t *__restrict__ qrow, const float *__restrict__ qrows,
                                                    const int32_t *__restrict__ padq) {
    constexpr int HD = 32, NIT = 8;
    __shared__ float red[4][36]; // per wave: 32 output columns, [32] max, [33] sum
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = cnt[b];
    const int64_t base = off[b];
    const int ld = 3 * d;
    const int jl = tid >> 3, c4 = tid & 7;
    float4 kv[NIT], vv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int j = jl + 32 * it;
        kv[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        vv[it] = kv[it];
        if (j < L) {
            const float *row = qkv + (base + j) * ld + h * HD + 4 * c4;
            kv[it] = *reinterpret_cast<const float4 *>(row + d);
            vv[it] = *reinterpret_cast<const float4 *>(row + 2 * d);
        }
    }
    int i = qrow[b] - (int)base; // query row within the packed sequence
    if (i < 0) i = 0;
    if (i >= L) i = L - 1;
    const bool has_tgt = (seq[(int64_t)b * Lmax + Lmax - 1] != 0);
    const bool irn = (mask_mode == IRS_MASK_IRN);
    const float add_allowed = irn ? r_u[b] : 0.f;
    const float scale = 1.0f / sqrtf((float)HD);
    const int pq = padq[b];
    float4 q4 = *reinterpret_cast<const float4 *>(qrows ? qrows + (int64_t)b * d + h * HD + 4 * c4
                                                         : qkv + (base + i) * ld + h * HD + 4 * c4);
    q4.x *= scale, q4.y *= scale, q4.z *= scale, q4.w *= scale;
    float sc[NIT];
    float mx = -INFINITY;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int j = jl + 32 * it;
        float part = q4.x * kv[it].x;
        part = __fmaf_rn(q4.y, kv[it].y, part);
        part = __fmaf_rn(q4.z, kv[it].z, part);
        part = __fmaf_rn(q4.w, kv[it].w, part);
        part = lanes_sum<7>(part);
        const bool is_tgt = irn && has_tgt && (j == L - 1);
        const bool ok = (j < L) && (j != pq) && (is_tgt || j <= i);
        sc[it] = ok ? part + (is_tgt ? 1.0f : add_allowed) : -INFINITY;
        mx = fmaxf(mx, sc[it]);
    }
    mx = lanes_max<56>(mx); // the 8 lanes of a key already agree
    if (lane == 0) red[wave][32] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0][32], red[1][32]), fmaxf(red[2][32], red[3][32]));
    float sum = 0.f;
    float4 o4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const float pv = (sc[it] == -INFINITY) ? 0.f : __expf(sc[it] - mx);
        sum += pv;
        o4.x = __fmaf_rn(pv, vv[it].x, o4.x);
        o4.y = __fmaf_rn(pv, vv[it].y, o4.y);
        o4.z = __fmaf_rn(pv, vv[it].z, o4.z);
        o4.w = __fmaf_rn(pv, vv[it].w, o4.w);
    }
    // over the 8 key rows of the wave (lanes with the same chunk): butterfly levels 8, 16, 32
    sum = lanes_sum<56>(sum);
    o4.x = lanes_sum<56>(o4.x);
    o4.y = lanes_sum<56>(o4.y);
    o4.z = lanes_sum<56>(o4.z);
    o4.w = lanes_sum<56>(o4.w);
    if (lane < 8) {
        red[wave][4 * c4 + 0] = o4.x, red[wave][4 * c4 + 1] = o4.y, red[wave][4 * c4 + 2] = o4.z, red[wave][4 * c4 + 3] = o4.w;
        if (lane == 0) red[wave][33] = sum;
    }
    __syncthreads();
    if (tid < HD) {
        const float o = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
        const float l = red[0][33] + red[1][33] + red[2][33] + red[3][33];
        out_rows[(int64_t)b * d + h * HD + tid] = o / l; // l == 0 -> NaN like torch
    }
}

// ------------------------------------------------------------------ layer norm
// y = LN(z; g1, b1); if (c) y = LN(y + c; g2, b2).  One wave per row, d <= 512.
__global__ void __launch_bounds__(256) k_ln(const float *__restrict__ z, const float *__restrict__ g1,
                                            const float *__restrict__ b1, const float *__restrict__ c,
                                            const float *__restrict__ g2, const float *__restrict__ b2,
                                            float *__restrict__ y, int rows, int d, const int32_t *__restrict__ m_dev) {
    int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (row >= rows || (m_dev && row >= m_dev[0])) return;
    const float *zr = z + (int64_t)row * d;
    float v[8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int col = lane + 64 * i;
        v[i] = (col < d) ? zr[col] : 0.f;
        s += v[i];
    }
    const float invd = 1.0f / (float)d;
    float mu = wave_sum(s) * invd;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int col = lane + 64 * i;
        float t = (col < d) ? v[i] - mu : 0.f;
        q += t * t;
    }
    float rstd = 1.0f / sqrtf(wave_sum(q) * invd + 1e-5f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int col = lane + 64 * i;
        if (col < d) v[i] = (v[i] - mu) * rstd * g1[col] + b1[col];
    }
    if (c) {
        s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int col = lane + 64 * i;
            if (col < d) {
                v[i] += c[col];
                s += v[i];
            }
        }
        mu = wave_sum(s) * invd;
        q = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int col = lane + 64 * i;
            float t = (col < d) ? v[i] - mu : 0.f;
            q += t * t;
        }
        rstd = 1.0f / sqrtf(wave_sum(q) * invd + 1e-5f);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int col = lane + 64 * i;
            if (col < d) v[i] = (v[i] - mu) * rstd * g2[col] + b2[col];
        }
    }
    float *yr = y + (int64_t)row * d;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int col = lane + 64 * i;
        if (col < d) yr[col] = v[i];
    }
}

__global__ void k_gather_rows(const float *__restrict__ x, const int32_t *__restrict__ pos, float *__restrict__ out,
                              int B, int L, int d) {
    int b = blockIdx.x;
    int p = pos[b];
    if (p < 0) p = 0;
    if (p >= L) p = L - 1;
    const float *src = x + ((int64_t)b * L + p) * d;
    for (int c = threadIdx.x; c < d; c += blockDim.x) out[(int64_t)b * d + c] = src[c];
}

// ------------------------------------------------------------------ host side
static int g_lin_bk = 16, g_ln_bk = 16; // K-slab depth (tools/gemm_lab.hip flips these to compare 16 vs 32)
static int launch_linear(irs_ctx *ctx, const float *X, const float *W, const float *bias, const float *R, float *Y,
                         int M, int N, int K, bool relu, hipStream_t s, const float *g1 = nullptr,
                         const float *b1 = nullptr, const float *c = nullptr, const float *g2 = nullptr,
                         const float *b2 = nullptr, const float *Rf = nullptr, float *Yf = nullptr,
                         const int32_t *m_dev = nullptr, const float *Xf = nullptr) {
    LinArgs a{X, W, bias, R, Y, M, N, K, relu ? 1 : 0, g1, b1, c, g2, b2, Rf, Yf, m_dev, 0, Xf};
    if (Xf && !(K <= 128 && K % 32 == 0 && M > 2048)) {
        if (ctx) snprintf(ctx->err, sizeof(ctx->err), "launch_linear: fragment-major X needs K <= 128, K %% 32 == 0, M > 2048");
        return IRS_E_INVALID;
    }
    if (ctx) irs_prof_begin(ctx, IRS_PROF_LINEAR, s);
    auto al16 = [](const void *p) { return p == nullptr || (((uintptr_t)p) & 15) == 0; };
    if ((M <= 2048 || (M <= 8192 && g1 == nullptr && Xf == nullptr && N >= 256)) && K <= 256) { // latency path: one wave per 32x32 tile, operands straight from L2
        // (wide outputs keep it up to 8192 rows: C5's 6400 rows x d = 256 give the 128 x 128 tiling 100-300 workgroups)
        if (g1 != nullptr) hipLaunchKernelGGL(k_linear_small<true>, dim3(1, (M + 31) / 32), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(k_linear_small<false>, dim3((N + 127) / 128, (M + 31) / 32), dim3(256), 0, s, a);
    } else if (g1 != nullptr && N > LIN_BN) { // fused residual + LayerNorm, rows of up to 256 values (8 tiles per token)
        if (N > 2 * LIN_BN || Xf || Rf || Yf) {
            if (ctx) snprintf(ctx->err, sizeof(ctx->err), "launch_linear: LayerNorm epilogue needs N <= 256 (and row-major operands above 128)");
            return IRS_E_INVALID;
        }
        dim3 grid((M + LIN_BM - 1) / LIN_BM);
        const bool full = (N == 2 * LIN_BN) && (K % 16 == 0) && al16(X) && al16(W) && al16(R) && al16(Y);
        if (full) hipLaunchKernelGGL((k_linear_ln<16, true, false, 8>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((k_linear_ln<16, false, false, 8>), grid, dim3(256), 0, s, a);
    } else if (g1 != nullptr) { // fused residual + LayerNorm: whole rows per wave (N <= 128)
        dim3 grid((M + LIN_BM - 1) / LIN_BM);
        const int bk = g_ln_bk;
        const bool full = (N == LIN_BN) && (K % bk == 0) && al16(X) && al16(W) && al16(R) && al16(Y) && al16(Rf) && al16(Yf);
        if (Xf && !(full && al16(Xf))) {
            if (ctx) snprintf(ctx->err, sizeof(ctx->err), "launch_linear: fragment-major X needs the aligned N == 128 case");
            return IRS_E_INVALID;
        }
        if (bk == 16) {
            if (Xf) hipLaunchKernelGGL((k_linear_ln<16, true, true>), grid, dim3(256), 0, s, a);
            else if (full) hipLaunchKernelGGL((k_linear_ln<16, true, false>), grid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((k_linear_ln<16, false, false>), grid, dim3(256), 0, s, a);
        } else {
            if (Xf) hipLaunchKernelGGL((k_linear_ln<32, true, true>), grid, dim3(256), 0, s, a);
            else if (full) hipLaunchKernelGGL((k_linear_ln<32, true, false>), grid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((k_linear_ln<32, false, false>), grid, dim3(256), 0, s, a);
        }
    } else {
        const int bk = g_lin_bk;
        static int n_cu = 0;
        if (n_cu == 0) {
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess ||
                hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
                n_cu = 256;
        }
        a.slots = n_cu * (bk == 16 ? 4 : 2); // matches the kernels' launch bounds
        const int ntm = (M + LIN_BM - 1) / LIN_BM, ntn = (N + LIN_BN - 1) / LIN_BN;
        // exact unit count when M is known here, its upper bound over any device-side M otherwise
        dim3 grid(m_dev ? ntm + (a.slots - 1) * (ntn - 1) : (ntm / a.slots) * a.slots + (ntm % a.slots) * ntn);
        const bool full = (K % bk == 0) && al16(X) && al16(W);
        if (Xf && !(full && al16(Xf))) {
            if (ctx) snprintf(ctx->err, sizeof(ctx->err), "launch_linear: fragment-major X needs aligned operands");
            return IRS_E_INVALID;
        }
        if (bk == 16) {
            if (Xf) hipLaunchKernelGGL((k_linear<true, 16, true>), grid, dim3(256), 0, s, a);
            else if (full) hipLaunchKernelGGL((k_linear<true, 16, false>), grid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((k_linear<false, 16, false>), grid, dim3(256), 0, s, a);
        } else {
            if (Xf) hipLaunchKernelGGL((k_linear<true, 32, true>), grid, dim3(256), 0, s, a);
            else if (full) hipLaunchKernelGGL((k_linear<true, 32, false>), grid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((k_linear<false, 32, false>), grid, dim3(256), 0, s, a);
        }
    }
    if (ctx) {
        irs_prof_end(ctx, IRS_PROF_LINEAR, s, 2.0 * M * (double)N * K,
                     4.0 * ((double)M * K + (double)N * K + (double)M * N * (R ? 2 : 1)));
        IRS_CHECK_HIP(ctx, hipGetLastError());
    }
    return IRS_OK;
}

static int g_attn16 = 1; // 16-query-block attention kernel for head dim 32 (0: k_attn_mfma everywhere)

// attn16 is the only attention kernel that can write the fragment-major image the fused block kernel consumes
static bool attn16_ok(const irs_ctx *ctx, const float *qkv, const float *out) {
    const int L = ctx->dims.max_len, d = ctx->dims.d, H = ctx->dims.n_heads;
    return g_attn16 && d % H == 0 && d / H == 32 && L <= 256 && d % 4 == 0 && ((((uintptr_t)qkv) | ((uintptr_t)out)) & 15) == 0;
}

static int launch_attn(irs_ctx *ctx, const float *qkv, const int64_t *seq, const float *r_u, float *out, int B,
                       hipStream_t s, const int32_t *off = nullptr, const int32_t *cnt = nullptr,
                       const int32_t *tok_row = nullptr, bool frag_out = false, bool kv_planes = false) {
    const int L = ctx->dims.max_len, d = ctx->dims.d, H = ctx->dims.n_heads, hd = d / H;
    const int HDP = hd <= 8 ? 8 : hd <= 16 ? 16 : hd <= 32 ? 32 : 64;
    const int Lp = ((L + 31) / 32) * 32, VW = HDP < 32 ? 32 : HDP;
    size_t lds = (size_t)L * (HDP + 1) * 4 + (size_t)L * VW * 4 + 64;
    const bool v4 = (hd % 4 == 0) && (d % 4 == 0) && ((((uintptr_t)qkv) & 15) == 0) && ((((uintptr_t)out) & 15) == 0);
    dim3 grid(H, B);
    const int mm = ctx->dims.mask_mode;
    irs_prof_begin(ctx, IRS_PROF_ATTN, s);
    if (frag_out && !attn16_ok(ctx, qkv, out)) IRS_FAIL(ctx, IRS_E_UNSUPPORTED, "fragment-major attention output needs the head-dim-32 kernel");
    if (hd == 32 && v4 && L <= 256 && g_attn16) {
        int S16 = (L + 7) & ~7; // attn16_vstride
        if ((S16 & 15) != 8) S16 += 8;
        const size_t lds16 = (size_t)32 * S16 * 4 + (size_t)((L + 15) & ~15) * 32 * 4 + 64;
        const size_t lds16d = (size_t)2 * ((L + 15) & ~15) * 32 * 4 + 64; // LDS-DMA form: V row-major like K
        if (H * B <= 64) grid.z = (((L + 15) / 16) + 3) / 4; // latency path: one query block per wave
        if (kv_planes) { // K / V arrive as float16 plane pairs (k_block_x6's tail): the split-float16 attention
            if (grid.z != 1) IRS_FAIL(ctx, IRS_E_STATE, "plane-format K / V on the latency path");
            if (tok_row)
                hipLaunchKernelGGL((k_attn16h<16, true, 4>), grid, dim3(256), lds16d, s, qkv, seq, r_u, out, L, d, mm, off, cnt,
                                   ctx->seq_padq, frag_out ? 1 : 0);
            else
                hipLaunchKernelGGL((k_attn16h<16, false, 4>), grid, dim3(256), lds16d, s, qkv, seq, r_u, out, L, d, mm, off, cnt,
                                   nullptr, frag_out ? 1 : 0);
        } else if (g_attn16 == 2) { // (lab A/B: the register-staged fill with the transposed V^T image)
            if (tok_row)
                hipLaunchKernelGGL((k_attn16<16, true, false>), grid, dim3(256), lds16, s, qkv, seq, r_u, out, L, d, mm, off, cnt,
                                   ctx->seq_padq, frag_out ? 1 : 0, H);
            else
                hipLaunchKernelGGL((k_attn16<16, false, false>), grid, dim3(256), lds16, s, qkv, seq, r_u, out, L, d, mm, off, cnt,
                                   nullptr, frag_out ? 1 : 0, H);
        } else if (tok_row)
            hipLaunchKernelGGL((k_attn16<16, true, true>), grid, dim3(256), lds16d, s, qkv, seq, r_u, out, L, d, mm, off, cnt,
                               ctx->seq_padq, frag_out ? 1 : 0, H);
        else
            hipLaunchKernelGGL((k_attn16<16, false, true>), grid, dim3(256), lds16d, s, qkv, seq, r_u, out, L, d, mm, off, cnt,
                               nullptr, frag_out ? 1 : 0, H);
        irs_prof_end(ctx, IRS_PROF_ATTN, s, 2.0 * B * (double)H * L * L * hd, 4.0 * 4.0 * B * (double)L * d);
        IRS_CHECK_HIP(ctx, hipGetLastError());
        return IRS_OK;
    }
#define A_(HDP_)                                                                                                   \
    do {                                                                                                           \
        if (v4) hipLaunchKernelGGL((k_attn_mfma<HDP_, true>), grid, dim3(256), lds, s, qkv, seq, r_u, out, L, d, hd, mm, off, cnt, tok_row); \
        else hipLaunchKernelGGL((k_attn_mfma<HDP_, false>), grid, dim3(256), lds, s, qkv, seq, r_u, out, L, d, hd, mm, off, cnt, tok_row);   \
    } while (0)
    switch (HDP) {
    case 8: A_(8); break;
    case 16: A_(16); break;
    case 32: A_(32); break;
    default: A_(64); break;
    }
#undef A_
    irs_prof_end(ctx, IRS_PROF_ATTN, s, 2.0 * B * (double)H * L * L * hd, 4.0 * 4.0 * B * (double)L * d);
    IRS_CHECK_HIP(ctx, hipGetLastError());
    return IRS_OK;
}

int irs_launch_pif(irs_ctx *ctx, const int64_t *user, int B, float *r_u, hipStream_t s) {
    if (ctx->dims.mask_mode != IRS_MASK_IRN || !ctx->user_emb) {
        hipLaunchKernelGGL(k_fill, dim3((B + 255) / 256), dim3(256), 0, s, r_u, 0.f, B);
    } else {
        hipLaunchKernelGGL(k_pif, dim3((B + 255) / 256), dim3(256), 0, s, user, ctx->user_emb, ctx->um_w, ctx->um_b,
                           r_u, B, ctx->dims.u_dim, ctx->dims.n_user);
    }
    IRS_CHECK_HIP(ctx, hipGetLastError());
    return IRS_OK;
}

int irs_launch_cross_const(irs_ctx *ctx, hipStream_t s) {
    const int d = ctx->dims.d;
    for (int l = 0; l < ctx->dims.n_layers; ++l) {
        const irs_layer_w &w = ctx->layer[l];
        hipLaunchKernelGGL(k_cross_const, dim3((d + 63) / 64), dim3(64), 0, s, w.ca_out_w, w.ca_in_b, w.ca_out_b,
                           ctx->c_l + (size_t)l * d, d);
    }
    IRS_CHECK_HIP(ctx, hipGetLastError());
    return IRS_OK;
}

// fragment-packed weight copies for k_block_small16: [n_layers][Wo | W1 | W2] then [n_layers][Win]
// split-bf16 fused layer kernel (k_block_x6): one 768 KB step stream per layer that has a successor (d = 128, F = 256,
// head dim 32: the shapes k_block<true, true> serves)
static bool x6_shape(const irs_ctx *ctx) {
    return (ctx->dims.d == 128 || ctx->dims.d == 256) && ctx->dims.ffn_dim == 256 && ctx->dims.d / ctx->dims.n_heads == 32 &&
           ctx->dims.n_layers >= 2;
}
static size_t x6_layer_b(const irs_ctx *ctx, int npl = 3) { return x6_layer_bytes(ctx->dims.d / 32, npl); }
// the float16 two-plane streams (IRS_GEMM_H3) follow the bf16 three-plane ones in the arena
static const uint4 *x6_stream(const irs_ctx *ctx, int npl, int layer) {
    const uint4 *base = ctx->w_x6;
    if (npl == 2) base += (size_t)ctx->dims.n_layers * (x6_layer_b(ctx, 3) / 16);
    return base + (size_t)layer * (x6_layer_b(ctx, npl) / 16);
}
// streams 0 .. n_layers - 2: layer l's out-projection / FFN and layer l + 1's q | k | v; stream n_layers - 1: layer 0's
// q | k | v alone (the embed kernel's; its other blocks are zero and never fetched)
static bool seq_shape(const irs_ctx *ctx) {
    const irs_dims &D = ctx->dims;
    return D.d == 128 && D.ffn_dim == 256 && D.n_heads == 4 && D.max_len <= 256 && D.n_layers > 1;
}
static size_t x6_streams_bytes(const irs_ctx *ctx) { return (size_t)ctx->dims.n_layers * (x6_layer_b(ctx, 3) + x6_layer_b(ctx, 2)); }
// (+ the sequence-resident kernel's packed parameter vectors: [n_layers][X6_SEQ_VECS] floats behind the streams)
size_t irs_x6_bytes(const irs_ctx *ctx) {
    return x6_shape(ctx) ? x6_streams_bytes(ctx) + (seq_shape(ctx) ? (size_t)ctx->dims.n_layers * X6_SEQ_VECS * 4 : 0) : 0;
}
static const float *seq_vecpack(const irs_ctx *ctx) {
    return reinterpret_cast<const float *>(reinterpret_cast<const char *>(ctx->w_x6) + x6_streams_bytes(ctx));
}
// layer l's parameter vectors in k_block_x6's LDS order: b1[256] b2 g3 b3 b_in(l + 1)[384] b_o g1 b1n c_l g2 b2n b_in(l)[384]
__global__ void __launch_bounds__(256) k_pack_seqvec(const float *b1, const float *b2, const float *g3, const float *b3,
                                                     const float *bin_next, const float *bo, const float *g1, const float *b1n,
                                                     const float *c, const float *g2, const float *b2n, const float *bq,
                                                     float *__restrict__ out) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= X6_SEQ_VECS) return;
    float v;
    if (e < 256) v = b1[e];
    else if (e < 384) v = b2[e - 256];
    else if (e < 512) v = g3[e - 384];
    else if (e < 640) v = b3[e - 512];
    else if (e < 1024) v = bin_next ? bin_next[e - 640] : 0.f;
    else if (e < 1152) v = bo ? bo[e - 1024] : 0.f;
    else if (e < 1280) v = g1[e - 1152];
    else if (e < 1408) v = b1n[e - 1280];
    else if (e < 1536) v = c[e - 1408];
    else if (e < 1664) v = g2[e - 1536];
    else if (e < 1792) v = b2n[e - 1664];
    else v = bq[e - 1792];
    out[e] = v;
}
int irs_launch_pack_x6(irs_ctx *ctx, hipStream_t s) {
    if (!ctx->w_x6) return IRS_OK;
    const int nl = ctx->dims.n_layers, NT = ctx->dims.d / 32;
    for (int npl = 3; npl >= 2; --npl) { // bf16 three-plane streams, then float16 two-plane streams
        const dim3 pgrid(x6_nstep(NT) * 8 * npl * 64 / 256);
        auto pack = [&](const float *Wo, const float *W1, const float *W2, const float *Win, uint4 *out) {
            if (NT == 8 && npl == 3) hipLaunchKernelGGL((k_pack_x6<8, 3>), pgrid, dim3(256), 0, s, Wo, W1, W2, Win, out);
            else if (NT == 8) hipLaunchKernelGGL((k_pack_x6<8, 2>), pgrid, dim3(256), 0, s, Wo, W1, W2, Win, out);
            else if (npl == 3) hipLaunchKernelGGL((k_pack_x6<4, 3>), pgrid, dim3(256), 0, s, Wo, W1, W2, Win, out);
            else hipLaunchKernelGGL((k_pack_x6<4, 2>), pgrid, dim3(256), 0, s, Wo, W1, W2, Win, out);
        };
        for (int l = 0; l + 1 < nl; ++l) {
            const irs_layer_w &w = ctx->layer[l];
            pack(w.sa_out_w, w.l1_w, w.l2_w, ctx->layer[l + 1].sa_in_w, const_cast<uint4 *>(x6_stream(ctx, npl, l)));
        }
        pack(nullptr, nullptr, nullptr, ctx->layer[0].sa_in_w, const_cast<uint4 *>(x6_stream(ctx, npl, nl - 1)));
    }
    if (seq_shape(ctx)) { // (behind irs_launch_cross_const: c_l exists)
        for (int l = 0; l < nl; ++l) {
            const irs_layer_w &w = ctx->layer[l];
            hipLaunchKernelGGL(k_pack_seqvec, dim3((X6_SEQ_VECS + 255) / 256), dim3(256), 0, s, w.l1_b, w.l2_b, w.n3_w, w.n3_b,
                               l + 1 < nl ? ctx->layer[l + 1].sa_in_b : nullptr, w.sa_out_b, w.n1_w, w.n1_b, ctx->c_l + (size_t)l * ctx->dims.d,
                               w.n2_w, w.n2_b, w.sa_in_b, const_cast<float *>(seq_vecpack(ctx)) + (size_t)l * X6_SEQ_VECS);
        }
    }
    IRS_CHECK_HIP(ctx, hipGetLastError());
    return IRS_OK;
}
// ---- float16 range bounds for IRS_GEMM_H3 and the float16 V planes (finalisation).  Float16 planes overflow at 65504; the
// operands are weights, embedded tokens, LayerNorm outputs (+ c_l), the FFN's hidden activations and the attention output / V
// rows.  Seven statistics of the bound weights give a bound for each (LayerNorm output: |z_i| <= sqrt(d) per normalised
// component, so |y_i| <= sqrt(d) max|gamma| + max|beta|, ||y|| <= sqrt(d) times that; a product row: ||in|| ||W_row|| + |b|).
// Non-negative floats order like their bit patterns: atomicMax on the bits.
__global__ void __launch_bounds__(256) k_absmax(const float *__restrict__ p, size_t n, unsigned int *__restrict__ out) {
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = fmaxf(m, fabsf(p[i]));
    if (!(m == m)) m = INFINITY; // a NaN weight fails the bound
    for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}
__global__ void __launch_bounds__(256) k_rownorm_max(const float *__restrict__ W, int rows, int cols, unsigned int *__restrict__ out) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    float q = 0.f;
    for (int c = lane; c < cols; c += 64) q = __fmaf_rn(W[(size_t)r * cols + c], W[(size_t)r * cols + c], q);
    for (int o = 32; o; o >>= 1) q += __shfl_xor(q, o);
    q = sqrtf(q);
    if (!(q == q)) q = INFINITY;
    if (lane == 0) atomicMax(out, __float_as_uint(q));
}
// statistics -> stats[0..7] (device, zeroed here): max|E|, max|pe|, max LayerNorm |gamma|, max LayerNorm |beta|, max|c_l|,
// max row norm of W1 and of the V rows of W_in, max of every other |weight| and |bias| that becomes or feeds an operand
int irs_launch_h3_range(irs_ctx *ctx, float *stats, hipStream_t s) {
    const irs_dims &D = ctx->dims;
    const int d = D.d, F = D.ffn_dim, nl = D.n_layers;
    unsigned int *u = reinterpret_cast<unsigned int *>(stats);
    IRS_CHECK_HIP(ctx, hipMemsetAsync(stats, 0, 8 * sizeof(float), s));
    auto amax = [&](const float *p, size_t n, int slot) {
        if (!p || !n) return;
        const int grid = (int)std::min<size_t>((n + 4095) / 4096, 2048);
        hipLaunchKernelGGL(k_absmax, dim3(grid), dim3(256), 0, s, p, n, u + slot);
    };
    auto rmax = [&](const float *W, int rows, int cols, int slot) {
        if (W) hipLaunchKernelGGL(k_rownorm_max, dim3((rows + 3) / 4), dim3(256), 0, s, W, rows, cols, u + slot);
    };
    amax(ctx->item_emb, (size_t)(D.n_item + 1) * d, 0);
    amax(ctx->pe, (size_t)D.max_len * d, 1);
    amax(ctx->c_l, (size_t)nl * d, 4);
    for (int l = 0; l < nl; ++l) {
        const irs_layer_w &w = ctx->layer[l];
        amax(w.n1_w, d, 2), amax(w.n2_w, d, 2), amax(w.n3_w, d, 2);
        amax(w.n1_b, d, 3), amax(w.n2_b, d, 3), amax(w.n3_b, d, 3);
        rmax(w.l1_w, F, d, 5);
        if (seq_shape(ctx)) rmax(w.sa_in_w, 3 * d, d, 6); // (the sequence-resident attention splits q and k rows too)
        else rmax(w.sa_in_w + (size_t)2 * d * d, d, d, 6);
        amax(w.sa_out_w, (size_t)d * d, 7), amax(w.l1_w, (size_t)F * d, 7), amax(w.l2_w, (size_t)d * F, 7), amax(w.sa_in_w, (size_t)3 * d * d, 7);
        amax(w.l1_b, F, 7), amax(w.sa_in_b, (size_t)3 * d, 7);
    }
    IRS_CHECK_HIP(ctx, hipGetLastError());
    return IRS_OK;
}
// the largest operand magnitude the statistics allow (host)
float irs_h3_operand_bound(const irs_ctx *ctx, const float *st) {
    const float sd = sqrtf((float)ctx->dims.d);
    const float a0 = st[0] * sd + st[1];                 // embedded token
    const float ln = sd * st[2] + st[3] + st[4];         // a LayerNorm output (+ c_l)
    const float xin = sd * fmaxf(a0, ln);                // norm of a layer's input row
    float v = xin * st[6] + st[7];                       // a V row (and the attention output, a convex combination of V rows)
    if (seq_shape(ctx)) v *= SEQ_KQ_SCALE;               // ... and a q or k row (st[6] then covers all of W_in's rows)
    const float h = sd * ln * st[5] + st[7];             // a hidden activation
    return fmaxf(fmaxf(fmaxf(a0, ln), fmaxf(v, h)), st[7] * x6_wscale(2)); // (the weight planes hold 2^8 x the weights)
}
static constexpr int X6_LDS_BYTES = x6_lds_bytes(4);
// one launcher for every instantiation of the fused layer kernel: QP0 (k | v-only tail), EMBED, NT (4: d = 128, 8: d = 256),
// NPL (3: bf16 six-product, 2: float16 three-product), NW (4; 1 = the lab's one-wave workgroups at NT = 8, bf16 only)
template <int QP0, int NW, bool EMBED, int NT, int NPL>
static void x6_launch_one(int rows, const BlockX6Args &xa, hipStream_t s) {
    auto kern = k_block_x6<QP0, NW, EMBED, NT, NPL>;
    constexpr int lds = x6_lds_bytes(NT, NPL);
    IRS_ONCE_PER_DEVICE((void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(kern, dim3((rows + 32 * NW - 1) / (32 * NW)), dim3(64 * NW), lds, s, xa);
}
#ifdef X6_STAMP
static unsigned long long *g_seq_stamps = nullptr;
extern "C" void *irs_lab_seq_stamps() { return g_seq_stamps; }
#endif
// the sequence-resident form: grid = an upper bound of the plan's workgroups (the kernel reads the count), eight waves, 160 KB of LDS
static void x6_launch_seq(int wg_cap, const BlockX6Args &xa, hipStream_t s) {
    auto kern = k_block_x6<3, 8, false, 4, 2, true>;
    constexpr int lds = x6_seq_lds_bytes();
    IRS_ONCE_PER_DEVICE((void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
#ifdef X6_STAMP
    BlockX6Args xs = xa; // (lab) 24 counters per wave
    if (!g_seq_stamps) (void)hipMalloc(&g_seq_stamps, (size_t)8192 * 8 * 24 * 8);
    (void)hipMemsetAsync(g_seq_stamps, 0, (size_t)8192 * 8 * 24 * 8, s);
    xs.stamps = g_seq_stamps;
    hipLaunchKernelGGL(kern, dim3(wg_cap < 8192 ? wg_cap : 8192), dim3(512), lds, s, xs);
#else
    hipLaunchKernelGGL(kern, dim3(wg_cap), dim3(512), lds, s, xa);
#endif
}
static void x6_launch(int qp0, bool embed, int nt, int npl, int rows, const BlockX6Args &xa, hipStream_t s) {
#define X6_L(Q_, E_) do {                                                                                                 \
        if (nt == 8 && npl == 2) x6_launch_one<Q_, 4, E_, 8, 2>(rows, xa, s);                                             \
        else if (nt == 8) x6_launch_one<Q_, 4, E_, 8, 3>(rows, xa, s);                                                    \
        else if (npl == 2) x6_launch_one<Q_, 4, E_, 4, 2>(rows, xa, s);                                                   \
        else x6_launch_one<Q_, X6_NW, E_, 4, 3>(rows, xa, s);                                                             \
    } while (0)
    if (embed) X6_L(0, true);
    else if (qp0) X6_L(1, false);
    else X6_L(0, false);
#undef X6_L
}

size_t irs_small_frag_floats(const irs_ctx *ctx) {
    const int d = ctx->dims.d, F = ctx->dims.ffn_dim;
    if (small_any_shape(d, F)) return (size_t)ctx->dims.n_layers * small_any_layer_floats(d, F);
    if (small_wide_shape(d, F)) return (size_t)ctx->dims.n_layers * (small_wide_layer_floats(d, F) + (size_t)3 * d * d);
    if (d != 128 || F != 256) return 0;
    return (size_t)ctx->dims.n_layers * (SMALL_WF_LAYER + SMALL_WF_WIN);
}

int irs_launch_pack_small(irs_ctx *ctx, hipStream_t s) {
    if (!ctx->w_frag16) return IRS_OK;
    const int nl = ctx->dims.n_layers;
    if (small_any_shape(ctx->dims.d, ctx->dims.ffn_dim)) {
        const int d = ctx->dims.d, F = ctx->dims.ffn_dim, dp = small_any_dp(d), Fp = (F + 15) & ~15, Qp = (3 * d + 15) & ~15;
        for (int l = 0; l < nl; ++l) {
            const irs_layer_w &w = ctx->layer[l];
            float *o = ctx->w_frag16 + (size_t)l * small_any_layer_floats(d, F);
            auto pk = [&](const float *W, float *out, int N, int K, int Np, int Kp) {
                hipLaunchKernelGGL(k_pack_frag16_any, dim3((Np * Kp / 4 + 255) / 256), dim3(256), 0, s, W, out, N, K, Np, Kp);
            };
            pk(w.sa_out_w, o, d, d, dp, dp);
            pk(w.l1_w, o + (size_t)dp * dp, F, d, Fp, dp);
            pk(w.l2_w, o + (size_t)dp * dp + (size_t)Fp * dp, d, F, dp, Fp);
            pk(w.sa_in_w, o + small_any_win_off(d, F), 3 * d, d, Qp, dp);
        }
        IRS_CHECK_HIP(ctx, hipGetLastError());
        return IRS_OK;
    }
    if (small_wide_shape(ctx->dims.d, ctx->dims.ffn_dim)) { // [n_layers][Wo | W1 | W2], then [n_layers][Win]
        const int d = ctx->dims.d, F = ctx->dims.ffn_dim;
        const size_t lf = small_wide_layer_floats(d, F);
        auto pk = [&](const float *W, float *out, int N, int K) {
            hipLaunchKernelGGL(k_pack_frag16, dim3((N * K / 4 + 255) / 256), dim3(256), 0, s, W, out, N, K);
        };
        for (int l = 0; l < nl; ++l) {
            const irs_layer_w &w = ctx->layer[l];
            float *o = ctx->w_frag16 + (size_t)l * lf;
            pk(w.sa_out_w, o, d, d);
            pk(w.l1_w, o + (size_t)d * d, F, d);
            pk(w.l2_w, o + (size_t)d * d + (size_t)F * d, d, F);
            pk(w.sa_in_w, ctx->w_frag16 + (size_t)nl * lf + (size_t)l * 3 * d * d, 3 * d, d);
        }
        IRS_CHECK_HIP(ctx, hipGetLastError());
        return IRS_OK;
    }
    for (int l = 0; l < nl; ++l) {
        const irs_layer_w &w = ctx->layer[l];
        float *o = ctx->w_frag16 + (size_t)l * SMALL_WF_LAYER;
        hipLaunchKernelGGL(k_pack_frag16, dim3(16), dim3(256), 0, s, w.sa_out_w, o + SMALL_WF_WO, 128, 128);
        hipLaunchKernelGGL(k_pack_frag16, dim3(32), dim3(256), 0, s, w.l1_w, o + SMALL_WF_W1, 256, 128);
        hipLaunchKernelGGL(k_pack_frag16, dim3(32), dim3(256), 0, s, w.l2_w, o + SMALL_WF_W2, 128, 256);
        hipLaunchKernelGGL(k_pack_frag16, dim3(48), dim3(256), 0, s, w.sa_in_w, ctx->w_frag16 + (size_t)nl * SMALL_WF_LAYER + (size_t)l * SMALL_WF_WIN, 384, 128);
    }
    IRS_CHECK_HIP(ctx, hipGetLastError());
    return IRS_OK;
}

int irs_launch_decode(irs_ctx *ctx, const int64_t *seq, const int64_t *user, int B, float *x_out, const int32_t *pos,
                      float *xrows, float *r_u_out, hipStream_t s) {
    const int L = ctx->dims.max_len, d = ctx->dims.d, F = ctx->dims.ffn_dim;
    const int rows = B * L;
    int rc;
    // rows-only decode of a few sequences: the plan kernel also computes r_u (and hands the step counter over)
    const bool small_plan = (x_out == nullptr) && pos && xrows && L >= 4 && B <= 64;
    const bool rows_only = (x_out == nullptr) && pos && xrows && L >= 4;
    if (ctx->step_pair && !small_plan) IRS_FAIL(ctx, IRS_E_STATE, "merged path step needs the single-workgroup plan kernel");
    if (!small_plan && (rc = irs_launch_pif(ctx, user, B, ctx->act_ru, s)) != IRS_OK) return rc;
    float *x = ctx->act_x, *y = ctx->act_y;
    // rows-only decode: the caller wants x[b, pos[b], :] only.  Then (1) the decoder runs on the PACKED
    // non-pad tokens (k_plan), every kernel clamping its row count to the device-side total, and (2) the
    // LAST layer is evaluated for the one consumed row per sequence (all earlier layers need every valid
    // row: they feed the next layer's keys and values).
    const int32_t *off = nullptr, *cnt = nullptr, *tok = nullptr, *qrow = nullptr, *m_dev = nullptr;
    // throughput shapes keep x / y ONLY in the fragment-major layout between the layers (see frag_index): the
    // LN-fused GEMMs write it, read their residual from it, and the QKV / FFN1 GEMMs load it as their X operand
    // The 16-token layer kernel's regime.  Measured crossover with the 128-token fragment-major path on C2 (L = 200):
    // 564 vs 720 us per path step at 128 users, 921 vs 797 at 256 (the big path has a ~700 us floor per step: one round
    // of 128-token tiles streams the weights serially through LDS whatever the number of tiles).
    const bool small_cfg = d == 128 && F == 256 && ctx->w_frag16 && rows <= SMALL_ROWS_MAX;
    // any other small shape: the generic fused layer tail.  Measured against the per-GEMM kernels (per path step):
    // default (d = 30, L = 60) 235 vs 582 us at 64 users, 859 vs 1055 at 1024; config 1 (d = 64, L = 50) 259 vs 653
    // at 64 users, 888 vs 1040 at 1024 -- ahead over the whole tested range
    const bool any_cfg = small_any_shape(d, F) && ctx->w_frag16 && rows <= 65536;
    // d = 256 (C4's decoder), rows-only decode of a throughput batch: the split-bf16 fused layer kernel at 8 accumulator tiles
    // per token (k_block_x6<.., NT = 8>) with fragment-major activations, like d = 128.  Everything else at d = 256 (full
    // decodes, small batches, IRS_GEMM_F32) keeps the per-GEMM float32 kernels.
    const bool x6d = d == 256 && F == 256 && ctx->use_x6 && ctx->w_x6 && rows_only && rows >= 32768 && ctx->dims.n_layers > 1 &&
                     attn16_ok(ctx, ctx->act_qkv, ctx->act_yf);
    // the split-precision layer kernels hand K / V to the attention as float16 plane pairs (k_attn16h) unless switched off
    // ONE predicate for the writers (embed kernel, layer kernel tail) and the reader (launch_attn: k_attn16h): it includes the
    // layer loop's own condition for the fused block + 16-query attention (fragment-major activations, head dim 32, L <= 256,
    // 16-byte aligned workspace), so a V section is never written as planes for an attention kernel that reads float32
    const bool frag = (d <= LIN_BN && d % 32 == 0 && rows > 2048 && !small_cfg && !any_cfg) || x6d;
    const bool fuse_block_cfg = frag && (d == 128 || x6d) && F == 256 && attn16_ok(ctx, ctx->act_qkv, ctx->act_yf);
    const bool kv_planes = fuse_block_cfg && ctx->use_attn_h3 && ctx->h3_ok && ctx->use_x6 && ctx->w_x6 && ctx->dims.n_layers > 1 &&
                           rows * (long long)ctx->dims.n_heads > 64 * (long long)L; // (not the z-split latency grid)
    // d = 256 below the throughput regime (C5's 32 beam windows, single users): the 16-token fused layer kernel for wide
    // models (k_block_small_wide) instead of ~8 per-GEMM launches per layer
    const bool wide_cfg = small_wide_shape(d, F) && ctx->w_frag16 && !x6d && rows < 32768;
    const size_t wide_lf = small_wide_layer_floats(d, F); // one wave (32 tokens) per workgroup: a few thousand tokens still reach every CU
    // one sequence (the reference IRN's own regime, and the latency metric's): self-attention runs inside the layer
    // kernel; q | k | v alternate between two buffers so that the last (rows-only) layer reads ctx->act_qkv
    const bool att_fused = small_cfg && rows_only && B == 1 && ctx->dims.n_heads == 4 && L <= 256 && ctx->act_qkv_b1 &&
                           ctx->dims.n_layers > 1;
    auto qkv_in = [&](int l) { return (!att_fused || ((ctx->dims.n_layers - 1 - l) & 1) == 0) ? ctx->act_qkv : ctx->act_qkv_b1; };
    float *xf = ctx->act_xf, *yf = ctx->act_yf;
    // the sequence-resident layer kernel (round 5, opt-in: irs_set_decoder_seq / IRS_DECODER_SEQ=1): the throughput shape of
    // config 2 / 3 on float16 planes; layers 0 .. n_layers - 2 are ONE launch each (q | k | v, attention and the layer body; K / V
    // stay in LDS), the rows-only last layer runs as before on the k | v rows the last of them writes
    // use_seq: 0 never, 1 whenever the shape allows, 2 (default) where it measured ahead: >= SEQ_AUTO_MIN_SEQS sequences (one
    // workgroup per CU: below ~4 rounds of workgroups the last, partly filled round costs more than the fusion saves --
    // profiles/r05/seq_sizes.txt: 6-layer decode 0.98 vs 1.06 ms at 1024 users, 3.83 vs 3.98 at 4096, but 0.89 vs 0.80 at 768)
    const bool seq_mode = (ctx->use_seq == 1 || (ctx->use_seq == 2 && B >= SEQ_AUTO_MIN_SEQS)) && rows_only && kv_planes && d == 128 &&
                          ctx->dims.n_heads == 4 && L <= 256 && !small_plan && ctx->tile_seq && ctx->use_x6 == IRS_GEMM_H3 && ctx->h3_ok &&
                          B <= 1024 * SEQ_PLAN_PER_THREAD;
    ctx->seq_last = seq_mode;
    if (rows_only) {
        const bool plan_in_embed = small_plan && B == 1 && L <= 256 && ctx->dims.n_layers > 1 && (att_fused || any_cfg);
        if (plan_in_embed) {
            // one sequence: the embedding kernel below derives the plan itself
        } else if (small_plan) {
            const bool pif = ctx->dims.mask_mode == IRS_MASK_IRN && ctx->user_emb;
            hipLaunchKernelGGL(k_plan_small, dim3(1), dim3(1024), 0, s, seq, pos, B, L, ctx->seq_cnt, ctx->seq_off, ctx->seq_qrow,
                               ctx->tok_row, ctx->seq_padq, ctx->m_dev, user, pif ? ctx->user_emb : nullptr, ctx->um_w, ctx->um_b,
                               ctx->act_ru, ctx->dims.u_dim, ctx->dims.n_user, ctx->step_pair);
        } else {
            hipLaunchKernelGGL(k_plan_count, dim3((B + 3) / 4), dim3(256), 0, s, seq, pos, B, L, ctx->seq_cnt, ctx->tile_seq);
            hipLaunchKernelGGL(k_plan_scan, dim3(1), dim3(1024), 0, s, ctx->seq_cnt, B, ctx->seq_off, ctx->m_dev);
            hipLaunchKernelGGL(k_plan_fill, dim3((B + 3) / 4), dim3(256), 0, s, seq, pos, B, L, ctx->seq_off, ctx->seq_qrow,
                               ctx->tok_row, ctx->seq_padq);
        }
        off = ctx->seq_off;
        cnt = ctx->seq_cnt;
        tok = ctx->tok_row;
        qrow = ctx->seq_qrow;
        m_dev = ctx->m_dev;
    }
    bool qkv0_done = false;
    int l_begin = 0;
    if (seq_mode) {
        const int nl = ctx->dims.n_layers;
        hipLaunchKernelGGL(k_plan_seq, dim3(1), dim3(1024), 0, s, ctx->seq_cnt, ctx->seq_off, ctx->seq_qrow, B, ctx->tile_seq, ctx->tile_idx,
                           ctx->seq_row0, ctx->qrow_tile, ctx->n_wg_dev, B * SEQ_WG_TILES);
        {   // layers 0 .. nl - 2 in ONE launch (x resident in registers from layer to layer); the last of them writes the k | v rows
            BlockX6Args xa{};
            xa.Af = yf, xa.Rf = xf, xa.Xf = xf, xa.QKV = ctx->act_qkv, xa.M = rows, xa.m_dev = m_dev, xa.qkv_pass0 = 3;
            xa.seq_qrow = ctx->seq_qrow;
            xa.seq = seq, xa.L = L;
            xa.E = ctx->item_emb, xa.pe = ctx->pe, xa.tok_row = tok, xa.sqrtd = sqrtf((float)d), xa.n_item = ctx->dims.n_item; // (the embedding is the launch's prologue)
            xa.Wbase = x6_stream(ctx, 2, 0), xa.wstride = (long long)(x6_layer_b(ctx, 2) / 16), xa.n_lay = nl - 1, xa.nl_total = nl;
            xa.vecpack = seq_vecpack(ctx);
            xa.c = ctx->c_l; // (non-null: the second LayerNorm always runs)
            xa.tile_seq = ctx->tile_seq, xa.tile_qb = ctx->tile_idx, xa.seq_off = ctx->seq_off, xa.seq_cnt = ctx->seq_cnt;
            xa.seq_padq = ctx->seq_padq, xa.seq_row0 = ctx->seq_row0, xa.n_wg_dev = ctx->n_wg_dev, xa.r_u = ctx->act_ru;
            xa.mask_mode = ctx->dims.mask_mode;
            const double fl = (double)(nl - 1) * rows * (8.0 * d * d + 4.0 * d * F);
            irs_prof_begin(ctx, IRS_PROF_LINEAR, s);
            irs_prof_begin(ctx, IRS_PROF_LAYER, s);
            x6_launch_seq(B, xa, s);
            irs_prof_end(ctx, IRS_PROF_LINEAR, s, fl, (double)(nl - 1) * 3.0 * 4.0 * rows * (double)d);
            irs_prof_end(ctx, IRS_PROF_LAYER, s, fl, (double)(nl - 1) * 3.0 * 4.0 * rows * (double)d);
        }
        qkv0_done = true;
        l_begin = nl - 1;
    } else if (frag && (d == 128 || x6d) && ctx->dims.n_layers > 1) { // embed + layer 0's QKV in one kernel
        EmbedQkvArgs ea{seq, ctx->item_emb, ctx->pe, tok, m_dev, rows, L, sqrtf((float)d), ctx->dims.n_item, xf,
                        ctx->layer[0].sa_in_w, ctx->layer[0].sa_in_b, ctx->act_qkv};
        irs_prof_begin(ctx, IRS_PROF_LINEAR, s);
        if (ctx->use_x6 && ctx->w_x6) { // the same kernel on split-bf16 MFMAs: k_block_x6's q | k | v steps behind an embed prologue
            BlockX6Args xa{};
            const int npl = (ctx->use_x6 == IRS_GEMM_H3 && ctx->h3_ok) ? 2 : 3;
            xa.Wx = x6_stream(ctx, npl, ctx->dims.n_layers - 1);
            xa.bin = ctx->layer[0].sa_in_b, xa.Xf = xf, xa.QKV = ctx->act_qkv, xa.M = rows, xa.m_dev = m_dev;
            xa.seq = seq, xa.E = ctx->item_emb, xa.pe = ctx->pe, xa.tok_row = tok, xa.L = L, xa.sqrtd = sqrtf((float)d), xa.n_item = ctx->dims.n_item;
            xa.kv_planes = kv_planes ? 1 : 0;
            x6_launch(0, true, x6d ? 8 : 4, npl, rows, xa, s);
        } else
        hipLaunchKernelGGL(k_embed_qkv, dim3((rows + 127) / 128), dim3(256), 0, s, ea);
        irs_prof_end(ctx, IRS_PROF_LINEAR, s, 6.0 * rows * (double)d * d, 4.0 * 4.0 * rows * (double)d);
        qkv0_done = true;
    } else if (frag)
        hipLaunchKernelGGL(k_embed_frag, dim3((rows + 127) / 128), dim3(256), 0, s, seq, ctx->item_emb, ctx->pe, xf, tok, m_dev,
                           rows, L, d, sqrtf((float)d), ctx->dims.n_item);
    else if (rows_only && small_cfg && ctx->dims.n_layers > 1) {
        // latency path (the k_block_small16 regime): embed + layer 0's QKV in one 16-token kernel
        SmallEmbedArgs ea{seq, ctx->item_emb, ctx->pe, tok, m_dev, rows, L, sqrtf((float)d), ctx->dims.n_item, x,
                          ctx->w_frag16 + (size_t)ctx->dims.n_layers * SMALL_WF_LAYER, ctx->layer[0].sa_in_b, qkv_in(0)};
        irs_prof_begin(ctx, IRS_PROF_LINEAR, s);
        if (att_fused) {
            const bool pif = ctx->dims.mask_mode == IRS_MASK_IRN && ctx->user_emb;
            ea.pos = pos, ea.cnt = ctx->seq_cnt, ea.off = ctx->seq_off, ea.qrow = ctx->seq_qrow, ea.tok_out = ctx->tok_row;
            ea.padq = ctx->seq_padq, ea.mdev_out = ctx->m_dev, ea.user = user, ea.U = pif ? ctx->user_emb : nullptr;
            ea.uw = ctx->um_w, ea.ub = ctx->um_b, ea.r_u = ctx->act_ru, ea.ud = ctx->dims.u_dim, ea.n_user = ctx->dims.n_user;
            ea.step_pair = ctx->step_pair;
            hipLaunchKernelGGL(k_embed_qkv_small16<true>, dim3((rows + 15) / 16), dim3(256), 0, s, ea);
        } else
            hipLaunchKernelGGL(k_embed_qkv_small16<false>, dim3((rows + 15) / 16), dim3(256), 0, s, ea);
        irs_prof_end(ctx, IRS_PROF_LINEAR, s, 6.0 * rows * (double)d * d, 4.0 * 4.0 * rows * (double)d);
        qkv0_done = true;
    } else if (rows_only && any_cfg && ctx->dims.n_layers > 1) {
        SmallEmbedArgs ea{seq, ctx->item_emb, ctx->pe, tok, m_dev, rows, L, sqrtf((float)d), ctx->dims.n_item, x,
                          ctx->w_frag16 + small_any_win_off(d, F), ctx->layer[0].sa_in_b, ctx->act_qkv};
        irs_prof_begin(ctx, IRS_PROF_LINEAR, s);
        const dim3 grid((rows + 15) / 16);
        const bool plan1 = small_plan && B == 1 && L <= 256; // the plan of the one sequence is derived in the kernel
        if (plan1) {
            const bool pif = ctx->dims.mask_mode == IRS_MASK_IRN && ctx->user_emb;
            ea.pos = pos, ea.cnt = ctx->seq_cnt, ea.off = ctx->seq_off, ea.qrow = ctx->seq_qrow, ea.tok_out = ctx->tok_row;
            ea.padq = ctx->seq_padq, ea.mdev_out = ctx->m_dev, ea.user = user, ea.U = pif ? ctx->user_emb : nullptr;
            ea.uw = ctx->um_w, ea.ub = ctx->um_b, ea.r_u = ctx->act_ru, ea.ud = ctx->dims.u_dim, ea.n_user = ctx->dims.n_user;
            ea.step_pair = ctx->step_pair;
        }
#define E_(T_) do { if (plan1) hipLaunchKernelGGL((k_embed_qkv_small_any<T_, true>), grid, dim3(256), 0, s, ea, d); \
                    else hipLaunchKernelGGL((k_embed_qkv_small_any<T_, false>), grid, dim3(256), 0, s, ea, d); } while (0)
        switch (small_any_dp(d)) {
        case 32: E_(2); break;
        case 64: E_(4); break;
        default: E_(6); break;
        }
#undef E_
        irs_prof_end(ctx, IRS_PROF_LINEAR, s, 6.0 * rows * (double)d * d, 4.0 * 4.0 * rows * (double)d);
        qkv0_done = true;
    } else if (rows_only)
        hipLaunchKernelGGL(k_embed_packed, dim3((rows + 3) / 4), dim3(256), 0, s, seq, ctx->item_emb, ctx->pe, x, tok, m_dev,
                           L, d, sqrtf((float)d), ctx->dims.n_item);
    else
        hipLaunchKernelGGL(k_embed, dim3((rows + 3) / 4), dim3(256), 0, s, seq, ctx->item_emb, ctx->pe, x, rows, L, d,
                           sqrtf((float)d), ctx->dims.n_item);
    IRS_CHECK_HIP(ctx, hipGetLastError());
    if (r_u_out) IRS_CHECK_HIP(ctx, hipMemcpyAsync(r_u_out, ctx->act_ru, sizeof(float) * B, hipMemcpyDeviceToDevice, s));
    bool qkv_done = qkv0_done, q_split = seq_mode; // (seq_mode: the last fused layer wrote k | v only)
    const int32_t *qrow_f = seq_mode ? ctx->qrow_tile : qrow; // the consumed rows in the fragment-major image's own row order
    for (int l = l_begin; l < ctx->dims.n_layers; ++l) {
        const irs_layer_w &w = ctx->layer[l];
        const bool last_rows = rows_only && (l + 1 == ctx->dims.n_layers);
        // qkv = x W_in^T + b_in (already produced by the previous layer's fused feed-forward kernel where that ran)
        if (!qkv_done &&
            (rc = launch_linear(ctx, x, w.sa_in_w, w.sa_in_b, nullptr, ctx->act_qkv, rows, 3 * d, d, false, s, nullptr, nullptr,
                                nullptr, nullptr, nullptr, nullptr, nullptr, m_dev, frag ? xf : nullptr)))
            return rc;
        qkv_done = false;
        const float *cl = ctx->c_l + (size_t)l * d;
        if (last_rows) {
            float *ao_r = ctx->act_ao;                 // [B, d] attention output rows
            float *x_r = ctx->act_ao + (size_t)B * d;  // [B, d] residual rows x[b, pos[b]]
            float *y_r = ctx->act_ao + (size_t)2 * B * d;
            float *h_r = ctx->act_h;                   // [B, F]
            const float *q_r = nullptr;
            if (seq_mode) {
                // the sequence-resident launch ran this layer's q | k | v and the attention of every consumed token's block: the
                // residual row and the attention row come out of the fragment-major images by the tile-order row index
                hipLaunchKernelGGL(k_gather_rows_frag, dim3(B), dim3(64), 0, s, xf, qrow_f, x_r, d, 4);
                hipLaunchKernelGGL(k_gather_rows_frag, dim3(B), dim3(64), 0, s, yf, qrow_f, ao_r, d, 4);
            } else if (q_split) { // the previous layer's kernel wrote k | v only: queries for the B consumed rows here
                hipLaunchKernelGGL(k_gather_rows_frag, dim3(B), dim3(64), 0, s, xf, qrow_f, x_r, d, d > 128 ? d / 32 : 4);
                if ((rc = launch_linear(ctx, x_r, w.sa_in_w, w.sa_in_b, nullptr, h_r, B, d, d, false, s))) return rc;
                q_r = h_r;
            }
            irs_prof_begin(ctx, IRS_PROF_ATTN, s);
            if (seq_mode) {
            } else if (d / ctx->dims.n_heads == 32 && L <= 256 && d % 4 == 0 && (((uintptr_t)ctx->act_qkv | (uintptr_t)q_r) & 15) == 0)
                hipLaunchKernelGGL(k_attn_row32, dim3(ctx->dims.n_heads, B), dim3(256), 0, s, ctx->act_qkv, seq, ctx->act_ru,
                                   ao_r, L, d, ctx->dims.mask_mode, off, cnt, qrow, q_r, ctx->seq_padq);
            else
                hipLaunchKernelGGL(k_attn_row, dim3(ctx->dims.n_heads, B), dim3(64), 0, s, ctx->act_qkv, seq, ctx->act_ru, pos,
                                   ao_r, L, d, d / ctx->dims.n_heads, ctx->dims.mask_mode, off, cnt, tok, qrow, q_r);
            irs_prof_end(ctx, IRS_PROF_ATTN, s, 4.0 * B * (double)L * d, 4.0 * 3.0 * B * (double)L * d);
            const bool fused_tail = d == 128 && F == 256;
            const bool any_tail = small_any_shape(d, F) && ctx->w_frag16 && !frag && B <= 2048;
            const bool wide_tail = small_wide_shape(d, F) && ctx->w_frag16 && !frag && !q_split;
            const bool idx_res = (fused_tail || any_tail || wide_tail) && !frag; // the layer kernel reads the residual rows x[qrow[b]] itself
            if (frag && !q_split) hipLaunchKernelGGL(k_gather_rows_frag, dim3(B), dim3(64), 0, s, xf, qrow, x_r, d, d > 128 ? d / 32 : 4);
            else if (!frag && !idx_res) hipLaunchKernelGGL(k_gather_rows_idx, dim3(B), dim3(64), 0, s, x, qrow, x_r, d);
            if (fused_tail) { // one launch for the rest of the layer on the B consumed rows
                SmallBlockArgs sb{ao_r, idx_res ? x : x_r, w.sa_out_w, w.sa_out_b, w.n1_w, w.n1_b, cl, w.n2_w, w.n2_b, w.l1_w, w.l1_b, w.l2_w, w.l2_b,
                                  w.n3_w, w.n3_b, xrows, nullptr, nullptr, nullptr, B, nullptr, idx_res ? qrow : nullptr,
                                  ctx->w_frag16 + (size_t)l * SMALL_WF_LAYER, nullptr};
                irs_prof_begin(ctx, IRS_PROF_LINEAR, s);
                if (B > 2048) hipLaunchKernelGGL((k_block_small16<false, 2>), dim3((B + 31) / 32), dim3(64 * SB_NW), 0, s, sb);
                else hipLaunchKernelGGL((k_block_small16<false, 1>), dim3((B + 15) / 16), dim3(64 * SB_NW), 0, s, sb);
                irs_prof_end(ctx, IRS_PROF_LINEAR, s, 2.0 * B * ((double)d * d + 2.0 * d * F), 4.0 * 3.0 * B * (double)d);
            } else if (wide_tail) { // the same for d = 256
                SmallBlockArgs sb{ao_r, x, w.sa_out_w, w.sa_out_b, w.n1_w, w.n1_b, cl, w.n2_w, w.n2_b, w.l1_w, w.l1_b, w.l2_w, w.l2_b,
                                  w.n3_w, w.n3_b, xrows, nullptr, nullptr, nullptr, B, nullptr, qrow,
                                  ctx->w_frag16 + (size_t)l * wide_lf, nullptr};
                irs_prof_begin(ctx, IRS_PROF_LINEAR, s);
                hipLaunchKernelGGL((k_block_small_wide<256, false>), dim3((B + 15) / 16), dim3(512), 0, s, sb);
                irs_prof_end(ctx, IRS_PROF_LINEAR, s, 2.0 * B * ((double)d * d + 2.0 * d * F), 4.0 * 3.0 * B * (double)d);
            } else if (any_tail) {
                SmallBlockArgs sb{ao_r, x, w.sa_out_w, w.sa_out_b, w.n1_w, w.n1_b, cl, w.n2_w, w.n2_b, w.l1_w, w.l1_b, w.l2_w, w.l2_b,
                                  w.n3_w, w.n3_b, xrows, nullptr, nullptr, nullptr, B, nullptr, qrow,
                                  ctx->w_frag16 + (size_t)l * small_any_layer_floats(d, F), nullptr};
                irs_prof_begin(ctx, IRS_PROF_LINEAR, s);
                launch_small_any(false, B, d, F, sb, s);
                irs_prof_end(ctx, IRS_PROF_LINEAR, s, 2.0 * B * ((double)d * d + 2.0 * d * F), 4.0 * 3.0 * B * (double)d);
            } else if (d <= LIN_BN) {
                if ((rc = launch_linear(ctx, ao_r, w.sa_out_w, w.sa_out_b, x_r, y_r, B, d, d, false, s, w.n1_w, w.n1_b, cl, w.n2_w,
                                        w.n2_b)))
                    return rc;
                if ((rc = launch_linear(ctx, y_r, w.l1_w, w.l1_b, nullptr, h_r, B, F, d, true, s))) return rc;
                if ((rc = launch_linear(ctx, h_r, w.l2_w, w.l2_b, y_r, xrows, B, d, F, false, s, w.n3_w, w.n3_b))) return rc;
            } else { // wider than the LN-fused GEMMs: residual GEMM, then LayerNorm, on the B rows
                float *z_r = ctx->act_ao + (size_t)3 * B * d;
                if ((rc = launch_linear(ctx, ao_r, w.sa_out_w, w.sa_out_b, x_r, z_r, B, d, d, false, s))) return rc;
                hipLaunchKernelGGL(k_ln, dim3((B + 3) / 4), dim3(256), 0, s, z_r, w.n1_w, w.n1_b, cl, w.n2_w, w.n2_b, y_r, B, d,
                                   (const int32_t *)nullptr);
                if ((rc = launch_linear(ctx, y_r, w.l1_w, w.l1_b, nullptr, h_r, B, F, d, true, s))) return rc;
                if ((rc = launch_linear(ctx, h_r, w.l2_w, w.l2_b, y_r, z_r, B, d, F, false, s))) return rc;
                hipLaunchKernelGGL(k_ln, dim3((B + 3) / 4), dim3(256), 0, s, z_r, w.n3_w, w.n3_b, (const float *)nullptr,
                                   (const float *)nullptr, (const float *)nullptr, xrows, B, d, (const int32_t *)nullptr);
            }
            IRS_CHECK_HIP(ctx, hipGetLastError());
            return IRS_OK;
        }
        // d = 128, F = 256, head dim 32: attention writes its output fragment-major and ONE kernel does the rest of
        // the layer (out-projection + LN1/LN2, feed-forward + LN3, the next layer's QKV) with y, h, x' in registers
        const bool fuse_block = fuse_block_cfg;
        if (!att_fused &&
            (rc = launch_attn(ctx, ctx->act_qkv, seq, ctx->act_ru, fuse_block ? yf : ctx->act_ao, B, s, off, cnt, tok, fuse_block,
                              kv_planes)))
            return rc;
        if (frag) {
            const bool last = l + 1 == ctx->dims.n_layers;
            const bool tail = !last && (d == 128 || x6d) && F == 256;
            BlockArgs ba{};
            ba.W1 = w.l1_w, ba.b1 = w.l1_b, ba.W2 = w.l2_w, ba.b2 = w.l2_b, ba.g = w.n3_w, ba.b = w.n3_b;
            ba.Xf = last ? nullptr : xf, ba.Y = last ? x : nullptr, ba.M = rows, ba.m_dev = m_dev;
            ba.Win = tail ? ctx->layer[l + 1].sa_in_w : nullptr, ba.bin = tail ? ctx->layer[l + 1].sa_in_b : nullptr;
            ba.QKV = ctx->act_qkv;
            // feeding the rows-only last layer: its queries are needed for B rows only -> write k | v, 8 of 12 tiles
            const bool kv_only = tail && rows_only && l + 2 == ctx->dims.n_layers;
            ba.qkv_n0 = kv_only ? 128 : 0, ba.qkv_nt1 = kv_only ? 2 : 6;
            q_split = kv_only;
            const double ffn_flops = 4.0 * rows * (double)d * F + (tail ? (kv_only ? 4.0 : 6.0) * rows * (double)d * d : 0.0);
            if (fuse_block) {
                ba.Af = yf, ba.Rf = xf, ba.Wo = w.sa_out_w, ba.bo = w.sa_out_b;
                ba.g1 = w.n1_w, ba.b1n = w.n1_b, ba.c = cl, ba.g2 = w.n2_w, ba.b2n = w.n2_b;
                irs_prof_begin(ctx, IRS_PROF_LINEAR, s);
                if (tail) irs_prof_begin(ctx, IRS_PROF_LAYER, s); // (one family is enabled at a time)
                if (tail && ctx->use_x6 && ctx->w_x6) { // the same layer tail on split-bf16 MFMAs
                    const int npl = (ctx->use_x6 == IRS_GEMM_H3 && ctx->h3_ok) ? 2 : 3;
                    BlockX6Args xa{yf, xf, x6_stream(ctx, npl, l), w.sa_out_b, w.n1_w, w.n1_b, cl, w.n2_w, w.n2_b,
                                   w.l1_b, w.l2_b, w.n3_w, w.n3_b, ctx->layer[l + 1].sa_in_b, xf, ctx->act_qkv, rows, m_dev, kv_only ? 1 : 0};
                    xa.kv_planes = kv_planes ? 1 : 0;
                    x6_launch(kv_only ? 1 : 0, false, x6d ? 8 : 4, npl, rows, xa, s);
                } else if (x6d) IRS_FAIL(ctx, IRS_E_STATE, "d = 256 fused layer kernel without a successor layer");
                else if (tail) hipLaunchKernelGGL((k_block<true, true>), dim3((rows + 127) / 128), dim3(256), 0, s, ba);
                else hipLaunchKernelGGL((k_block<true, false>), dim3((rows + 127) / 128), dim3(256), 0, s, ba);
                irs_prof_end(ctx, IRS_PROF_LINEAR, s, ffn_flops + 2.0 * rows * (double)d * d, (8.0 + 4.0 + (tail ? 12.0 : 0.0)) * rows * (double)d);
                if (tail) irs_prof_end(ctx, IRS_PROF_LAYER, s, ffn_flops + 2.0 * rows * (double)d * d, (8.0 + 4.0 + 12.0) * rows * (double)d);
                qkv_done = tail;
                IRS_CHECK_HIP(ctx, hipGetLastError());
                continue;
            }
            // y <- LN2(LN1(x + ao W_o^T + b_o) + c_l): residual from xf, result to yf only
            if ((rc = launch_linear(ctx, ctx->act_ao, w.sa_out_w, w.sa_out_b, nullptr, nullptr, rows, d, d, false, s, w.n1_w,
                                    w.n1_b, cl, w.n2_w, w.n2_b, xf, yf, m_dev)))
                return rc;
            // x <- LN3(y + relu(y W1^T + b1) W2^T + b2) back into xf (the last layer of a full decode writes the
            // row-major x the caller receives instead); d = 128, F = 256 runs as one kernel with h in registers
            if (d == 128 && F == 256) {
                ba.Yf = yf;
                irs_prof_begin(ctx, IRS_PROF_LINEAR, s);
                if (tail) hipLaunchKernelGGL((k_block<false, true>), dim3((rows + 127) / 128), dim3(256), 0, s, ba);
                else hipLaunchKernelGGL((k_block<false, false>), dim3((rows + 127) / 128), dim3(256), 0, s, ba);
                irs_prof_end(ctx, IRS_PROF_LINEAR, s, ffn_flops, (8.0 + (tail ? 12.0 : 0.0)) * rows * (double)d);
                qkv_done = tail;
            } else {
                if ((rc = launch_linear(ctx, nullptr, w.l1_w, w.l1_b, nullptr, ctx->act_h, rows, F, d, true, s, nullptr, nullptr,
                                        nullptr, nullptr, nullptr, nullptr, nullptr, m_dev, yf)))
                    return rc;
                if ((rc = launch_linear(ctx, ctx->act_h, w.l2_w, w.l2_b, nullptr, last ? x : nullptr, rows, d, F, false, s, w.n3_w,
                                        w.n3_b, nullptr, nullptr, nullptr, yf, last ? nullptr : xf, m_dev)))
                    return rc;
            }
        } else if (small_cfg) {
            // latency path: the rest of the layer (and the next layer's QKV) in one launch per 16 (or 32) tokens; x -> y buffer
            const bool last = l + 1 == ctx->dims.n_layers;
            SmallBlockArgs sb{ctx->act_ao, x, w.sa_out_w, w.sa_out_b, w.n1_w, w.n1_b, cl, w.n2_w, w.n2_b, w.l1_w, w.l1_b, w.l2_w, w.l2_b,
                              w.n3_w, w.n3_b, y, last ? nullptr : ctx->layer[l + 1].sa_in_w, last ? nullptr : ctx->layer[l + 1].sa_in_b,
                              qkv_in(l + 1), rows, m_dev, nullptr, ctx->w_frag16 ? ctx->w_frag16 + (size_t)l * SMALL_WF_LAYER : nullptr,
                              (last || !ctx->w_frag16) ? nullptr
                                                       : ctx->w_frag16 + (size_t)ctx->dims.n_layers * SMALL_WF_LAYER + (size_t)(l + 1) * SMALL_WF_WIN};
            if (att_fused) {
                sb.QKVin = qkv_in(l), sb.r_u = ctx->act_ru, sb.seq_last = seq + (L - 1), sb.cnt = cnt, sb.padq = ctx->seq_padq;
                sb.mask_mode = ctx->dims.mask_mode;
            }
            irs_prof_begin(ctx, IRS_PROF_LINEAR, s);
            // 32 tokens per workgroup once the 16-token tiles outnumber the resident workgroups (2 per CU): per path
            // step 518 vs 563 us at 128 users, but 267 vs 222 at 16 users; 64 tokens measured no better than 32
            if (rows > SMALL_MT2_ROWS) {
                if (sb.Win) hipLaunchKernelGGL((k_block_small16<true, 2>), dim3((rows + 31) / 32), dim3(64 * SB_NW), 0, s, sb);
                else hipLaunchKernelGGL((k_block_small16<false, 2>), dim3((rows + 31) / 32), dim3(64 * SB_NW), 0, s, sb);
            } else if (att_fused) {
                if (sb.Win) hipLaunchKernelGGL((k_block_small16<true, 1, true>), dim3((rows + 15) / 16), dim3(64 * SB_NW), 0, s, sb);
                else hipLaunchKernelGGL((k_block_small16<false, 1, true>), dim3((rows + 15) / 16), dim3(64 * SB_NW), 0, s, sb);
            } else {
                if (sb.Win) hipLaunchKernelGGL((k_block_small16<true, 1>), dim3((rows + 15) / 16), dim3(64 * SB_NW), 0, s, sb);
                else hipLaunchKernelGGL((k_block_small16<false, 1>), dim3((rows + 15) / 16), dim3(64 * SB_NW), 0, s, sb);
            }
            irs_prof_end(ctx, IRS_PROF_LINEAR, s, 2.0 * rows * ((double)d * d + 2.0 * d * F + (last ? 0.0 : 3.0 * d * d)),
                         4.0 * (3.0 + (last ? 0.0 : 3.0)) * rows * (double)d);
            qkv_done = !last;
            float *tswap = x; // the new x lives in the other buffer
            x = y;
            y = tswap;
        } else if (wide_cfg) {
            // d = 256, small batches: the rest of the layer (and the next layer's QKV) in one launch per 16 tokens; x -> y buffer
            const bool last = l + 1 == ctx->dims.n_layers;
            SmallBlockArgs sb{ctx->act_ao, x, w.sa_out_w, w.sa_out_b, w.n1_w, w.n1_b, cl, w.n2_w, w.n2_b, w.l1_w, w.l1_b, w.l2_w, w.l2_b,
                              w.n3_w, w.n3_b, y, last ? nullptr : ctx->layer[l + 1].sa_in_w, last ? nullptr : ctx->layer[l + 1].sa_in_b,
                              ctx->act_qkv, rows, m_dev, nullptr, ctx->w_frag16 + (size_t)l * wide_lf,
                              last ? nullptr : ctx->w_frag16 + (size_t)ctx->dims.n_layers * wide_lf + (size_t)(l + 1) * 3 * d * d};
            irs_prof_begin(ctx, IRS_PROF_LINEAR, s);
            if (last) hipLaunchKernelGGL((k_block_small_wide<256, false>), dim3((rows + 15) / 16), dim3(512), 0, s, sb);
            else hipLaunchKernelGGL((k_block_small_wide<256, true>), dim3((rows + 15) / 16), dim3(512), 0, s, sb);
            irs_prof_end(ctx, IRS_PROF_LINEAR, s, 2.0 * rows * ((double)d * d + 2.0 * d * F + (last ? 0.0 : 3.0 * d * d)),
                         4.0 * (3.0 + (last ? 0.0 : 3.0)) * rows * (double)d);
            qkv_done = !last;
            float *tswap = x; // the new x lives in the other buffer
            x = y;
            y = tswap;
        } else if (any_cfg) {
            // launch-bound small shapes: the rest of the layer (and the next layer's QKV) in one launch per 16 tokens
            const bool last = l + 1 == ctx->dims.n_layers;
            SmallBlockArgs sb{ctx->act_ao, x, w.sa_out_w, w.sa_out_b, w.n1_w, w.n1_b, cl, w.n2_w, w.n2_b, w.l1_w, w.l1_b, w.l2_w, w.l2_b,
                              w.n3_w, w.n3_b, y, last ? nullptr : ctx->layer[l + 1].sa_in_w, last ? nullptr : ctx->layer[l + 1].sa_in_b,
                              ctx->act_qkv, rows, m_dev, nullptr, ctx->w_frag16 + (size_t)l * small_any_layer_floats(d, F),
                              last ? nullptr : ctx->w_frag16 + (size_t)(l + 1) * small_any_layer_floats(d, F) + small_any_win_off(d, F)};
            irs_prof_begin(ctx, IRS_PROF_LINEAR, s);
            launch_small_any(!last, rows, d, F, sb, s);
            irs_prof_end(ctx, IRS_PROF_LINEAR, s, 2.0 * rows * ((double)d * d + 2.0 * d * F + (last ? 0.0 : 3.0 * d * d)),
                         4.0 * (3.0 + (last ? 0.0 : 3.0)) * rows * (double)d);
            qkv_done = !last;
            float *tswap = x; // the new x lives in the other buffer
            x = y;
            y = tswap;
        } else if (d <= LIN_BN || (d <= 2 * LIN_BN && rows >= 32768)) {
            // (rows of 129..256 values: one workgroup per 128 tokens covers ALL columns, so below ~256 workgroups the
            //  unfused form -- one workgroup per 128 x 128 output block, then a LayerNorm pass -- fills the chip better:
            //  C5's 32 windows = 6400 rows ran 4 % slower fused)
            // x <- LN2(LN1(x + ao W_o^T + b_o) + c_l), fused into the GEMM epilogue (y, not in place)
            if ((rc = launch_linear(ctx, ctx->act_ao, w.sa_out_w, w.sa_out_b, x, y, rows, d, d, false, s, w.n1_w, w.n1_b,
                                    cl, w.n2_w, w.n2_b, nullptr, nullptr, m_dev)))
                return rc;
            // h = relu(y W1^T + b1); x <- LN3(y + h W2^T + b2)
            if ((rc = launch_linear(ctx, y, w.l1_w, w.l1_b, nullptr, ctx->act_h, rows, F, d, true, s, nullptr, nullptr, nullptr,
                                    nullptr, nullptr, nullptr, nullptr, m_dev)))
                return rc;
            if ((rc = launch_linear(ctx, ctx->act_h, w.l2_w, w.l2_b, y, x, rows, d, F, false, s, w.n3_w, w.n3_b, nullptr,
                                    nullptr, nullptr, nullptr, nullptr, m_dev)))
                return rc;
        } else {
            // y = x + ao W_o^T + b_o ; x = LN2(LN1(y) + c_l)          (packed rows: every kernel clamps to m_dev)
            if ((rc = launch_linear(ctx, ctx->act_ao, w.sa_out_w, w.sa_out_b, x, y, rows, d, d, false, s, nullptr, nullptr, nullptr,
                                    nullptr, nullptr, nullptr, nullptr, m_dev)))
                return rc;
            hipLaunchKernelGGL(k_ln, dim3((rows + 3) / 4), dim3(256), 0, s, y, w.n1_w, w.n1_b, cl, w.n2_w, w.n2_b, x,
                               rows, d, m_dev);
            // h = relu(x W1^T + b1); y = x + h W2^T + b2; x = LN3(y)
            if ((rc = launch_linear(ctx, x, w.l1_w, w.l1_b, nullptr, ctx->act_h, rows, F, d, true, s, nullptr, nullptr, nullptr,
                                    nullptr, nullptr, nullptr, nullptr, m_dev)))
                return rc;
            if ((rc = launch_linear(ctx, ctx->act_h, w.l2_w, w.l2_b, x, y, rows, d, F, false, s, nullptr, nullptr, nullptr, nullptr,
                                    nullptr, nullptr, nullptr, m_dev)))
                return rc;
            hipLaunchKernelGGL(k_ln, dim3((rows + 3) / 4), dim3(256), 0, s, y, w.n3_w, w.n3_b, (const float *)nullptr,
                               (const float *)nullptr, (const float *)nullptr, x, rows, d, m_dev);
        }
        IRS_CHECK_HIP(ctx, hipGetLastError());
    }
    if (x_out) IRS_CHECK_HIP(ctx, hipMemcpyAsync(x_out, x, sizeof(float) * (size_t)rows * d, hipMemcpyDeviceToDevice, s));
    if (pos && xrows) {
        hipLaunchKernelGGL(k_gather_rows, dim3(B), dim3(64), 0, s, x, pos, xrows, B, L, d);
        IRS_CHECK_HIP(ctx, hipGetLastError());
    }
    return IRS_OK;
}
