// Score-all-items contraction and the selections on it, gfx950.
//
// Replaces (reference, /root/reference):
//   project = nn.Linear(d, n_item)          model/influentialRS.py:83, :214 ; model/uRS.py:45, :68
//   softmax + topk(100) of one row          model/influentialRS.py:418-421
//   sort + history filter + nonzero (rank)  model/influentialRS.py:375-389 ; model/evaluator.py:121-131,266-286
//   log-softmax + scalar gathers            model/evaluator.py:195-205, 307-322
//
// Exact score (shared with oracle/oracle_score.c):
//   e[m][j] = fmaf(x[m][d-1], W[j][d-1], ... fmaf(x[m][0], W[j][0], b[j]))   float32, k ascending
//
// top-k pipeline (irs_launch_topk), nothing of size M x N is ever materialised:
//   1. k_prep_x      rows -> bf16 MFMA fragments + per-row error bound eps
//   2. k_sweep PRE   approximate scores of a catalog prefix; only per-wave group maxima are written
//   3. k_select_thr  T0 = k-th largest group maximum  (>= k items score >= T0)
//   4. k_sweep EMIT  full catalog; (score, id) pairs with score >= T0 - 2 eps appended per row
//   5. k_refine      A_k = k-th largest emitted approx score; survivors (>= A_k - 2 eps) are
//                    re-scored with the exact chain and sorted by (score desc, id asc)
//   6. k_exhaustive  rows whose buffers overflowed are redone exactly over the whole shard
// With |approx - exact| <= eps every member of the exact top-k survives 4 and 5,
// so the result equals the exhaustive exact top-k bit for bit ("Why the
// bf16 filter is exact").  The fp32 sweep (IRS_SWEEP_F32) uses v_mfma_f32_32x32x2_f32,
// which is itself the exact chain, with eps = 0.
//
// MFMA orientation: D[item][row] = W_tile[item][k] . x^T[k][row]; a lane owns one
// scored row (column lane&31) and 16 items, so row-wise reductions (max, count,
// log-sum-exp) are register-local.  W is streamed straight from HBM into
// registers (pre-packed in fragment order: 1 KiB contiguous per wave load);
// the row block x is staged once per workgroup in LDS.
#include <type_traits>

#include "irs_internal.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define MODE_PRE 0
#define MODE_EMIT 1
#define MODE_COUNT 2
#define MODE_DENSE 3
#define MODE_LSE 4
#define MODE_CEGRAD 5 // DENSE's orientation; writes scale * (softmax - onehot(label)) instead of the logits

struct SweepArgs {
    // operands
    const uint4 *wp;     // bf16 fragments [n_tiles][KS][64]
    const float *w32;    // fp32 [n_local][d] (fp32 sweep)
    const float *bias;   // [n_tiles*32], -inf beyond n_local
    const uint4 *xb;     // bf16 fragments of the rows [UT][KS][64]
    const float *x32;    // fp32 rows [M][d] (fp32 sweep)
    int M, M_pad, UT, d;
    int64_t n_local;
    // tile range and decomposition
    int tile_begin, tile_end, tiles_per_wave, n_strips, n_ublocks;
    int tile_stride; // PRE samples tiles tile_begin + i * tile_stride
    int no_stagger;   // ring kernel, development switch: all waves take the step barrier at the same k-step
    int tiles_per_wg; // ring kernel, EMIT: tiles per workgroup strip (0: 4 * tiles_per_wave, the PRE group structure)
    // outputs
    float *gm;                   // PRE  [M_pad / 4][n_groups][4]: the four rows a selection workgroup owns are one contiguous block
    int n_groups;                // PRE
    const float *thr;            // EMIT [M_pad]
    unsigned int *cnt;           // EMIT [M_pad][IRS_CAND_BUCKETS]
    unsigned long long *cand;    // EMIT [M_pad][IRS_CAND_BUCKETS][IRS_CAND_SLOTS]
    int cap;
    const float *ref_score;      // COUNT [M]
    const int64_t *ref_id;       // COUNT [M] (local 0-based, may be out of range)
    unsigned long long *count;   // COUNT [M]
    float *dense;                // DENSE [M][ld]
    int64_t ld;
    float *lse_part;             // LSE [slots][M_pad][2]
    const float *ce_lse;         // CEGRAD [M] log-sum-exp of each row over the WHOLE catalog
    const int64_t *ce_label;     // CEGRAD [M] local 0-based label (outside [0, n_local): no one-hot on this shard); < -2^62: row ignored
    float ce_scale;              // CEGRAD dL/dloss / n_valid
};

__device__ __forceinline__ size_t gm_index(const SweepArgs &a, int group, int row) {
    return ((size_t)(row >> 2) * a.n_groups + group) * 4 + (row & 3);
}

// IEEE-754-2019 maximum (llvm.maximum -> v_maximum3_f32 on gfx950): unlike fmaxf / maxnum it needs no canonicalising
// self-maxima in front (hipcc adds 6 of them per 16-way maximum in IEEE mode), and unlike inline asm it keeps the
// compiler's MFMA -> VALU hazard handling (an asm v_max3_f32 on fresh MFMA results read stale registers).  A NaN
// score propagates, compares false against every threshold and is never emitted.
__device__ __forceinline__ float vmax2(float a, float b) { return __builtin_elementwise_maximum(a, b); }
__device__ __forceinline__ float vmax3(float a, float b, float c) { return vmax2(vmax2(a, b), c); }
// maximum of registers 4i .. 4i+3 (the four consecutive items 8i + 4h .. 8i + 4h + 3 of a lane)
__device__ __forceinline__ float quad_max(const f32x16 &a, int i) { return vmax2(vmax3(a[4 * i], a[4 * i + 1], a[4 * i + 2]), a[4 * i + 3]); }
__device__ __forceinline__ float max16(const f32x16 &a) {
    return vmax2(vmax3(vmax3(a[0], a[1], a[2]), vmax3(a[3], a[4], a[5]), vmax3(a[6], a[7], a[8])),
                 vmax3(vmax3(a[9], a[10], a[11]), vmax3(a[12], a[13], a[14]), a[15]));
}

// workgroup -> (strip, ublock): the n_ublocks workgroups that stream the same
// W strip get ids congruent mod 8, i.e. land on one XCD and share its L2.
__device__ __forceinline__ bool sweep_map(const SweepArgs &a, int &strip, int &ublock) {
    int id = blockIdx.x;
    int xcd = id & 7, local = id >> 3;
    ublock = local % a.n_ublocks;
    strip = (local / a.n_ublocks) * 8 + xcd;
    return strip < a.n_strips;
}

// ---- epilogues (lane = scored row `user`, regs = 16 items of tile `t`, half h)
// In-kernel timing (s_memtime) of the first version showed the emission epilogue at ~680 cycles per 32x32 tile
// against ~260 cycles of MFMA: every hit paid a returning LDS atomic inside a divergent branch.  The second version
// (ballot-compacted wave-private queue) still cost ~560 cycles per tile with a hit (a third of all tiles at ~450
// emitted items per row): its queue fill count lived in a VGPR, so each of the 16 per-register segments carried a
// vector compare, an exec save and two branches for the overflow test, and every hit built its order key in place.
// Now
//   * a tile whose 16-register maximum is below the threshold in every lane is skipped after one v_max3 tree
//     (no canonicalising self-maxima) and one ballot (~2 tiles in 3),
//   * otherwise the four quad maxima the tree produced on the way are tested first (4 ballots); only the
//     registers of a quad with a hit are compared one by one (a tile with one hit: 8 ballots, not 16),
//   * the queue fill count is a SCALAR (readfirstlane after every update), room for the whole tile is checked
//     once (16 entries per lane with a hit) and a hit costs two mbcnt, one address shift and three LDS stores of
//     registers that already exist (score bits, item, row) -- the order key is formed at flush time,
//   * a full queue is flushed with all its global atomics in flight together.
#define EMIT_Q 128
struct EmitQ {
    float *score;        // [EMIT_Q] approximate scores (raw float bits)
    unsigned int *item;  // [EMIT_Q] local item ids
    unsigned int *users; // [EMIT_Q] scored rows
    int n;               // entries queued (wave-uniform, kept in an SGPR)
};

// candidate lists are bucketed by item tile (bucket = tile mod 64): 64 counters per row keep the
// same-address atomic chains short, and interleaving tiles spreads any id-locality of the scores.
// A full bucket hands the candidate to ONE second choice, the opposite bucket (the buckets only spread the atomics; a
// row's candidates are one unordered set): a row overflows when two buckets are full, not when 65 candidates share a
// tile class -- then bit 31 of its first counter is set and k_refine redoes the row exhaustively.  (A probe loop over
// all buckets, inline or as a call, cost the d = 128 emission sweep 10-35 %.)  Counters of full buckets keep counting
// the attempts; readers clamp them to IRS_CAND_SLOTS.
#define IRS_CAND_OVERFLOW 0x80000000u
__device__ __forceinline__ void emit_append_global(const SweepArgs &a, unsigned int user, unsigned long long key) {
    const unsigned int bucket = (((unsigned int)key) >> 5) & (IRS_CAND_BUCKETS - 1);
    size_t cell = (size_t)user * IRS_CAND_BUCKETS + bucket;
    unsigned int slot = atomicAdd(&a.cnt[cell], 1u) & ~IRS_CAND_OVERFLOW;
    if (slot >= IRS_CAND_SLOTS) { // full: one second choice (the opposite bucket), then the row counts as overflowed
        cell ^= IRS_CAND_BUCKETS / 2;
        slot = atomicAdd(&a.cnt[cell], 1u) & ~IRS_CAND_OVERFLOW;
        if (slot >= IRS_CAND_SLOTS) {
            atomicOr(&a.cnt[(size_t)user * IRS_CAND_BUCKETS], IRS_CAND_OVERFLOW);
            return;
        }
    }
    a.cand[cell * IRS_CAND_SLOTS + slot] = key;
}

__device__ __forceinline__ void emit_flush(const SweepArgs &a, EmitQ &q, int lane) {
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < q.n; i += 64)
        emit_append_global(a, q.users[i], ((unsigned long long)irs_fkey(q.score[i]) << 32) | q.item[i]);
    __builtin_amdgcn_wave_barrier();
    q.n = 0;
}

__device__ __forceinline__ void emit_one(EmitQ &q, float v, float thr, unsigned int item, unsigned int user) {
    const unsigned long long mask = __ballot(v >= thr);
    if (mask) { // wave-uniform
        if (v >= thr) {
            const unsigned int pos = __builtin_amdgcn_mbcnt_hi((unsigned int)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mask, (unsigned int)q.n));
            q.score[pos] = v;
            q.item[pos] = item;
            q.users[pos] = user;
        }
        q.n = __builtin_amdgcn_readfirstlane(q.n + (int)__popcll(mask));
    }
}

__device__ __forceinline__ void emit_candidates(const SweepArgs &a, const f32x16 &acc, float thr, int user, int t, int h,
                                                EmitQ &q, int lane) {
    q.n = __builtin_amdgcn_readfirstlane(q.n);
    float qm[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) qm[i] = quad_max(acc, i);
    const float m = vmax2(vmax3(qm[0], qm[1], qm[2]), qm[3]);
    const unsigned long long lanes = __ballot(m >= thr);
    if (!lanes) return;
    const int need = 16 * (int)__popcll(lanes); // upper bound of this tile's hits
    const unsigned int item0 = (unsigned int)(t * 32 + 4 * h);
    if (q.n + need > EMIT_Q) {
        emit_flush(a, q, lane);
        if (need > EMIT_Q) { // more than 8 lanes with a hit (dense ties): room is checked register by register
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (q.n + 64 > EMIT_Q) emit_flush(a, q, lane);
                emit_one(q, acc[r], thr, item0 + (r & 3) + 8 * (r >> 2), (unsigned int)user);
            }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (__ballot(qm[i] >= thr)) { // wave-uniform
#pragma unroll
            for (int e = 0; e < 4; ++e) emit_one(q, acc[4 * i + e], thr, item0 + e + 8 * i, (unsigned int)user);
        }
    }
}

__device__ __forceinline__ EmitQ emit_queue(char *base, int wave) {
    // per wave: 128 x (4 B score, 4 B item, 4 B row), 16 B pad
    char *p = base + wave * (EMIT_Q * 12 + 16);
    EmitQ q;
    q.score = reinterpret_cast<float *>(p);
    q.item = reinterpret_cast<unsigned int *>(p + EMIT_Q * 4);
    q.users = reinterpret_cast<unsigned int *>(p + EMIT_Q * 8);
    q.n = 0;
    return q;
}
#define EMIT_Q_BYTES (4 * (EMIT_Q * 12 + 16))

// =============================== bf16 sweep ===============================
template <int KS, int UB, int MODE>
__global__ void __launch_bounds__(256, 2) k_sweep_bf16(SweepArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4 *xs = reinterpret_cast<uint4 *>(smem); // [UB][KS][64]
    int strip, ublock;
    if (!sweep_map(a, strip, ublock)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // uniform for the compiler too: the emission queue's fill count stays scalar
    const int r = lane & 31, h = lane >> 5;
    const int ut0 = ublock * UB;
    const int ubc = min(UB, a.UT - ut0);
    EmitQ eq = emit_queue(smem + (size_t)UB * KS * 1024, wave);
    {
        const uint4 *src = a.xb + (size_t)ut0 * KS * 64;
        for (int i = tid; i < ubc * KS * 64; i += 256) xs[i] = src[i];
    }
    __syncthreads();
    const int gw = strip * 4 + wave;
    const int ts = a.tile_stride;
    int t0 = a.tile_begin + gw * a.tiles_per_wave * ts;
    int t1 = min(t0 + a.tiles_per_wave * ts, a.tile_end);

    float aux[UB]; // PRE: running max; EMIT: threshold
#pragma unroll
    for (int u = 0; u < UB; ++u) {
        if (MODE == MODE_PRE) aux[u] = -INFINITY;
        else aux[u] = (u < ubc) ? fmaxf(a.thr[(ut0 + u) * 32 + r], -3.0e38f) : INFINITY;
    }
    if (t0 < t1) {
        uint4 an[KS];
        float4 bn[4];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) an[ks] = a.wp[((size_t)t0 * KS + ks) * 64 + lane];
#pragma unroll
        for (int q = 0; q < 4; ++q) bn[q] = *reinterpret_cast<const float4 *>(a.bias + (size_t)t0 * 32 + 8 * q + 4 * h);
        for (int t = t0; t < t1; t += ts) {
            uint4 ac[KS];
            float4 bc[4];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) ac[ks] = an[ks];
#pragma unroll
            for (int q = 0; q < 4; ++q) bc[q] = bn[q];
            if (t + ts < t1) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) an[ks] = a.wp[((size_t)(t + ts) * KS + ks) * 64 + lane];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    bn[q] = *reinterpret_cast<const float4 *>(a.bias + (size_t)(t + ts) * 32 + 8 * q + 4 * h);
            }
            // the bias vector is the C operand of each chain's first MFMA (no accumulator initialisation), and
            // the B fragments of the next row tile are requested from LDS before the current chain is issued
            f32x16 biasv;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                biasv[4 * q + 0] = bc[q].x;
                biasv[4 * q + 1] = bc[q].y;
                biasv[4 * q + 2] = bc[q].z;
                biasv[4 * q + 3] = bc[q].w;
            }
            constexpr bool PF = KS <= 8; // register budget: no fragment prefetch at d_pad = 256
            uint4 bcur[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) bcur[ks] = xs[ks * 64 + lane];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                if (u < ubc) {
                    uint4 bnx[KS];
                    if (PF && u + 1 < UB && u + 1 < ubc) {
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) bnx[ks] = xs[((u + 1) * KS + ks) * 64 + lane];
                    }
                    f32x16 acc;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const uint4 bv = PF ? bcur[ks] : xs[(u * KS + ks) * 64 + lane];
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ac[ks]),
                                                                      __builtin_bit_cast(bf16x8, bv), ks == 0 ? biasv : acc, 0, 0, 0);
                    }
                    if (MODE == MODE_PRE) aux[u] = fmaxf(aux[u], max16(acc));
                    else emit_candidates(a, acc, aux[u], (ut0 + u) * 32 + r, t, h, eq, lane);
                    if (PF && u + 1 < UB && u + 1 < ubc) {
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) bcur[ks] = bnx[ks];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    if (MODE == MODE_EMIT) emit_flush(a, eq, lane);
    if (MODE == MODE_PRE) {
#pragma unroll
        for (int u = 0; u < UB; ++u)
            if (u < ubc) a.gm[gm_index(a, gw * 2 + h, (ut0 + u) * 32 + r)] = aux[u];
    }
}

// =============================== bf16 sweep, row-stationary 2 x RT blocks (>= one full row block) ===============================
// k_sweep_bf16 reads one B fragment from LDS per MFMA and runs one dependent MFMA chain per wave; measured
// in-kernel (s_memtime) that is ~415 cycles for 256 cycles of matrix work per 32x32 tile.  Once the sweep is
// compute bound (hundreds of rows) the roles are swapped: a wave keeps the bf16 fragments of ITS RT row tiles
// in registers for the whole kernel, the four waves of a workgroup share the streamed W tiles through LDS
// (staged ST tiles at a time by all 256 threads: registers -> LDS, double buffered, one barrier per stage), and
// two item tiles are multiplied at once: per k-step 2 fragment reads feed 2 x RT MFMAs on independent
// accumulators (half the LDS traffic per MFMA, no dependent-issue stalls).  Tiles, PRE groups and every
// accumulated value are identical to k_sweep_bf16's: the workgroup walks the 4 x tiles_per_wave tiles its four
// waves would have walked there; group id = (strip * 4 + quarter) * 2 + lane half.
template <int KS, int RT, int MODE>
__global__ void __launch_bounds__(256, 2) k_sweep_bf16_rs(SweepArgs a) {
    constexpr int ST = KS >= 16 ? 2 : 4;       // W tiles per stage (even)
    constexpr int STAGE_U4 = ST * KS * 64;     // uint4 per stage (fragments)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4 *ws = reinterpret_cast<uint4 *>(smem);                                  // [2][ST][KS][64]
    float *bs = reinterpret_cast<float *>(smem + (size_t)2 * STAGE_U4 * 16);      // [2][ST][32] bias
    int strip, ublock;
    if (!sweep_map(a, strip, ublock)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // uniform for the compiler too: the emission queue's fill count stays scalar
    const int r = lane & 31, h = lane >> 5;
    const int ut0 = ublock * 4 * RT + wave * RT; // this wave's first row tile
    EmitQ eq = emit_queue(smem + (size_t)2 * STAGE_U4 * 16 + 2 * ST * 32 * 4, wave);
    const int ts = a.tile_stride, tpw = a.tiles_per_wave;
    const int tfirst = a.tile_begin + strip * 4 * tpw * ts;
    int ntile = 4 * tpw; // tiles this workgroup walks: t(i) = tfirst + i * ts, while < tile_end
    {
        const int avail = (a.tile_end - tfirst + ts - 1) / ts;
        if (avail < ntile) ntile = avail;
    }
    if (ntile <= 0) return;
    // this wave's rows: fragments in registers for the whole kernel
    uint4 xr[RT][KS];
    float aux[RT];
#pragma unroll
    for (int u = 0; u < RT; ++u) {
        const bool live = ut0 + u < a.UT;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            xr[u][ks] = live ? a.xb[((size_t)(ut0 + u) * KS + ks) * 64 + lane] : make_uint4(0u, 0u, 0u, 0u);
        if (MODE == MODE_PRE) aux[u] = -INFINITY;
        else aux[u] = live ? fmaxf(a.thr[(ut0 + u) * 32 + r], -3.0e38f) : INFINITY;
    }
    // staging: thread -> uint4 slots tid, tid + 256, ... of the stage image
    constexpr int NLD = (STAGE_U4 + 255) / 256;
    uint4 stg[NLD];
    float stb = 0.f;
    auto load_stage = [&](int i0) { // tiles i0 .. i0 + ST - 1 (clamped to the last tile: duplicates are never consumed)
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int idx = tid + j * 256;
            if (idx < STAGE_U4) {
                const int tl = idx / (KS * 64);
                const int t = tfirst + min(i0 + tl, ntile - 1) * ts;
                const unsigned int *p = reinterpret_cast<const unsigned int *>(a.wp + (size_t)t * KS * 64 + (idx - tl * KS * 64));
                stg[j] = make_uint4(p[0], p[1], p[2], p[3]);
            }
        }
        if (tid < ST * 32) {
            const int t = tfirst + min(i0 + (tid >> 5), ntile - 1) * ts;
            stb = a.bias[(size_t)t * 32 + (tid & 31)];
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int idx = tid + j * 256;
            if (idx < STAGE_U4) ws[buf * STAGE_U4 + idx] = stg[j];
        }
        if (tid < ST * 32) bs[buf * ST * 32 + tid] = stb;
    };
    load_stage(0);
    store_stage(0);
    __syncthreads();
    int cur = 0;
    int gcount = 0, gw = strip * 4; // PRE: tiles seen in the current group, group (= wave id of k_sweep_bf16)
    for (int i0 = 0; i0 < ntile; i0 += ST) {
        const bool more = i0 + ST < ntile;
        if (more) load_stage(i0 + ST);
#pragma unroll
        for (int tl = 0; tl < ST; tl += 2) {
            const int i = i0 + tl;
            if (i < ntile) { // workgroup-uniform; tile i + 1 may be a clamped duplicate (ignored below)
                const uint4 *w0 = ws + cur * STAGE_U4 + tl * KS * 64 + lane;
                const uint4 *w1 = w0 + KS * 64;
                f32x16 acc[2][RT];
#pragma unroll
                for (int ti = 0; ti < 2; ++ti) {
                    const float *bt = bs + (cur * ST + tl + ti) * 32 + 4 * h;
                    const float4 b0 = *reinterpret_cast<const float4 *>(bt), b1 = *reinterpret_cast<const float4 *>(bt + 8);
                    const float4 b2 = *reinterpret_cast<const float4 *>(bt + 16), b3 = *reinterpret_cast<const float4 *>(bt + 24);
                    f32x16 bv;
                    bv[0] = b0.x, bv[1] = b0.y, bv[2] = b0.z, bv[3] = b0.w, bv[4] = b1.x, bv[5] = b1.y, bv[6] = b1.z, bv[7] = b1.w;
                    bv[8] = b2.x, bv[9] = b2.y, bv[10] = b2.z, bv[11] = b2.w, bv[12] = b3.x, bv[13] = b3.y, bv[14] = b3.z, bv[15] = b3.w;
                    const uint4 av = ti == 0 ? w0[0] : w1[0];
#pragma unroll
                    for (int u = 0; u < RT; ++u) // the bias vector is the C operand of each chain's first MFMA
                        acc[ti][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av),
                                                                             __builtin_bit_cast(bf16x8, xr[u][0]), bv, 0, 0, 0);
                }
#pragma unroll
                for (int ks = 1; ks < KS; ++ks) {
                    const uint4 a0 = w0[ks * 64], a1 = w1[ks * 64];
#pragma unroll
                    for (int u = 0; u < RT; ++u) {
                        acc[0][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0),
                                                                            __builtin_bit_cast(bf16x8, xr[u][ks]), acc[0][u], 0, 0, 0);
                        acc[1][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1),
                                                                            __builtin_bit_cast(bf16x8, xr[u][ks]), acc[1][u], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int ti = 0; ti < 2; ++ti) {
                    if (i + ti < ntile) {
                        const int t = tfirst + (i + ti) * ts;
#pragma unroll
                        for (int u = 0; u < RT; ++u) {
                            if (ut0 + u < a.UT) { // wave-uniform
                                if (MODE == MODE_PRE) aux[u] = fmaxf(aux[u], max16(acc[ti][u]));
                                else emit_candidates(a, acc[ti][u], aux[u], (ut0 + u) * 32 + r, t, h, eq, lane);
                            }
                        }
                        if (MODE == MODE_PRE) {
                            ++gcount;
                            if (gcount == tpw || i + ti + 1 == ntile) { // end of a group: publish, restart
#pragma unroll
                                for (int u = 0; u < RT; ++u) {
                                    if (ut0 + u < a.UT) a.gm[gm_index(a, gw * 2 + h, (ut0 + u) * 32 + r)] = aux[u];
                                    aux[u] = -INFINITY;
                                }
                                gcount = 0;
                                ++gw;
                            }
                        }
                    }
                }
            }
        }
        if (more) store_stage(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    if (MODE == MODE_EMIT) emit_flush(a, eq, lane);
}

// =============================== bf16 sweep, LDS-DMA ring (compute-bound form) ===============================
// What bounded k_sweep_bf16_rs (lab: tools/sweep_lab.hip, 1M x 128, 1024 rows): a bare MFMA stream with every
// operand in registers takes 170-190 us on this chip (it holds ~1.5 GHz under dense bf16 MFMA work: ~60 % of the
// 2.4 GHz peak is the ceiling); LDS fragment reads the compiler issues one k-step ahead add ~15 %, the 16-way maxima
// of four accumulator sets, each followed at once by its own compare + branch, another ~15 %, and the emission
// ~20 % more.  This kernel keeps the row-stationary blocking and changes everything around the MFMAs:
//   * W tiles stream HBM -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR staging, no ds_write pass) into a ring
//     of NSLOT steps; ONE raw s_barrier per step, placed in the MIDDLE of a step's MFMAs, publishes step s + 1 and
//     frees the slot of step s - 1 for the DMA of step s + NSLOT - 1 -- no wave ever waits at it for its own operands;
//   * the fragment reads are inline-asm ds_read_b128 with hand-counted lgkmcnt waits, issued two k-steps ahead and,
//     for the first k-steps of the NEXT step, before the epilogue of this one.  (Compiler-visible LDS reads made
//     hipcc put an s_waitcnt vmcnt(0) in front of the first read after every DMA issue: it cannot prove that the
//     slot being filled is not the slot being read, so each step waited for the load it had just started.)
//   * the epilogue is split: the quad / tile maxima and threshold ballots of ALL accumulator sets of a step form one
//     branch-free block (their latencies overlap); only then are the sets with a hit handled;
//   * hits go to the wave-private LDS queue of the other sweeps (ballot-compacted, scalar fill count), written
//     with inline-asm stores.
// Tiles, PRE groups and accumulated values are those of k_sweep_bf16 / k_sweep_bf16_rs (bit-identical output).
typedef __attribute__((address_space(3))) void irs_lds_void;
typedef const __attribute__((address_space(1))) void irs_glb_void;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int OFF>
__device__ __forceinline__ u32x4 lds_read16(unsigned int addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
// s_waitcnt lgkmcnt(N) that the listed registers depend on (keeps their consumers behind the wait)
template <int N>
__device__ __forceinline__ void lds_wait(u32x4 &a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N)); }
template <int N>
__device__ __forceinline__ void lds_wait(u32x4 &a, u32x4 &b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); }
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
template <int OFF>
__device__ __forceinline__ u32x2 lds_read8(unsigned int addr) {
    u32x2 v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int N>
__device__ __forceinline__ void lds_wait8(u32x2 &a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N)); }

// one candidate into the wave's LDS queue; the stores are inline asm for the same reason as the reads (a
// compiler-visible LDS store behind an LDS-DMA issue gets an s_waitcnt vmcnt(0) in front)
__device__ __forceinline__ void ring_emit_one(EmitQ &q, unsigned int q_addr, float v, float thr, unsigned int item, unsigned int row) {
    const unsigned long long mask = __ballot(v >= thr);
    if (mask) { // wave-uniform
        if (v >= thr) {
            const unsigned int pos = __builtin_amdgcn_mbcnt_hi((unsigned int)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mask, (unsigned int)q.n));
            const unsigned int ad = q_addr + 4u * pos;
            asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:%4\n\tds_write_b32 %0, %3 offset:%5"
                         :: "v"(ad), "v"(v), "v"(item), "v"(row), "n"(EMIT_Q * 4), "n"(EMIT_Q * 8) : "memory");
        }
        q.n = __builtin_amdgcn_readfirstlane(q.n + (int)__popcll(mask));
    }
}

template <int KS, int RT, int TPS, int NW, int WPS, int NSLOT, int MODE, int DBG = 0>
__global__ void __launch_bounds__(NW * 64, WPS) k_sweep_ring(SweepArgs a) {
    static_assert(MODE == MODE_PRE || MODE == MODE_EMIT, "top-k modes only");
    constexpr int SLOT_B = TPS * KS * 1024;  // fragment bytes per ring slot
    constexpr int NP = TPS * KS + TPS;       // DMA pieces per step: KiB fragment pieces + one 256-B bias piece per tile
    constexpr int PPW = (NP + NW - 1) / NW;  // pieces per wave (the tail repeats the last piece)
    constexpr int D = NSLOT - 1;             // steps resident beyond the one being multiplied
    static_assert(NSLOT >= 3, "the ring advances in the middle of a step: at least three slots");
    constexpr int KSYNC = KS >= 2 ? KS / 2 : 0; // k-step in front of which the ring advances
    constexpr int PD = KS >= 3 ? ((KS * TPS >= 32 || (DBG & 2)) ? 1 : 2) : KS - 1; // fragment prefetch distance in k-steps (registers at d_pad = 256)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int strip, ublock;
    if (!sweep_map(a, strip, ublock)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int ut0 = ublock * NW * RT + wave * RT; // this wave's first row tile
    const int ts = a.tile_stride, tpw = a.tiles_per_wave;
    int ntile = a.tiles_per_wg ? a.tiles_per_wg : 4 * tpw; // tiles of this workgroup: t(i) = tfirst + i * ts, while < tile_end
    const int tfirst = a.tile_begin + strip * ntile * ts;
    {
        const int avail = (a.tile_end - tfirst + ts - 1) / ts;
        if (avail < ntile) ntile = avail;
    }
    if (ntile <= 0) return;
    const int nstep = (ntile + TPS - 1) / TPS;
    const unsigned int lds0 = (unsigned int)(size_t)(__attribute__((address_space(3))) char *)smem;
    const unsigned int frag_addr = lds0 + lane * 16;                           // + slot * SLOT_B + (ti * KS + ks) * 1024
    const unsigned int bias_addr = lds0 + NSLOT * SLOT_B + h * 16;             // + (slot * TPS + ti) * 256 + 32 q
    // this wave's rows: fragments in registers for the whole kernel (ordinary loads, complete before the first DMA)
    uint4 xr[RT][KS];
    float aux[RT];
#pragma unroll
    for (int u = 0; u < RT; ++u) {
        const bool live = ut0 + u < a.UT;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            xr[u][ks] = live ? a.xb[((size_t)(ut0 + u) * KS + ks) * 64 + lane] : make_uint4(0u, 0u, 0u, 0u);
        if (MODE == MODE_PRE) aux[u] = -INFINITY;
        else aux[u] = live ? fmaxf(a.thr[(ut0 + u) * 32 + r], -3.0e38f) : INFINITY;
    }
    // a waitcnt the compiler's own counter model sees (vmcnt(0), other counters untouched): with an inline-asm wait it
    // keeps the row loads above on its scoreboard and puts a vmcnt(0) -- draining the ring -- in front of their first
    // use in EVERY iteration of the step loop
    __builtin_amdgcn_s_waitcnt(0x0F70);
    auto issue = [&](int s) { // DMA of step s into slot s % NSLOT (tiles beyond the end repeat the last one: never consumed)
        const int slot = s % NSLOT;
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            int p = wave + j * NW;
            if (p > NP - 1) p = NP - 1;
            if (p < TPS * KS) {
                const int tl = p / KS, ks = p - tl * KS;
                const int t = tfirst + min(s * TPS + tl, ntile - 1) * ts;
                __builtin_amdgcn_global_load_lds((irs_glb_void *)(a.wp + ((size_t)t * KS + ks) * 64 + lane),
                                                 (irs_lds_void *)(smem + slot * SLOT_B + p * 1024), 16, 0, 0);
            } else {
                const int tl = p - TPS * KS;
                const int t = tfirst + min(s * TPS + tl, ntile - 1) * ts;
                __builtin_amdgcn_global_load_lds((irs_glb_void *)(a.bias + (size_t)t * 32 + r),
                                                 (irs_lds_void *)(smem + NSLOT * SLOT_B + (slot * TPS + tl) * 256), 4, 0, 0);
            }
        }
    };
    EmitQ eq = emit_queue(smem + NSLOT * SLOT_B + NSLOT * TPS * 256, wave);
    const unsigned int q_addr = lds0 + NSLOT * SLOT_B + NSLOT * TPS * 256 + wave * (EMIT_Q * 12 + 16);
#pragma unroll
    for (int s = 0; s < D; ++s)
        if (s < nstep) issue(s);
    // step 0 visible to every wave
    if (D - 1 < nstep) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * PPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    f32x16 bv[TPS];          // bias vectors of the step's tiles = C operand of each chain's first MFMA
    u32x4 af[PD + 1][TPS];   // fragment ring over k-steps (slot = ks % (PD + 1))
    auto prefetch_head = [&](int s) { // bias and the first PD k-steps of step s (already published)
        const unsigned int fa = frag_addr + (s % NSLOT) * SLOT_B, ba = bias_addr + (s % NSLOT) * TPS * 256;
#pragma unroll
        for (int ti = 0; ti < TPS; ++ti) {
            const u32x4 q0 = ti == 0 ? lds_read16<0>(ba) : lds_read16<256>(ba);
            const u32x4 q1 = ti == 0 ? lds_read16<32>(ba) : lds_read16<256 + 32>(ba);
            const u32x4 q2 = ti == 0 ? lds_read16<64>(ba) : lds_read16<256 + 64>(ba);
            const u32x4 q3 = ti == 0 ? lds_read16<96>(ba) : lds_read16<256 + 96>(ba);
            typedef __attribute__((ext_vector_type(16))) unsigned int u32x16;
            const u32x16 w = __builtin_shufflevector(__builtin_shufflevector(q0, q1, 0, 1, 2, 3, 4, 5, 6, 7),
                                                     __builtin_shufflevector(q2, q3, 0, 1, 2, 3, 4, 5, 6, 7), 0, 1, 2, 3, 4, 5, 6, 7, 8,
                                                     9, 10, 11, 12, 13, 14, 15);
            bv[ti] = __builtin_bit_cast(f32x16, w); // valid once k-step 0's wait has passed
        }
#pragma unroll
        for (int ks = 0; ks < PD; ++ks) {
            af[ks][0] = ks == 0 ? lds_read16<0>(fa) : lds_read16<1024>(fa);
            if (TPS > 1) af[ks][TPS - 1] = ks == 0 ? lds_read16<KS * 1024>(fa) : lds_read16<(KS + 1) * 1024>(fa);
        }
    };
    static_assert(TPS == 1 || TPS == 2, "one or two tiles per step");
    static_assert(PD <= 2, "prefetch_head is written for a distance of at most two k-steps");
    // The head registers are carried around the loop (requested in the prologue / before an epilogue, consumed in the
    // next iteration).  Registers with an LDS return still pending must not meet a compiler-made copy: it would
    // copy stale contents (seen: the prologue's head landed in other registers than the loop's, and the first tile
    // of every workgroup was multiplied with whatever those held).  So every head is WAITED for -- with all its
    // registers tied to the wait -- before control leaves the straight-line code that requested it: after the
    // prologue and at the end of an iteration, behind the epilogue that hid its latency.
    auto head_landed = [&]() {
#pragma unroll
        for (int ti = 0; ti < TPS; ++ti) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bv[ti]));
#pragma unroll
        for (int ks = 0; ks < PD; ++ks)
#pragma unroll
            for (int ti = 0; ti < TPS; ++ti) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[ks][ti]));
    };
    prefetch_head(0);
    head_landed();
    int gcount = 0, gw = strip * 4; // PRE: tiles seen in the current group; group (= wave id of k_sweep_bf16)
    // Eight waves = two per SIMD running the same program between the same barriers would march in lockstep (MFMA
    // phases together, epilogues together).  The second half of the workgroup takes the step's barrier in front
    // of k-step 0 instead of k-step KSYNC: it then runs half a step behind the first half for the whole kernel,
    // so a SIMD's two waves alternate between matrix work and epilogue.
    const bool late = NW == 8 && wave >= NW / 2 && !a.no_stagger;
    for (int s = 0; s < nstep; ++s) {
        const unsigned int fa = frag_addr + (s % NSLOT) * SLOT_B;
        f32x16 acc[TPS][RT];
        // ---- MFMAs of step s; LDS reads run PD k-steps ahead; the ring advances in front of k-step KSYNC
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (KS > 1 && s + 1 < nstep && (late ? ks == 0 : ks == KSYNC)) {
                // this wave's pieces of step s + 1 have landed: steps s + 1 .. s + D - 1 are in flight in a full
                // pipeline (step s + D is issued behind the barrier), so all but the (D - 2) * PPW youngest DMAs
                if (s + D <= nstep) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * PPW) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier(); // step s + 1 is visible to all; every wave has left step s - 1
                if (s + D < nstep) issue(s + D); // into the slot of step s - 1
            }
            if (DBG & 4) {
                asm volatile("s_nop 15");
                asm volatile("s_nop 15");
                asm volatile("s_nop 15");
                asm volatile("s_nop 15");
            }
            if (ks + PD < KS) {
                switch (ks + PD) { // compile-time after unrolling
#define IRS_RD_(K_)                                                                                        \
    case K_:                                                                                               \
        af[K_ % (PD + 1)][0] = lds_read16<(K_)*1024>(fa);                                                  \
        if (TPS > 1) af[K_ % (PD + 1)][TPS - 1] = lds_read16<(KS + (K_)) * 1024>(fa);                      \
        break;
                    IRS_RD_(0) IRS_RD_(1) IRS_RD_(2) IRS_RD_(3) IRS_RD_(4) IRS_RD_(5) IRS_RD_(6) IRS_RD_(7) IRS_RD_(8) IRS_RD_(9)
                    IRS_RD_(10) IRS_RD_(11) IRS_RD_(12) IRS_RD_(13) IRS_RD_(14) IRS_RD_(15)
#undef IRS_RD_
                default: break;
                }
            }
            // k-steps < PD arrived with the head; reads younger than k-step ks's: those of the next min(PD, KS - 1 - ks) k-steps
            if (ks >= PD) {
                u32x4 &f0 = af[ks % (PD + 1)][0];
                const int younger = (DBG & 1) ? 0 : (KS - 1 - ks < PD ? KS - 1 - ks : PD) * TPS;
                if constexpr (TPS == 1) {
                    if (younger >= 2 * TPS) lds_wait<2 * TPS>(f0);
                    else if (younger >= TPS) lds_wait<TPS>(f0);
                    else lds_wait<0>(f0);
                } else {
                    u32x4 &f1 = af[ks % (PD + 1)][TPS - 1];
                    if (younger >= 2 * TPS) lds_wait<2 * TPS>(f0, f1);
                    else if (younger >= TPS) lds_wait<TPS>(f0, f1);
                    else lds_wait<0>(f0, f1);
                }
            }
#pragma unroll
            for (int ti = 0; ti < TPS; ++ti) {
                const u32x4 av = af[ks % (PD + 1)][ti];
                if (ks == 0) {
#pragma unroll
                    for (int u = 0; u < RT; ++u) // the bias vector is the C operand of each chain's first MFMA
                        acc[ti][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av),
                                                                             __builtin_bit_cast(bf16x8, xr[u][0]), bv[ti], 0, 0, 0);
                } else {
#pragma unroll
                    for (int u = 0; u < RT; ++u)
                        acc[ti][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av),
                                                                             __builtin_bit_cast(bf16x8, xr[u][ks]), acc[ti][u], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0); // keep the k-steps (their waits and MFMAs) in program order
        }
        if (KS == 1 && s + 1 < nstep) { // (no interior k-step to hang the ring advance on)
            if (s + D <= nstep) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * PPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (s + D < nstep) issue(s + D);
        }
        // the head of step s + 1 is requested before this step's epilogue runs
        if (DBG & 8) {
            asm volatile("s_nop 15");
            asm volatile("s_nop 15");
            asm volatile("s_nop 15");
            asm volatile("s_nop 15");
        }
        if (s + 1 < nstep) prefetch_head(s + 1);
        // ---- epilogue of step s
        if (MODE == MODE_PRE) {
#pragma unroll
            for (int ti = 0; ti < TPS; ++ti) {
                if (s * TPS + ti < ntile) {
#pragma unroll
                    for (int u = 0; u < RT; ++u)
                        if (ut0 + u < a.UT) aux[u] = vmax2(aux[u], max16(acc[ti][u]));
                    ++gcount;
                    if (gcount == tpw || s * TPS + ti + 1 == ntile) { // end of a group: publish, restart
#pragma unroll
                        for (int u = 0; u < RT; ++u) {
                            if (ut0 + u < a.UT) a.gm[gm_index(a, gw * 2 + h, (ut0 + u) * 32 + r)] = aux[u];
                            aux[u] = -INFINITY;
                        }
                        gcount = 0;
                        ++gw;
                    }
                }
            }
        } else {
            unsigned long long hit[TPS][RT];
#pragma unroll
            for (int ti = 0; ti < TPS; ++ti)
#pragma unroll
                for (int u = 0; u < RT; ++u) {
                    const unsigned long long b = __ballot(max16(acc[ti][u]) >= aux[u]);
                    hit[ti][u] = (s * TPS + ti < ntile && ut0 + u < a.UT) ? b : 0ull;
                }
#pragma unroll
            for (int ti = 0; ti < TPS; ++ti)
#pragma unroll
                for (int u = 0; u < RT; ++u) {
                    if (hit[ti][u]) { // wave-uniform, no VALU result awaited
                        const unsigned int item0 = (unsigned int)((tfirst + (s * TPS + ti) * ts) * 32 + 4 * h);
                        const unsigned int row = (unsigned int)((ut0 + u) * 32 + r);
                        const float thr = aux[u];
                        eq.n = __builtin_amdgcn_readfirstlane(eq.n);
                        // room for the whole tile (16 entries per lane with a hit), else flush; dense ties go register by register
                        const int need = 16 * (int)__popcll(hit[ti][u]);
                        bool dense = false;
                        if (eq.n + need > EMIT_Q) {
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the queue's asm stores
                            emit_flush(a, eq, lane);
                            dense = need > EMIT_Q;
                        }
                        if (dense) {
#pragma unroll
                            for (int rr = 0; rr < 16; ++rr) {
                                if (eq.n + 64 > EMIT_Q) {
                                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                                    emit_flush(a, eq, lane);
                                }
                                ring_emit_one(eq, q_addr, acc[ti][u][rr], thr, item0 + (rr & 3) + 8 * (rr >> 2), row);
                            }
                        } else {
                            // compares in batches of four, each batch ahead of the scalar tests of its masks: a
                            // v_cmp followed at once by a branch on its result waits out the VALU -> SALU latency
                            unsigned long long qmask[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) qmask[i] = __ballot(quad_max(acc[ti][u], i) >= thr);
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                if (qmask[i]) {
                                    unsigned long long rmask[4];
#pragma unroll
                                    for (int e = 0; e < 4; ++e) rmask[e] = __ballot(acc[ti][u][4 * i + e] >= thr);
                                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                    for (int e = 0; e < 4; ++e)
                                        if (rmask[e]) ring_emit_one(eq, q_addr, acc[ti][u][4 * i + e], thr, item0 + e + 8 * i, row);
                                }
                            }
                        }
                    }
                }
        }
        if (s + 1 < nstep) head_landed();
    }
    if (MODE == MODE_PRE) { // the last strip may hold fewer than four groups: the selection reads all of them
        for (; gw < strip * 4 + 4; ++gw)
#pragma unroll
            for (int u = 0; u < RT; ++u)
                if (ut0 + u < a.UT) a.gm[gm_index(a, gw * 2 + h, (ut0 + u) * 32 + r)] = -INFINITY;
    }
    if (MODE == MODE_EMIT) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        emit_flush(a, eq, lane);
    }
}

// =============================== bf16 sweep, LDS-DMA ring on 16x16x32 MFMAs ===============================
// The same ring (slots, DMA pieces, one mid-step barrier) and the same row-stationary blocking as k_sweep_ring, with
// the multiply on v_mfma_f32_16x16x32_bf16 (KS >= 2): a 32-item tile is two 16-item sub-tiles, a wave keeps RT16
// row tiles of 16 rows in registers.  What it changes:
//   * a sub-tile's result is 4 accumulator registers per row tile (not 16): the threshold test of a row tile is two
//     maximum instructions and one compare, and TWO accumulator sets fit where one did -- the test of sub-tile
//     n runs in the shadow of sub-tile n + 1's MFMAs (a slice per k-step, between the MFMAs of that k-step) instead
//     of behind them, and only the scalar "any hit" branch and the (rare) hit handling sit between two MFMA phases;
//   * the chip holds a higher clock under 16x16x32 than under 32x32x16 at equal cycles per FLOP (MI355X_MICROARCH.md,
//     DVFS give-back item 7).
// Operand maps (cdna_hip_programming.md section 3): A lane (i = l & 15, q = l >> 4) holds W[item i][k = 8q .. 8q+7] of
// a 32-wide k-step, B lane holds x[row i][the same k], D[item 4q + reg][row i].  The catalog stays in its 32x32x16
// fragment order ([tile][ks][64 lanes] x 16 B, lane (r, h) = W[item r][16 ks + 8h ..]): lane (i, q) of sub-tile `sub`,
// k-step ks2 reads the 16 bytes of piece 2 ks2 + (q >> 1), lane (q & 1) * 32 + 16 sub + i -- one ds_read_b128, and
// conflict-free (a 16-lane service group covers 16 distinct 16-byte slots of a 256-byte bank row).  The approximate
// scores need not equal the 32x32x16 kernels' bit for bit (another summation order inside the instruction): both are
// within eps of the exact chain, which is all the filter's proof uses (thresholds may come from either).
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define R16_RAW_CAP 64 // raw queue entries per wave (>= 64: the hit lanes of one row tile always fit an empty queue)
static constexpr size_t ring16_lds_bytes(int KS, int RT16, int NW, int NSLOT) {
    return (size_t)NSLOT * KS * 1024 + (size_t)NSLOT * 256 +
           (size_t)NW * ((EMIT_Q * 12 + 16) + R16_RAW_CAP * 20 + RT16 * 64);
}

template <int KS, int RT16, int NW, int NSLOT, int MODE, int DBG = 0>
__global__ void __launch_bounds__(NW * 64, 2) k_sweep_ring16(SweepArgs a) {
    static_assert(MODE == MODE_PRE || MODE == MODE_EMIT, "top-k modes only");
    static_assert(KS >= 2 && KS % 2 == 0 && RT16 % 2 == 0, "32-wide k-steps, whole 32-row tiles");
    constexpr int KS2 = KS / 2;              // k-steps of 32 per sub-tile
    constexpr int RT = RT16 / 2;             // 32-row tiles per wave (a.UT, a.xb and a.thr count those)
    constexpr int SLOT_B = KS * 1024;        // fragment bytes per ring slot (one 32-item tile per step)
    constexpr int NP = KS + 1;               // DMA pieces per step: KiB fragment pieces + one 256-B bias piece
    constexpr int PPW = (NP + NW - 1) / NW;  // pieces per wave (the tail repeats the last piece)
    constexpr int D = NSLOT - 1;
    static_assert(NSLOT >= 3, "the ring advances in the middle of a step: at least three slots");
    // fragment reads run PD k-steps ahead of their MFMAs through a register ring of PD + 1 (which must divide the KS
    // k-steps of a step: the ring position of a step's first k-step is then the same in every step); a k-step is
    // RT16 x 16 MFMA cycles
    constexpr int PD = (RT16 >= 8 || KS < 8) ? 1 : 3;
    constexpr int RING = PD + 1;
    static_assert(KS % RING == 0 && PD <= KS2, "ring position static per step; next-step reads only behind the barrier");
    constexpr int CPK = (RT16 + KS2 - 1) / KS2; // row tiles whose test rides one k-step
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int strip, ublock;
    if (!sweep_map(a, strip, ublock)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4;
    const int ut0 = (ublock * NW + wave) * RT; // this wave's first 32-row tile
    const int ts = a.tile_stride, tpw = a.tiles_per_wave;
    int ntile = a.tiles_per_wg ? a.tiles_per_wg : 4 * tpw;
    const int tfirst = a.tile_begin + strip * ntile * ts;
    {
        const int avail = (a.tile_end - tfirst + ts - 1) / ts;
        if (avail < ntile) ntile = avail;
    }
    if (ntile <= 0) return;
    const int nstep = ntile;
    const unsigned int lds0 = (unsigned int)(size_t)(__attribute__((address_space(3))) char *)smem;
    const unsigned int frag_addr = lds0 + (q >> 1) * 1024 + ((q & 1) * 32 + c) * 16; // + slot * SLOT_B + ks2 * 2048 + sub * 256
    const unsigned int bias_addr = lds0 + NSLOT * SLOT_B + q * 16;                   // + slot * 256 + sub * 64
    // this wave's rows: B fragments in registers for the whole kernel
    uint4 xr[RT16][KS2];
    float aux[RT16]; // EMIT: the row's threshold; PRE: running group maximum
#pragma unroll
    for (int u = 0; u < RT16; ++u) {
        const int ut = ut0 + (u >> 1);
        const bool live = ut < a.UT;
#pragma unroll
        for (int k2 = 0; k2 < KS2; ++k2)
            xr[u][k2] = live ? a.xb[((size_t)ut * KS + 2 * k2 + (q >> 1)) * 64 + (q & 1) * 32 + (u & 1) * 16 + c]
                             : make_uint4(0u, 0u, 0u, 0u);
        if (MODE == MODE_PRE) aux[u] = -INFINITY;
        else aux[u] = live ? fmaxf(a.thr[ut * 32 + (u & 1) * 16 + c], -3.0e38f) : INFINITY;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0), seen by the compiler's counter model (see k_sweep_ring)
    auto issue = [&](int s) __attribute__((always_inline)) { // DMA of step s into slot s % NSLOT
        const int slot = s % NSLOT;
        const int t = tfirst + s * ts;
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            int p = wave + j * NW;
            if (p > NP - 1) p = NP - 1;
            if (p < KS) {
                __builtin_amdgcn_global_load_lds((irs_glb_void *)(a.wp + ((size_t)t * KS + p) * 64 + lane),
                                                 (irs_lds_void *)(smem + slot * SLOT_B + p * 1024), 16, 0, 0);
            } else {
                __builtin_amdgcn_global_load_lds((irs_glb_void *)(a.bias + (size_t)t * 32 + (lane & 31)),
                                                 (irs_lds_void *)(smem + NSLOT * SLOT_B + slot * 256), 4, 0, 0);
            }
        }
    };
    // per wave behind the ring: dense queue (EMIT_Q x 12 + 16), raw queue (RAW_CAP entries: four scores, then RAW_CAP tags),
    // the rows' thresholds [RT16][16]
    constexpr int RAW_CAP = R16_RAW_CAP;
    constexpr int WAVE_LDS = (EMIT_Q * 12 + 16) + RAW_CAP * 20 + RT16 * 64;
    char *wave_lds = smem + NSLOT * SLOT_B + NSLOT * 256 + wave * WAVE_LDS;
    EmitQ eq = emit_queue(wave_lds, 0);
    const float *raw_s = reinterpret_cast<const float *>(wave_lds + (EMIT_Q * 12 + 16));
    const unsigned int *raw_t = reinterpret_cast<const unsigned int *>(wave_lds + (EMIT_Q * 12 + 16) + RAW_CAP * 16);
    float *thr_q = reinterpret_cast<float *>(wave_lds + (EMIT_Q * 12 + 16) + RAW_CAP * 20);
    const unsigned int raw_addr = lds0 + NSLOT * SLOT_B + NSLOT * 256 + wave * WAVE_LDS + (EMIT_Q * 12 + 16);
    int raw_n = 0; // raw entries queued (wave-uniform)
    const unsigned int lane_v = (unsigned int)lane;
    if (MODE == MODE_EMIT && q == 0) {
#pragma unroll
        for (int u = 0; u < RT16; ++u) thr_q[u * 16 + c] = aux[u];
    }
    __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0): the threshold table, ahead of the first LDS-DMA issue
#pragma unroll
    for (int s = 0; s < D; ++s)
        if (s < nstep) issue(s);
    if (D - 1 < nstep) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * PPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    u32x4 af[RING]; // fragment register ring over the k-steps of a step (position = j % RING)
    // bias words of a sub-tile = C operand of each chain's first MFMA; one register set serves both sub-tiles when the
    // next sub-tile's read is requested after this one's first k-step has consumed it (KS2 > PD)
    constexpr int NB = KS2 > PD ? 1 : 2;
    u32x4 bvr[NB];
    // read of step-relative k-step JJ (JJ >= KS: the next step's JJ - KS); a sub-tile's bias read goes in front of
    // its first fragment read.  Always issued (beyond the last step it reads a stale slot that is never multiplied): the
    // hand-counted waits below assume every read of the schedule is in flight.
#define R16_ONE_(JJ_, FA_, BA_)                                                                                 \
    {                                                                                                           \
        constexpr int jj_ = (JJ_) % KS, sub_ = jj_ / KS2, k2_ = jj_ % KS2;                                      \
        if (k2_ == 0) bvr[sub_ % NB] = lds_read16<sub_ * 64>(BA_);                                              \
        af[jj_ % RING] = lds_read16<k2_ * 2048 + sub_ * 256>(FA_);                                              \
    }
    auto all_landed = [&]() __attribute__((always_inline)) { // every carried register tied to a completed wait (see k_sweep_ring: head_landed)
#pragma unroll
        for (int i = 0; i < NB; ++i) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bvr[i]));
#pragma unroll
        for (int i = 0; i < RING; ++i) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[i]));
    };
    {
        const unsigned int fa0 = frag_addr, ba0 = bias_addr; // slot 0
#pragma unroll
        for (int i = 0; i < NB; ++i) bvr[i] = lds_read16<0>(ba0); // (re-read at its place in the schedule; here only so that the registers are defined)
#pragma unroll
        for (int i = 0; i < RING; ++i) af[i] = lds_read16<0>(fa0);
        all_landed();
        if constexpr (PD >= 1) R16_ONE_(0, fa0, ba0)
        if constexpr (PD >= 2) R16_ONE_(1, fa0, ba0)
        if constexpr (PD >= 3) R16_ONE_(2, fa0, ba0)
        all_landed();
    }
    // DBG & 1 (lab only): s_memtime stamps -- cycles at the step barrier, in the hit handling, at the end-of-step wait
    unsigned long long dbg_t[4] = {0ull, 0ull, 0ull, 0ull}, dbg_n = 0ull;
    auto stamp = [&]() __attribute__((always_inline)) -> unsigned long long {
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        return t;
    };
    const unsigned long long dbg_t0 = (DBG & 1) ? stamp() : 0ull;
    f32x4 acc[2][RT16];
    unsigned long long hm[RT16];
#pragma unroll
    for (int u = 0; u < RT16; ++u) {
        hm[u] = 0ull;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[1][u][e] = -INFINITY; // "previous sub-tile" of the first step: tests false, folds as -inf
    }
    int gcount = 0, gw = strip * 4; // PRE: tiles folded into the current group; group index (k_sweep_bf16's wave id)

    // test (EMIT) or fold (PRE) of row tile U of accumulator set Y
#define R16_CHECK_(Y_, U_)                                                                                      \
    {                                                                                                           \
        const float m_ = vmax2(vmax3(acc[Y_][U_][0], acc[Y_][U_][1], acc[Y_][U_][2]), acc[Y_][U_][3]);          \
        if (MODE == MODE_PRE) aux[U_] = vmax2(aux[U_], m_);                                                     \
        else hm[U_] = __ballot(m_ >= aux[U_]);                                                                  \
    }
    // Hits of the pending sub-tile (accumulator set Y = sub-tile Y of step sy).  What in-kernel stamps and two failed
    // forms established: (1) per row tile AND register a compare, a branch and a ballot-compacted append (the 32x32x16
    // kernel's scheme) cost ~1000 cycles per handled sub-tile here -- dozens of taken branches through cold code;
    // (2) a branch-free dump of ALL accumulators of the lanes with a hit (9 LDS stores per sub-tile under EXEC = hit
    // lanes) was no better: an LDS store costs its ~13 cycles of the CU's store path whatever EXEC holds, and eight
    // waves issuing 18 of them per step saturate it; (3) the vector ALU is the scarce port -- an MFMA blocks vector issue
    // for half its 16 cycles, the threshold tests already take 24 of a sub-tile's ~32 free slots per wave.
    // So: one scalar test per row tile (the compare's mask is already in SGPRs); a row tile with a hit (one in nine)
    // takes a SHORT out-of-line block: the hit lanes store the row tile's four accumulators as they stand (one
    // ds_write_b128) and a tag (lane, row tile, sub-tile number) at a ballot-compacted position of a wave-private raw
    // queue: 7 vector instructions, 2 LDS stores.  Which of the four scores pass, their item ids and rows are worked
    // out when the raw queue is drained (flush_raw: LDS only, 16 entries per pass), into the dense (score, item, row)
    // queue the other sweeps use.
    auto flush_raw = [&]() __attribute__((always_inline)) { // raw entries -> dense queue (-> global candidate lists when that fills)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the raw queue's asm stores
        __builtin_amdgcn_wave_barrier();
        const int n = __builtin_amdgcn_readfirstlane(raw_n);
        const int ee = lane & 3;
        for (int i = 0; i < n; i += 16) {
            const int ent = i + (lane >> 2);
            const bool in = ent < n;
            const float sc = raw_s[(in ? ent : 0) * 4 + ee];
            const unsigned int tag = raw_t[in ? ent : 0];
            const unsigned int sl = tag & 63u, uu = (tag >> 6) & 7u, seq = tag >> 9;
            const float thr = thr_q[uu * 16 + (sl & 15u)];
            const bool hit = in && sc >= thr;
            const unsigned long long mask = __ballot(hit);
            if (mask) { // wave-uniform
                if (eq.n + (int)__popcll(mask) > EMIT_Q) emit_flush(a, eq, lane);
                if (hit) {
                    const unsigned int pos = __builtin_amdgcn_mbcnt_hi((unsigned int)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mask, (unsigned int)eq.n));
                    eq.score[pos] = sc;
                    eq.item[pos] = (unsigned int)((tfirst + (int)(seq >> 1) * ts) * 32) + (seq & 1u) * 16u + 4u * (sl >> 4) + (unsigned int)ee;
                    eq.users[pos] = (unsigned int)(ut0 * 32) + 16u * uu + (sl & 15u);
                }
                eq.n = __builtin_amdgcn_readfirstlane(eq.n + (int)__popcll(mask));
            }
        }
        __builtin_amdgcn_wave_barrier();
        raw_n = __builtin_amdgcn_readfirstlane(0);
    };
    auto dump_u = [&](const f32x4 &v, unsigned long long m, unsigned int stag) __attribute__((always_inline)) { // the hit lanes of one row tile
        const unsigned int pos = __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, (unsigned int)raw_n));
        const unsigned int ads = raw_addr + pos * 16u, adt = raw_addr + RAW_CAP * 16 + pos * 4u;
        unsigned int tag; // formed inside the block: as C++ the eight ORs were hoisted into the hot path
        asm volatile("v_or_b32 %0, %5, %6\n\ts_mov_b64 exec, %1\n\tds_write_b128 %2, %3\n\tds_write_b32 %4, %0\n\ts_mov_b64 exec, -1"
                     : "=&v"(tag) : "s"(m), "v"(ads), "v"(v), "v"(adt), "s"(stag), "v"(lane_v) : "memory");
        raw_n = __builtin_amdgcn_readfirstlane(raw_n + (int)__popcll(m)); // (keeps the count in an SGPR)
    };
    auto handle = [&](auto y_tag, int sy) __attribute__((always_inline)) {
        constexpr int Y = decltype(y_tag)::value;
        const unsigned int seq9 = (unsigned int)(2 * sy + Y) << 9;
        unsigned int retry = 0u;
#pragma unroll
        for (int u = 0; u < RT16; ++u) {
            if (__builtin_expect(hm[u] != 0ull, 0)) {
                if (__builtin_expect(raw_n + (int)__popcll(hm[u]) > RAW_CAP, 0)) retry |= 1u << u;
                else dump_u(acc[Y][u], hm[u], seq9 | ((unsigned int)u << 6));
            }
        }
        while (__builtin_expect(retry != 0u, 0)) { // raw queue full: drain it (the one inlined copy per accumulator set), then the rest
            flush_raw();
            unsigned int again = 0u;
#pragma unroll
            for (int u = 0; u < RT16; ++u) {
                if ((retry >> u) & 1u) {
                    if (raw_n + (int)__popcll(hm[u]) > RAW_CAP) again |= 1u << u;
                    else dump_u(acc[Y][u], hm[u], seq9 | ((unsigned int)u << 6));
                }
            }
            retry = again;
        }
    };
    auto pre_group_end = [&](int tiles_done_incl) __attribute__((always_inline)) { // PRE: tile index (0-based within the strip) just folded completely
        ++gcount;
        if (gcount == tpw || tiles_done_incl + 1 == ntile) {
#pragma unroll
            for (int u = 0; u < RT16; ++u) {
                // items 4q .. 4q+3 (+16): item bit 2 = q & 1 is the group half h of the 32x32 kernels; fold q with q ^ 2
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(aux[u]), __float_as_uint(aux[u]), false, false);
                const float g = vmax2(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
                const int ut = ut0 + (u >> 1);
                if (ut < a.UT && q < 2) a.gm[gm_index(a, gw * 2 + q, ut * 32 + (u & 1) * 16 + c)] = g;
                aux[u] = -INFINITY;
            }
            gcount = 0;
            ++gw;
        }
    };

    for (int s = 0; s < nstep; ++s) {
        const unsigned int fa = frag_addr + (s % NSLOT) * SLOT_B, ba = bias_addr + (s % NSLOT) * 256;
        const unsigned int fan = frag_addr + ((s + 1) % NSLOT) * SLOT_B, ban = bias_addr + ((s + 1) % NSLOT) * 256;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            // accumulator set `sub` is multiplied; set sub ^ 1 (the previous sub-tile) is tested between the MFMAs
#pragma unroll
            for (int k2 = 0; k2 < KS2; ++k2) {
                const int j = sub * KS2 + k2; // compile-time after unrolling
                if (sub == 1 && k2 == 0 && s + 1 < nstep) {
                    // ring advance: this wave's pieces of step s + 1 have landed (all but the (D - 2) * PPW youngest DMAs)
                    const unsigned long long tb0 = (DBG & 1) ? stamp() : 0ull;
                    if (s + D <= nstep) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * PPW) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    const unsigned long long tb1 = (DBG & 1) ? stamp() : 0ull;
                    __builtin_amdgcn_s_barrier(); // step s + 1 visible to all; every wave has left step s - 1
                    if (DBG & 1) {
                        const unsigned long long tb2 = stamp();
                        dbg_t[0] += tb1 - tb0;
                        dbg_t[1] += tb2 - tb1;
                    }
                    if (s + D < nstep) issue(s + D);
                }
                // read of k-step j + PD (of the next step behind the barrier above: PD <= KS2)
                switch (j + PD) {
#define R16_RD_(K_)                                                                  \
    case K_:                                                                         \
        if constexpr ((K_) < KS) R16_ONE_(K_, fa, ba) else R16_ONE_(K_, fan, ban)    \
        break;
                    R16_RD_(1) R16_RD_(2) R16_RD_(3) R16_RD_(4) R16_RD_(5) R16_RD_(6) R16_RD_(7) R16_RD_(8) R16_RD_(9) R16_RD_(10)
                    R16_RD_(11) R16_RD_(12) R16_RD_(13) R16_RD_(14) R16_RD_(15) R16_RD_(16) R16_RD_(17) R16_RD_(18)
#undef R16_RD_
                default: break;
                }
                // reads younger than k-step j's: PD fragment reads and the bias reads among them
                {
                    int younger = PD;
#pragma unroll
                    for (int i = 1; i <= PD; ++i) younger += ((j + i) % KS2 == 0) ? 1 : 0;
                    u32x4 &f0 = af[j % RING];
                    if (k2 == 0) {
                        if (younger == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(f0), "+v"(bvr[sub % NB]));
                        else if (younger == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(f0), "+v"(bvr[sub % NB]));
                        else if (younger == 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(f0), "+v"(bvr[sub % NB]));
                        else if (younger == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(f0), "+v"(bvr[sub % NB]));
                        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f0), "+v"(bvr[sub % NB]));
                    } else {
                        if (younger == 1) lds_wait<1>(f0);
                        else if (younger == 2) lds_wait<2>(f0);
                        else if (younger == 3) lds_wait<3>(f0);
                        else if (younger == 4) lds_wait<4>(f0);
                        else lds_wait<0>(f0);
                    }
                }
                const bf16x8 av = __builtin_bit_cast(bf16x8, af[j % RING]);
                const f32x4 cv = __builtin_bit_cast(f32x4, bvr[sub % NB]);
#pragma unroll
                for (int u = 0; u < RT16; ++u) {
                    acc[sub][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf16x8, xr[u][k2]),
                                                                          k2 == 0 ? cv : acc[sub][u], 0, 0, 0);
                    // the previous sub-tile's row tiles k2 * CPK .. : one test per MFMA slot
                    if (u < CPK && k2 * CPK + u < RT16) R16_CHECK_(sub ^ 1, k2 * CPK + u)
                }
                __builtin_amdgcn_sched_barrier(0); // keep the k-steps (their waits, MFMAs and test slices) in program order
            }
            // between two MFMA phases: the pending sub-tile's scalar "any hit" test and, rarely, its hits
            if (MODE == MODE_EMIT) {
                const unsigned long long th0 = (DBG & 1) ? stamp() : 0ull;
                if (sub == 0) {
                    if (s > 0) handle(std::integral_constant<int, 1>{}, s - 1);
                } else
                    handle(std::integral_constant<int, 0>{}, s);
                if (DBG & 1) {
                    const unsigned long long th1 = stamp();
                    dbg_t[2] += th1 - th0;
                    if (th1 - th0 > 100) ++dbg_n;
                }
            } else if (sub == 0 && s > 0)
                pre_group_end(s - 1); // tile s - 1 is folded completely (its second sub-tile rode this phase)
        }
        all_landed(); // the reads of the next step's first k-steps were requested PD k-steps ago
    }
    if ((DBG & 1) && lane == 0) { // (a.gm is unused by EMIT: the lab reads the stamps from there)
        unsigned long long *o = reinterpret_cast<unsigned long long *>(a.gm) + ((size_t)blockIdx.x * NW + wave) * 8;
        o[0] = stamp() - dbg_t0;
        o[1] = dbg_t[0];
        o[2] = dbg_t[1];
        o[3] = dbg_t[2];
        o[4] = dbg_n;
        o[5] = (unsigned long long)nstep;
    }
    // the last sub-tile (set 1 of step nstep - 1) has not been tested yet
#pragma unroll
    for (int u = 0; u < RT16; ++u) R16_CHECK_(1, u)
    if (MODE == MODE_EMIT) {
        handle(std::integral_constant<int, 1>{}, nstep - 1);
        flush_raw();
        emit_flush(a, eq, lane);
    } else {
        pre_group_end(nstep - 1);
        for (; gw < strip * 4 + 4; ++gw) // the last strip may hold fewer than four groups: the selection reads all of them
#pragma unroll
            for (int u = 0; u < RT16; ++u) {
                const int ut = ut0 + (u >> 1);
                if (ut < a.UT && q < 2) a.gm[gm_index(a, gw * 2 + q, ut * 32 + (u & 1) * 16 + c)] = -INFINITY;
            }
    }
#undef R16_ONE_
#undef R16_CHECK_
}

// =============================== fp32 sweep ===============================
// v_mfma_f32_32x32x2_f32: lane (r, h) supplies A[r][k=h] and B[k=h][r]; one
// instruction is fma(a_k1, b_k1, fma(a_k0, b_k0, C)), so issuing k pairs in
// ascending order reproduces the exact chain.  LDS image of the rows:
// xs4[u][q][lane] = float4 {x[row r][8q + 2t + h]}, t = 0..3.
template <int KS, int UB, int MODE, bool VEC>
__global__ void __launch_bounds__(256, 2) k_sweep_f32(SweepArgs a) {
    constexpr int QN = 2 * KS; // float4 groups per row (d_pad / 8)
    // The log-sum-exp is a float32 sum of exp() over the catalog in no particular order: it does not need the chain's
    // summation order, so each lane loads ONE float4 per 8 k (its half's four consecutive k) instead of two.
    constexpr bool PERM = MODE == MODE_LSE && VEC;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *xs4 = reinterpret_cast<float4 *>(smem); // [UB][QN][64]
    int strip, ublock;
    if (!sweep_map(a, strip, ublock)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // uniform for the compiler too: the emission queue's fill count stays scalar
    const int r = lane & 31, h = lane >> 5;
    const int ut0 = ublock * UB;
    const int ubc = min(UB, a.UT - ut0);
    EmitQ eq = emit_queue(smem + (size_t)UB * KS * 2048, wave);
    for (int i = tid; i < ubc * QN * 64; i += 256) {
        int ln = i & 63, q = (i >> 6) % QN, u = (i >> 6) / QN;
        int row = (ut0 + u) * 32 + (ln & 31), hh = ln >> 5;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < a.M) {
            const float *xr = a.x32 + (size_t)row * a.d;
            if (PERM) { // half h holds k = 8q + 4h + c: same pairing on both operands, another summation order
                int k = 8 * q + 4 * hh;
                if (k < a.d) v.x = xr[k];
                if (k + 1 < a.d) v.y = xr[k + 1];
                if (k + 2 < a.d) v.z = xr[k + 2];
                if (k + 3 < a.d) v.w = xr[k + 3];
            } else {
                int k = 8 * q + hh;
                if (k < a.d) v.x = xr[k];
                if (k + 2 < a.d) v.y = xr[k + 2];
                if (k + 4 < a.d) v.z = xr[k + 4];
                if (k + 6 < a.d) v.w = xr[k + 6];
            }
        }
        xs4[i] = v;
    }
    __syncthreads();
    const int gw = strip * 4 + wave;
    const int ts = a.tile_stride;
    int t0 = a.tile_begin + gw * a.tiles_per_wave * ts;
    int t1 = min(t0 + a.tiles_per_wave * ts, a.tile_end);

    float aux[UB];        // PRE: running max; EMIT: thr; COUNT: ref score; LSE: running max
    float aux2[UB];       // LSE: running sum
    unsigned int cnt[UB]; // COUNT
    long long refid[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
        int user = (ut0 + u) * 32 + r;
        aux[u] = -INFINITY;
        aux2[u] = 0.f;
        cnt[u] = 0;
        refid[u] = -1;
        if (MODE == MODE_EMIT) aux[u] = (u < ubc) ? fmaxf(a.thr[user], -3.0e38f) : INFINITY;
        if (MODE == MODE_COUNT) {
            aux[u] = (u < ubc && user < a.M) ? a.ref_score[user] : INFINITY;
            refid[u] = (u < ubc && user < a.M) ? (long long)a.ref_id[user] : -1;
        }
    }
    for (int t = t0; t < t1; t += ts) {
        // A fragments for the whole tile
        float4 af[QN];
        const int64_t item = (int64_t)t * 32 + r;
        if (PERM) {
            const float *wr = a.w32 + (size_t)(item < a.n_local ? item : 0) * a.d + 4 * h;
#pragma unroll
            for (int q = 0; q < QN; ++q) {
                float4 s = *reinterpret_cast<const float4 *>(wr + 8 * q);
                if (item >= a.n_local) s = make_float4(0.f, 0.f, 0.f, 0.f);
                af[q] = s;
            }
        } else if (VEC) {
            const float *wr = a.w32 + (size_t)(item < a.n_local ? item : 0) * a.d;
#pragma unroll
            for (int q = 0; q < QN; ++q) {
                float4 v0 = *reinterpret_cast<const float4 *>(wr + 8 * q);
                float4 v1 = *reinterpret_cast<const float4 *>(wr + 8 * q + 4);
                float4 s = h ? make_float4(v0.y, v0.w, v1.y, v1.w) : make_float4(v0.x, v0.z, v1.x, v1.z);
                if (item >= a.n_local) s = make_float4(0.f, 0.f, 0.f, 0.f);
                af[q] = s;
            }
        } else {
#pragma unroll
            for (int q = 0; q < QN; ++q) {
                float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
                if (item < a.n_local) {
                    const float *wr = a.w32 + (size_t)item * a.d;
                    int k = 8 * q + h;
                    if (k < a.d) s.x = wr[k];
                    if (k + 2 < a.d) s.y = wr[k + 2];
                    if (k + 4 < a.d) s.z = wr[k + 4];
                    if (k + 6 < a.d) s.w = wr[k + 6];
                }
                af[q] = s;
            }
        }
        float4 bc[4];
        float bd = 0.f;
        if (MODE == MODE_DENSE || MODE == MODE_CEGRAD) bd = a.bias[(size_t)t * 32 + r];
        else {
#pragma unroll
            for (int q = 0; q < 4; ++q) bc[q] = *reinterpret_cast<const float4 *>(a.bias + (size_t)t * 32 + 8 * q + 4 * h);
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            if (u < ubc) {
                f32x16 acc;
                if (MODE == MODE_DENSE || MODE == MODE_CEGRAD) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[i] = bd;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc[4 * q + 0] = bc[q].x;
                        acc[4 * q + 1] = bc[q].y;
                        acc[4 * q + 2] = bc[q].z;
                        acc[4 * q + 3] = bc[q].w;
                    }
                }
#pragma unroll
                for (int q = 0; q < QN; ++q) {
                    float4 bv = xs4[(u * QN + q) * 64 + lane];
                    if (MODE == MODE_DENSE || MODE == MODE_CEGRAD) { // D[row][item]: rows on registers, items on lanes
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bv.x, af[q].x, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bv.y, af[q].y, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bv.z, af[q].z, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bv.w, af[q].w, acc, 0, 0, 0);
                    } else {
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].x, bv.x, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].y, bv.y, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].z, bv.z, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].w, bv.w, acc, 0, 0, 0);
                    }
                }
                const int user = (ut0 + u) * 32 + r;
                if (MODE == MODE_PRE) aux[u] = fmaxf(aux[u], max16(acc));
                else if (MODE == MODE_EMIT) emit_candidates(a, acc, aux[u], user, t, h, eq, lane);
                else if (MODE == MODE_COUNT) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        long long it = (long long)t * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                        bool before = acc[i] > aux[u] || (acc[i] == aux[u] && it < refid[u]);
                        cnt[u] += (before && it < a.n_local) ? 1u : 0u;
                    }
                } else if (MODE == MODE_DENSE) {
                    const int64_t col = (int64_t)t * 32 + r;
                    if (col < a.n_local) {
                        int row0 = (ut0 + u) * 32 + 4 * h;
                        asm volatile("" : "+v"(row0)); // keep 128 hoisted row addresses out of registers
                        float *dst = a.dense + (size_t)row0 * a.ld + col;
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int dr = (i & 3) + 8 * (i >> 2);
                            if (row0 + dr < a.M) dst[(size_t)dr * a.ld] = acc[i];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0); // one row tile's accumulators live at a time
                } else if (MODE == MODE_CEGRAD) {
                    // dL/dlogit[row][item] = scale * (exp(logit - lse[row]) - [item == label[row]]); ignored rows: 0
                    const int64_t col = (int64_t)t * 32 + r;
                    if (col < a.n_local) {
                        int row0 = (ut0 + u) * 32 + 4 * h;
                        asm volatile("" : "+v"(row0));
                        float *dst = a.dense + (size_t)row0 * a.ld + col;
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int dr = (i & 3) + 8 * (i >> 2);
                            if (row0 + dr < a.M) {
                                const int64_t lab = a.ce_label[row0 + dr];
                                const float p = __expf(acc[i] - a.ce_lse[row0 + dr]);
                                dst[(size_t)dr * a.ld] = (lab < -(1ll << 62)) ? 0.f : a.ce_scale * (p - (lab == col ? 1.f : 0.f));
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                } else if (MODE == MODE_LSE) {
                    float m = fmaxf(aux[u], max16(acc));
                    if (m > -INFINITY) {
                        float s = aux2[u] * __expf(aux[u] - m);
#pragma unroll
                        for (int i = 0; i < 16; ++i) s += __expf(acc[i] - m);
                        aux[u] = m;
                        aux2[u] = s;
                    }
                }
            }
        }
    }
    if (MODE == MODE_EMIT) emit_flush(a, eq, lane);
#pragma unroll
    for (int u = 0; u < UB; ++u) {
        if (u >= ubc) continue;
        const int user = (ut0 + u) * 32 + r;
        if (MODE == MODE_PRE) a.gm[gm_index(a, gw * 2 + h, user)] = aux[u];
        if (MODE == MODE_COUNT) {
            if (user < a.M && cnt[u]) atomicAdd(&a.count[user], (unsigned long long)cnt[u]);
        }
        if (MODE == MODE_LSE) {
            float *p = a.lse_part + ((size_t)(gw * 2 + h) * a.M_pad + user) * 2;
            p[0] = aux[u];
            p[1] = aux2[u];
        }
    }
}

// Log-sum-exp over the catalog, one wave per SIMD (512 registers): a tile's 32 W rows go row-major from global
// memory straight into MFMA A fragments (lane (r, h): row r, four consecutive k of half h per 8 k -- the pairing of
// k is the same on both operands, the summation order is not the chain's, which a sum of exp() does not need), the
// next tile's fragments are requested before the current tile's MFMAs start.  The x image sits in LDS.
// EMIT: the same pass also emits every item whose float32 score reaches the row's emission threshold (a.thr, from the
// bf16 pre-pass) -- a search that needs both the top-k and the log-sum-exp (beam search) then reads the float32
// catalog once and the bf16 one not at all.  The scores here differ from the chain's by < eps[row] (a different
// float32 summation order is within the accumulation term of eps), so k_refine's validity test holds unchanged.
// XREG (one row tile per workgroup, i.e. at most 32 rows -- a beam search's rows): the rows' fragments live in REGISTERS
// (QN float4 per lane) and two fragment sets of W are in flight instead of three.  The LDS form's inner loop came out as
// "ds_read_b128; s_waitcnt lgkmcnt(0); 4 MFMAs": with one wave per SIMD the LDS round trip (~120 cycles) stood in
// front of every 256 cycles of matrix work (MFMA busy ~50 %), and the register file (506 of 512) had no room for a read
// ahead -- every attempt spilled and put a vmcnt(0) in front of each read.
template <int KS, int UB, bool EMIT, bool XREG = false>
__global__ void __launch_bounds__(256, 1) k_lse_f32(SweepArgs a) {
    static_assert(!XREG || UB == 1, "rows in registers: one row tile per workgroup");
    constexpr int QN = 2 * KS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *xs4 = reinterpret_cast<float4 *>(smem); // [UB][QN][64]
    int strip, ublock;
    if (!sweep_map(a, strip, ublock)) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int ut0 = ublock * UB;
    const int ubc = min(UB, a.UT - ut0);
    EmitQ eq = emit_queue(smem + (size_t)UB * KS * 2048, wave);
    float4 xr4[XREG ? QN : 1];
    if (XREG) {
        const int row = ut0 * 32 + r;
#pragma unroll
        for (int q = 0; q < QN; ++q)
            xr4[q] = row < a.M ? *reinterpret_cast<const float4 *>(a.x32 + (size_t)row * a.d + 8 * q + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        for (int i = tid; i < ubc * QN * 64; i += 256) {
            const int ln = i & 63, q = (i >> 6) % QN, u = (i >> 6) / QN;
            const int row = (ut0 + u) * 32 + (ln & 31);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < a.M) v = *reinterpret_cast<const float4 *>(a.x32 + (size_t)row * a.d + 8 * q + 4 * (ln >> 5));
            xs4[i] = v;
        }
        __syncthreads();
    }
    const int gw = strip * 4 + wave;
    const int ts = a.tile_stride;
    const int t0 = a.tile_begin + gw * a.tiles_per_wave * ts;
    const int t1 = min(t0 + a.tiles_per_wave * ts, a.tile_end);
    float mx[UB], sm[UB], thr[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
        mx[u] = -INFINITY;
        sm[u] = 0.f;
        thr[u] = (EMIT && u < ubc) ? fmaxf(a.thr[(ut0 + u) * 32 + r], -3.0e38f) : INFINITY;
    }
    if (t0 < t1) {
        const int tlast = t0 + ((t1 - 1 - t0) / ts) * ts;
        // loads are unconditional (a request past the strip re-reads its last tile): a load under a branch costs a
        // full vmcnt(0) where the branches meet.  Pad items read row 0; their bias is -inf, so is their score.
        auto load = [&](float4(&af)[QN], float4(&bc)[4], int t) {
            const int tc = min(t, tlast);
            const int64_t item = (int64_t)tc * 32 + r;
            const float *wr = a.w32 + (size_t)(item < a.n_local ? item : 0) * a.d + 4 * h;
#pragma unroll
            for (int q = 0; q < QN; ++q) af[q] = *reinterpret_cast<const float4 *>(wr + 8 * q);
#pragma unroll
            for (int q = 0; q < 4; ++q) bc[q] = *reinterpret_cast<const float4 *>(a.bias + (size_t)tc * 32 + 8 * q + 4 * h);
        };
        auto compute = [&](const float4(&af)[QN], const float4(&bc)[4], int t) {
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                if (u < ubc) {
                    f32x16 acc;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc[4 * q + 0] = bc[q].x;
                        acc[4 * q + 1] = bc[q].y;
                        acc[4 * q + 2] = bc[q].z;
                        acc[4 * q + 3] = bc[q].w;
                    }
#pragma unroll
                    for (int q = 0; q < QN; ++q) {
                        const float4 bv = XREG ? xr4[q] : xs4[(u * QN + q) * 64 + lane];
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].x, bv.x, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].y, bv.y, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].z, bv.z, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].w, bv.w, acc, 0, 0, 0);
                    }
                    if (EMIT) emit_candidates(a, acc, thr[u], (ut0 + u) * 32 + r, t, h, eq, lane);
                    const float m = fmaxf(mx[u], max16(acc));
                    if (m > -INFINITY) {
                        float sacc = sm[u] * __expf(mx[u] - m);
#pragma unroll
                        for (int i = 0; i < 16; ++i) sacc += __expf(acc[i] - m);
                        mx[u] = m;
                        sm[u] = sacc;
                    }
                }
            }
        };
        // three fragment sets: two tiles (64 KB per wave, 256 KB per CU) are in flight behind the one being multiplied
        if constexpr (XREG) { // two sets: one tile (32 KB per wave, 128 KB per CU) in flight behind the one being multiplied
            float4 fa[QN], fb[QN], ba[4], bb[4];
            load(fa, ba, t0);
            for (int t = t0; t < t1; t += 2 * ts) {
                load(fb, bb, t + ts);
                compute(fa, ba, t);
                if (t + ts >= t1) break;
                load(fa, ba, t + 2 * ts);
                compute(fb, bb, t + ts);
            }
        } else {
            float4 fa[QN], fb[QN], fc[QN], ba[4], bb[4], bc2[4];
            load(fa, ba, t0);
            load(fb, bb, t0 + ts);
            for (int t = t0; t < t1; t += 3 * ts) {
                load(fc, bc2, t + 2 * ts);
                compute(fa, ba, t);
                if (t + ts >= t1) break;
                load(fa, ba, t + 3 * ts);
                compute(fb, bb, t + ts);
                if (t + 2 * ts >= t1) break;
                load(fb, bb, t + 4 * ts);
                compute(fc, bc2, t + 2 * ts);
            }
        }
    }
    if (EMIT) emit_flush(a, eq, lane);
#pragma unroll
    for (int u = 0; u < UB; ++u) {
        if (u >= ubc) continue;
        float *p = a.lse_part + ((size_t)(gw * 2 + h) * a.M_pad + (ut0 + u) * 32 + r) * 2;
        p[0] = mx[u];
        p[1] = sm[u];
    }
}

// ------------------------------------------------------------------ log-sum-exp (+ emission), at most 32 rows: ring form
// k_lse_f32<.., XREG> with the float32 catalog staged through LDS by LDS-DMA instead of loaded into fragment registers.
// There a lane's 16-byte loads walk 32 different item rows per instruction (32-byte pieces: every 128-byte line is
// requested by four instructions of the wave, and a tile's 32 KB does not stay in the 32 KB L1 between them) and only one
// tile per wave fits in registers behind the one being multiplied: 3.8 TB/s on 10M x 256 (beam search, C5) with the
// float32 MFMAs at ~47 %.  Here
//   * a STEP is one CK-wide k chunk of a 32-item tile (CK = 32: 4 KB, four DMA instructions of 1 KB = eight items' 128-byte
//     segments each; the tile's 32 biases ride with its first chunk as one 4-byte-per-lane DMA); each wave owns a ring of
//     four slots (three steps in flight behind the one being multiplied, counted vmcnt -- no barrier, the waves share
//     nothing); eight waves per CU, two per SIMD, so that one wave's DMA issue (~100 cycles per instruction) and epilogue
//     run under the other's MFMAs (four waves with 8-KB steps: 2.65 ms per C5 step against 3.10 of the register form;
//     this form 2.61.  What bounds it now is the data in flight: ~96 KB per CU is all the LDS holds, i.e. ~4.6 TB/s at the
//     loaded HBM latency; a timing experiment with every step reading 4 KB of CONTIGUOUS memory changed nothing, so a
//     tile-major copy of the float32 catalog would not help);
//   * the 16-byte chunk p of item i's segment sits at chunk position p ^ key(i), key(i) = (i / items per 256 bytes) mod
//     chunks per segment (the XOR is applied to the global source address, the DMA itself is lane-linear), so the fragment
//     reads -- lane (item r, half h) takes chunks 2 q + h -- are conflict-free in each 16-lane group of ds_read_b128;
//   * fragment reads are inline asm, one chunk ahead inside a tile (counted lgkmcnt); across the tile boundary nothing is
//     in flight, because the epilogue (emission queue, exp) is compiler code that must not find pending asm returns;
//   * the rows' fragments live in registers as in XREG; one wave per SIMD.
// Pad items read row 0 and carry bias -inf.  Summation order per score: that of k_lse_f32 (bias first, then k ascending in
// the MFMA pairing), so the emitted candidates and the (max, sum) partials are those of the register form bit for bit.
template <int B_, int E_, class Fn>
__device__ __forceinline__ void irs_static_for(Fn &&fn) {
    if constexpr (B_ < E_) {
        fn(std::integral_constant<int, B_>{});
        irs_static_for<B_ + 1, E_>(fn);
    }
}
#define LSE_RING_NSLOT 4
#define LSE_RING_NW 8   // waves per workgroup = per CU (two per SIMD: one wave's DMA issue and epilogue under the other's MFMAs)
#define LSE_RING_CK 32  // k values per step
#define LSE_RING_WAVE_B (LSE_RING_NSLOT * (LSE_RING_CK * 128 + 256))
template <int KS, bool EMIT, int NW, int CK>
__global__ void __launch_bounds__(64 * NW, NW / 4) k_lse_ring(SweepArgs a) {
    constexpr int NC = 16 * KS / CK, QN = 2 * KS, NSLOT = LSE_RING_NSLOT, SB = CK * 128, QS = CK / 8;
    constexpr int CH = CK / 4;          // 16-byte chunks of an item's segment
    constexpr int IPB = 16 / CH;        // items per 256 bytes (the bank period)
    static_assert(IPB * 4 * CK == 256 && (1024 / (4 * CK)) / IPB == 4, "key(item) = (4 i + (lane >> 4)) % CH below");
    constexpr int NI = SB / 1024;       // 1-KB DMA instructions per step
    constexpr int NPAT = CH / 4;        // distinct source-chunk patterns over the instructions of a step
    static_assert((NC % NSLOT == 0) || (NSLOT % NC == 0), "slots must stay compile-time constants");
    constexpr int TPI = NC >= NSLOT ? 1 : NSLOT / NC; // tiles per loop iteration
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // a workgroup covers NW / 4 strips of four waves (the sweeps' decomposition unit), XCD-interleaved like sweep_map
    const int strip = ((int)(blockIdx.x >> 3) * 8 + (int)(blockIdx.x & 7)) * (NW / 4) + (wave >> 2);
    if (strip >= a.n_strips) return; // (no barrier in this kernel: a half workgroup may leave)
    const int r = lane & 31, h = lane >> 5;
    EmitQ eq = emit_queue(smem + NW * LSE_RING_WAVE_B, wave);
    float4 xr4[QN];
#pragma unroll
    for (int q = 0; q < QN; ++q)
        xr4[q] = r < a.M ? *reinterpret_cast<const float4 *>(a.x32 + (size_t)r * a.d + 8 * q + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int gw = strip * 4 + (wave & 3);
    const int ts = a.tile_stride;
    const int t0 = a.tile_begin + gw * a.tiles_per_wave * ts;
    const int t1 = min(t0 + a.tiles_per_wave * ts, a.tile_end);
    float mx = -INFINITY, sm = 0.f;
    const float thr = EMIT ? fmaxf(a.thr[r], -3.0e38f) : INFINITY;
    if (t0 < t1) {
        const int ntile = (t1 - 1 - t0) / ts + 1;
        char *ring = smem + wave * LSE_RING_WAVE_B;            // [NSLOT][SB], then [NSLOT][256 B] biases
        const unsigned int ring_a = (unsigned int)(size_t)(__attribute__((address_space(3))) char *)ring;
        // DMA: instruction i of a step brings items i (1024 / (4 CK)) + lane / CH; lane chunk position p = lane % CH holds the
        // item's source chunk p ^ key(item), key(item) = (item / IPB) % CH = (4 i + (lane >> 4)) % CH
        const int jl = lane >> 4, p = lane & (CH - 1), it_l = lane / CH;
        int goff[NPAT]; // float offset inside the item row (without the chunk's CK c), per i % NPAT
#pragma unroll
        for (int i = 0; i < NPAT; ++i) goff[i] = (p ^ ((4 * i + jl) & (CH - 1))) * 4;
        // fragment read addresses: item r, chunks (2 q + h) ^ key(r)
        unsigned int fra[QS];
#pragma unroll
        for (int q = 0; q < QS; ++q) fra[q] = ring_a + (unsigned int)(r * (CK * 4) + (((2 * q + h) ^ ((r / IPB) & (CH - 1))) << 4));
        const unsigned int ba = ring_a + NSLOT * SB + 16 * h; // bias float4 q: + 32 q (+ 256 slot)
        auto issue = [&](int tix, auto cc, auto sc) __attribute__((always_inline)) { // step (tile index tix, chunk C) -> slot S
            constexpr int C = decltype(cc)::value, S = decltype(sc)::value;
            const int t = t0 + min(tix, ntile - 1) * ts; // (a request past the strip re-reads its last tile: counts stay fixed)
            irs_static_for<0, NI>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                const int64_t item = (int64_t)t * 32 + i * (1024 / (4 * CK)) + it_l;
                const float *src = a.w32 + (size_t)(item < a.n_local ? item : 0) * a.d + CK * C + goff[i % NPAT];
                __builtin_amdgcn_global_load_lds((irs_glb_void *)src, (irs_lds_void *)(ring + S * SB + i * 1024), 16, 0, 0);
            });
            if constexpr (C == 0) // the tile's 32 biases ride with its first chunk
                __builtin_amdgcn_global_load_lds((irs_glb_void *)(a.bias + (size_t)t * 32 + (lane & 31)),
                                                 (irs_lds_void *)(ring + NSLOT * SB + S * 256), 4, 0, 0);
        };
// DMA instructions of the two steps behind step (chunk C_): the count a wait for step C_'s slot leaves in flight
#define LSE_RING_YOUNGER(C_) (2 * NI + ((((C_) + 1) % NC) == 0 ? 1 : 0) + ((((C_) + 2) % NC) == 0 ? 1 : 0))
        // prologue: steps 0, 1, 2
        issue(0, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        issue(1 / NC, std::integral_constant<int, 1 % NC>{}, std::integral_constant<int, 1>{});
        issue(2 / NC, std::integral_constant<int, 2 % NC>{}, std::integral_constant<int, 2>{});
        u32x4 fa[2][QS];
        for (int tb = 0; tb < ntile; tb += TPI) {
            irs_static_for<0, TPI>([&](auto tic) __attribute__((always_inline)) {
                constexpr int TI = decltype(tic)::value;
                const int tix = tb + TI;
                if (tix < ntile) { // wave-uniform
                    constexpr int S0 = (TI * NC) % NSLOT; // slot of this tile's chunk 0
                    const int t = t0 + tix * ts;
                    f32x16 acc;
                    irs_static_for<0, NC>([&](auto cc) __attribute__((always_inline)) {
                        constexpr int C = decltype(cc)::value, S = (S0 + C) % NSLOT, SET = C & 1;
                        if constexpr (C == 0) {
                            // this step's slot has landed once only the two younger steps are in flight; slot + 3 is free
                            // (read out a step ago)
                            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LSE_RING_YOUNGER(0)) : "memory");
                            issue(tix + (C + 3) / NC, std::integral_constant<int, (C + 3) % NC>{}, std::integral_constant<int, (S + 3) % NSLOT>{});
                            irs_static_for<0, QS>([&](auto qc) __attribute__((always_inline)) {
                                constexpr int q = decltype(qc)::value;
                                fa[SET][q] = lds_read16<S * SB>(fra[q]);
                            });
                            u32x4 bq[4];
                            bq[0] = lds_read16<S * 256>(ba), bq[1] = lds_read16<S * 256 + 32>(ba), bq[2] = lds_read16<S * 256 + 64>(ba),
                            bq[3] = lds_read16<S * 256 + 96>(ba);
                            if constexpr (QS == 8)
                                asm volatile("s_waitcnt lgkmcnt(0)"
                                             : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]), "+v"(fa[SET][0]), "+v"(fa[SET][1]), "+v"(fa[SET][2]),
                                               "+v"(fa[SET][3]), "+v"(fa[SET][4]), "+v"(fa[SET][5]), "+v"(fa[SET][6]), "+v"(fa[SET][QS - 1]));
                            else
                                asm volatile("s_waitcnt lgkmcnt(0)"
                                             : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]), "+v"(fa[SET][0]), "+v"(fa[SET][1]), "+v"(fa[SET][2]),
                                               "+v"(fa[SET][QS - 1]));
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float4 b4 = __builtin_bit_cast(float4, bq[q]);
                                acc[4 * q + 0] = b4.x, acc[4 * q + 1] = b4.y, acc[4 * q + 2] = b4.z, acc[4 * q + 3] = b4.w;
                            }
                        }
                        if constexpr (C + 1 < NC) // the next chunk's slot has landed (read ahead below)
                            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LSE_RING_YOUNGER(C + 1)) : "memory");
                        irs_static_for<0, QS>([&](auto qc) __attribute__((always_inline)) {
                            constexpr int q = decltype(qc)::value;
                            if constexpr (C + 1 < NC) {
                                fa[SET ^ 1][q] = lds_read16<((S + 1) % NSLOT) * SB>(fra[q]);
                                lds_wait<QS>(fa[SET][q]);
                            } else
                                lds_wait<QS - 1 - q>(fa[SET][q]);
                            const float4 af = __builtin_bit_cast(float4, fa[SET][q]);
                            const float4 bv = xr4[QS * C + q];
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bv.x, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bv.y, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bv.z, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bv.w, acc, 0, 0, 0);
                        });
                        // this chunk's slot is read out (its last read returned before the last MFMA group): refill it
                        if constexpr (C + 1 < NC)
                            issue(tix + (C + 4) / NC, std::integral_constant<int, (C + 4) % NC>{}, std::integral_constant<int, S>{});
                    });
                    if (EMIT) emit_candidates(a, acc, thr, r, t, h, eq, lane);
                    const float m = fmaxf(mx, max16(acc));
                    if (m > -INFINITY) {
                        float sacc = sm * __expf(mx - m);
#pragma unroll
                        for (int i = 0; i < 16; ++i) sacc += __expf(acc[i] - m);
                        mx = m;
                        sm = sacc;
                    }
                }
            });
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the requests past the strip
#undef LSE_RING_YOUNGER
    }
    if (EMIT) emit_flush(a, eq, lane);
    float *pp = a.lse_part + ((size_t)(gw * 2 + h) * a.M_pad + r) * 2;
    pp[0] = mx;
    pp[1] = sm;
}

// =============================== small kernels ===============================
// 8 consecutive k of one row -> one 16-byte bf16 fragment piece (RNE, finite inputs).  Accumulates the squared
// norms the error bound is made of: ss of the values, ssr of the rounded values, ssd of the rounding errors
// (v - bf16(v) is exact in float32: both share an exponent range and the difference has <= 16 significant bits).
__device__ __forceinline__ uint4 pack8_bf16(const float *__restrict__ src, int k0, int d, bool row_ok, float &ss, float &ssr,
                                            float &ssd) {
    unsigned int h[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v = (row_ok && k0 + j < d) ? src[k0 + j] : 0.f;
        unsigned int u = __float_as_uint(v);
        u += 0x7FFFu + ((u >> 16) & 1u);
        h[j] = u >> 16;
        const float vr = __uint_as_float(h[j] << 16), dv = v - vr;
        ss += v * v;
        ssr += vr * vr;
        ssd += dv * dv;
    }
    return make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
}

// rows -> bf16 fragments + eps.  One wave per row of the padded row block;
// lane l < 2*KS packs k = 8l .. 8l+7 (ks = l>>1, half = l&1).
// eps[row] bounds |bf16-MFMA score - exact chain| for EVERY item (DESIGN.md "Why the bf16 filter is exact"):
//   x^ w^ - x w = (x^ - x) w^ + x (w^ - w)   =>   |sum| <= ||dx|| max||w^_j|| + ||x|| max||dw_j||   (Cauchy-Schwarz on the
//   ACTUAL rounding-error vectors: rigorous, and ~0.6x the worst case 2u ||x|| ||w|| on ordinary data)
//   + the float32 accumulation of both sides: (d_pad + 8) 2^-22 (max(||x||, ||x^||) max||w|| + max|b|).
// wn = {max_j max(||w_j||, ||w^_j||), max_j ||w_j - w^_j||, max_j |b_j|}, each rounded up by k_pack_w.
__global__ void __launch_bounds__(256) k_prep_x(const float *__restrict__ x, int M, int M_pad, int d, int KS,
                                                uint4 *__restrict__ xb, float *__restrict__ eps,
                                                const float *__restrict__ wn, float acc_factor,
                                                unsigned int *__restrict__ cand_cnt, int32_t *__restrict__ status,
                                                unsigned int *__restrict__ fb_count, float *__restrict__ thr_carry,
                                                float *__restrict__ traw_carry) {
    int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (fb_count && blockIdx.x == 0 && threadIdx.x == 0) *fb_count = 0u; // (the cooperative fallback's row counter: a memset launch less)
    if (row >= M_pad) return;
    // also resets this row's candidate bucket counters and status word (two memset launches less per call)
    cand_cnt[(size_t)row * IRS_CAND_BUCKETS + lane] = 0u; // IRS_CAND_BUCKETS == 64 == wave size
    if (lane == 0 && row < M) status[row] = 0;
    const int ut = row >> 5, r = row & 31;
    float ss = 0.f, ssr = 0.f, ssd = 0.f;
    if (lane < 2 * KS) {
        uint4 v = pack8_bf16(x + (size_t)row * d, 8 * lane, d, row < M, ss, ssr, ssd);
        xb[((size_t)ut * KS + (lane >> 1)) * 64 + (lane & 1) * 32 + r] = v;
    }
    ss = lanes_sum<63>(ss);
    ssr = lanes_sum<63>(ssr);
    ssd = lanes_sum<63>(ssd);
    if (lane == 0) {
        const float nx = sqrtf(fmaxf(ss, ssr)), ndx = sqrtf(ssd);
        // 1.001: the float32 sums of squares / square roots above (relative error < d 2^-24 + 2^-23)
        const float e = (row < M) ? 1.001f * (ndx * wn[0] + nx * wn[1]) + acc_factor * (1.001f * nx * wn[0] + wn[2]) : 0.f;
        eps[row] = e;
        // carried emission thresholds (irs_launch_topk, carry): EMIT keeps a >= thr as before; k_refine's check "at least k emitted
        // items score >= traw" proves the exact top k were emitted iff traw - 2 eps >= thr for THIS row's eps: traw = thr + 2 eps.
        // (A threshold that is not a number would disable both tests: it becomes -inf = emit everything.)
        if (thr_carry && row < M) {
            float t = thr_carry[row];
            if (!(t == t)) t = -INFINITY, thr_carry[row] = t;
            traw_carry[row] = t > -INFINITY ? t + 2.0f * e : -INFINITY;
        }
    }
}

__global__ void k_zero_eps(float *eps, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) eps[i] = 0.f;
}

// W -> bf16 fragments, padded bias, and the three maxima of the error bound (see k_prep_x).  One wave per item row (padded).
__global__ void __launch_bounds__(256) k_pack_w(const float *__restrict__ W, const float *__restrict__ b, int64_t n_local,
                                                int n_tiles, int d, int KS, uint4 *__restrict__ wp,
                                                float *__restrict__ bias_pad, unsigned int *__restrict__ wn_bits) {
    int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (row >= (int64_t)n_tiles * 32) return;
    const int64_t t = row >> 5;
    const int r = (int)(row & 31);
    float ss = 0.f, ssr = 0.f, ssd = 0.f;
    if (lane < 2 * KS) {
        uint4 v = pack8_bf16(W + (size_t)row * d, 8 * lane, d, row < n_local, ss, ssr, ssd);
        wp[((size_t)t * KS + (lane >> 1)) * 64 + (lane & 1) * 32 + r] = v;
    }
    ss = lanes_sum<63>(ss);
    ssr = lanes_sum<63>(ssr);
    ssd = lanes_sum<63>(ssd);
    if (lane == 0) {
        bias_pad[row] = (row < n_local) ? b[row] : -INFINITY;
        if (row < n_local) { // non-negative floats order like their bit patterns
            atomicMax(wn_bits + 0, __float_as_uint(sqrtf(fmaxf(ss, ssr)) * 1.001f));
            atomicMax(wn_bits + 1, __float_as_uint(sqrtf(ssd) * 1.001f));
            atomicMax(wn_bits + 2, __float_as_uint(fabsf(b[row])));
        }
    }
}

// thr[row] = (k-th largest of the row's G group maxima) - 2 eps ; traw[row] = that order statistic itself
// (-inf when G < k; +inf for padding rows so that they never emit).  G <= 2048 by construction of the pre-pass.
// ONE WAVE PER ROW, four rows per workgroup, no histograms.  The pre-pass leaves the four rows' maxima as one
// contiguous [G][4] block (coalesced 16-byte reads); they are transposed through LDS (each lane ends up with 32 keys
// of its wave's row in registers) and the r-th largest key is found bit by bit: count(keys >= candidate) is 32 compare + add-carry pairs
// per lane against a SCALAR candidate, a four-step DPP reduction inside the 16-lane rows and four v_readlane -- the
// count, the decision and the prefix all live in SGPRs.  The leading bits every key of the row shares are skipped.
// History: the radix form spent ~34 us of a 1024-row call in LDS atomics on the handful of bins near-equal maxima
// fall into; a form with 8 rows per workgroup and 32 lanes per row (64 keys per lane, ds_bpermute reductions) still
// took 28-32 us whatever the row count: 128 workgroups at 1024 rows, each a long chain.
#define SEL2_G IRS_MAX_GROUPS
static_assert(SEL2_G == 2048, "k_select_thr_bits: 32 keys per lane");
template <int CTRL>
__device__ __forceinline__ unsigned int dpp_u32(unsigned int v) {
    return (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
// all lanes of the wave contribute; result is wave-uniform (SGPR)
__device__ __forceinline__ unsigned int wave_sum_u32(unsigned int v) {
    v += dpp_u32<0xB1>(v);  // quad_perm [1,0,3,2]
    v += dpp_u32<0x4E>(v);  // quad_perm [2,3,0,1]
    v += dpp_u32<0x141>(v); // row_half_mirror: the other quad of the 8
    v += dpp_u32<0x140>(v); // row_mirror: the other 8 of the 16
    return (unsigned int)__builtin_amdgcn_readlane((int)v, 0) + (unsigned int)__builtin_amdgcn_readlane((int)v, 16) +
           (unsigned int)__builtin_amdgcn_readlane((int)v, 32) + (unsigned int)__builtin_amdgcn_readlane((int)v, 48);
}
__device__ __forceinline__ unsigned int wave_max_u32(unsigned int v) {
    unsigned int o;
    o = dpp_u32<0xB1>(v), v = o > v ? o : v;
    o = dpp_u32<0x4E>(v), v = o > v ? o : v;
    o = dpp_u32<0x141>(v), v = o > v ? o : v;
    o = dpp_u32<0x140>(v), v = o > v ? o : v;
    const unsigned int a = (unsigned int)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned int)__builtin_amdgcn_readlane((int)v, 16);
    const unsigned int c = (unsigned int)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned int)__builtin_amdgcn_readlane((int)v, 48);
    const unsigned int ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}
__global__ void __launch_bounds__(256) k_select_thr_bits(const float *__restrict__ gm, int G, int M, int M_pad, int k,
                                                         const float *__restrict__ eps, float *__restrict__ thr,
                                                         float *__restrict__ traw) {
    constexpr int STRIDE = SEL2_G + 16; // row stride = 16 mod 64 words: the transposing stores of a wave hit 64 banks once
    constexpr int KPL = SEL2_G / 64;    // keys per lane
    __shared__ unsigned int sel_lds[4 * STRIDE];
    const int tid = threadIdx.x;
    {
        const int rl = tid & 3, tq = tid >> 2; // four consecutive lanes read the four rows of one group (16 bytes)
        const int row = blockIdx.x * 4 + rl;
        const float *blk = gm + (size_t)blockIdx.x * G * 4 + rl; // M_pad is a multiple of 32: the block exists
        float raw[KPL]; // unconditional loads at clamped indices: all in flight together
#pragma unroll
        for (int i = 0; i < KPL; ++i) {
            const int g = tq + 64 * i;
            raw[i] = blk[(size_t)(g < G ? g : G - 1) * 4];
        }
#pragma unroll
        for (int i = 0; i < KPL; ++i) asm volatile("" : "+v"(raw[i])); // keep every load in front of the selects below
#pragma unroll
        for (int i = 0; i < KPL; ++i) {
            const int g = tq + 64 * i;
            const unsigned int kk = irs_fkey(raw[i]);
            sel_lds[rl * STRIDE + g] = (g < G && row < M) ? kk : 0u;
        }
    }
    __syncthreads();
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = blockIdx.x * 4 + wave;
    unsigned int key[KPL];
    unsigned int mx = 0u, mnc = 0u; // mnc = max of the complements = complement of the min over the real groups
#pragma unroll
    for (int i = 0; i < KPL; ++i) {
        key[i] = sel_lds[wave * STRIDE + lane + 64 * i];
        mx = key[i] > mx ? key[i] : mx;
        const unsigned int c = (lane + 64 * i < G) ? ~key[i] : 0u;
        mnc = c > mnc ? c : mnc;
    }
    const unsigned int smx = wave_max_u32(mx), smn = ~wave_max_u32(mnc);
    // keys of the row agree on their bits above `top`; the answer carries those bits too
    const unsigned int diff = smx ^ smn;
    const int top = diff ? 31 - __builtin_clz(diff) : -1;
    unsigned int prefix = top >= 31 ? 0u : (smx & ~((2u << top) - 1u));
    if (top < 0) prefix = smx;
    for (int bit = top; bit >= 0; --bit) {
        const unsigned int cand = prefix | (1u << bit);
        unsigned int c0 = 0, c1 = 0, c2 = 0, c3 = 0; // four partial counts: shorter dependent add chains
#pragma unroll
        for (int i = 0; i < KPL; i += 4) {
            c0 += key[i] >= cand ? 1u : 0u;
            c1 += key[i + 1] >= cand ? 1u : 0u;
            c2 += key[i + 2] >= cand ? 1u : 0u;
            c3 += key[i + 3] >= cand ? 1u : 0u;
        }
        const unsigned int cnt = wave_sum_u32((c0 + c1) + (c2 + c3));
        if (cnt >= (unsigned int)k) prefix = cand;
    }
    if (lane == 0 && row < M_pad) {
        const bool enough = G >= k;
        const float t = enough ? irs_unkey(prefix) : -INFINITY;
        thr[row] = (row < M) ? (enough ? t - 2.0f * eps[row] : -INFINITY) : INFINITY;
        traw[row] = (row < M) ? t : INFINITY;
    }
}

// in-LDS bitonic sort, descending, n a power of two, 256 threads
__device__ __forceinline__ void bitonic_desc(unsigned long long *keys, int n) {
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int i = threadIdx.x; i < n / 2; i += blockDim.x) {
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
}

// Wave-parallel search of the radix bin holding the `need`-th largest key: lane l owns bins
// 4l..4l+3; an inclusive suffix sum over lanes (shuffles) replaces the serial 256-bin scan.
// Executed by wave 0; returns (bin, need - count above bin) through the two shared words.
__device__ __forceinline__ void radix_find_bin(const unsigned int *hist, unsigned int need, int lane, int shift,
                                               unsigned int *s_prefix, unsigned int *s_k) {
    const unsigned int h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
    const unsigned int own = h0 + h1 + h2 + h3;
    unsigned int suf = own; // inclusive suffix sum over lanes >= lane
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        unsigned int o = __shfl_down(suf, off, 64);
        if (lane + off < 64) suf += o;
    }
    const unsigned int above = suf - own;
    if (above < need && suf >= need) {
        unsigned int cum = above;
        int bin;
        if (cum + h3 >= need) bin = 4 * lane + 3;
        else {
            cum += h3;
            if (cum + h2 >= need) bin = 4 * lane + 2;
            else {
                cum += h2;
                if (cum + h1 >= need) bin = 4 * lane + 1;
                else {
                    cum += h1;
                    bin = 4 * lane;
                }
            }
        }
        *s_k = need - cum;
        *s_prefix |= ((unsigned int)bin) << shift;
    }
}

// Exhaustive exact top-k of one row over the whole shard, by one 256-thread workgroup (fallback of the filter
// pipeline -- k_refine calls it for a row whose buffers overflowed -- and GPU-side yard-stick): streams every item
// through the exact chain; keeps the best k by (score desc, id asc).  `buf`: EXH_BUF keys of LDS, `xs`: 256 floats.
#define EXH_BUF 2048
__device__ void exhaustive_row(const float *__restrict__ x, int d, const float *__restrict__ W, const float *__restrict__ bias,
                               int64_t n_local, int64_t item_lo, int k, int row, float *__restrict__ val,
                               int64_t *__restrict__ ids, int32_t *__restrict__ status, unsigned long long *buf, float *xs,
                               bool load_x) {
    __shared__ unsigned int s_n;
    __shared__ unsigned long long s_thr;
    const int tid = threadIdx.x;
    __syncthreads(); // callers may still be reading the LDS this reuses
    if (load_x)
        for (int i = tid; i < d; i += 256) xs[i] = x[(size_t)row * d + i];
    if (tid == 0) {
        s_n = 0;
        s_thr = 0ull;
    }
    __syncthreads();
    for (int64_t base = 0; base < n_local; base += 256) {
        int64_t j = base + tid;
        if (j < n_local) {
            float e = irs_chain(xs, W + (size_t)j * d, bias[j], d);
            unsigned long long key = ((unsigned long long)irs_fkey(e) << 32) | (0xFFFFFFFFu - (unsigned int)j);
            if (key > s_thr) {
                unsigned int slot = atomicAdd(&s_n, 1u);
                buf[slot] = key; // slot < EXH_BUF: compaction below keeps s_n <= EXH_BUF - 256 at loop top
            }
        }
        __syncthreads();
        if (s_n > EXH_BUF - 256) {
            unsigned int n = s_n;
            for (int i = n + tid; i < EXH_BUF; i += 256) buf[i] = 0ull;
            bitonic_desc(buf, EXH_BUF);
            if (tid == 0) {
                s_n = (n < (unsigned int)k) ? n : k;
                if (n >= (unsigned int)k) s_thr = buf[k - 1];
            }
            __syncthreads();
        }
    }
    unsigned int n = s_n;
    for (int i = n + tid; i < EXH_BUF; i += 256) buf[i] = 0ull;
    bitonic_desc(buf, EXH_BUF);
    for (int i = tid; i < k; i += 256) {
        if (i < (int)n) {
            unsigned long long kk = buf[i];
            val[(size_t)row * k + i] = irs_unkey((unsigned int)(kk >> 32));
            ids[(size_t)row * k + i] = item_lo + (int64_t)(0xFFFFFFFFu - (unsigned int)kk);
        } else {
            val[(size_t)row * k + i] = -INFINITY;
            ids[(size_t)row * k + i] = -1;
        }
    }
    if (tid == 0 && (int)n < k) status[row] |= IRS_ROW_FEWER_THAN_K;
}

// ---- cooperative exhaustive fallback (big shards).  A row k_refine cannot finish from its candidate lists (a bucket
// overflowed, or the speculative threshold failed its check: thousands of items within the filter's error of the
// row's k-th score -- near-duplicate catalogs) used to be redone by ONE workgroup walking the whole shard: 25 ms per
// row at 10M x 256.  Now k_refine only records such a row; k_exh_strips spreads the exact scoring of the recorded rows
// over EXH_STRIPS item strips (every workgroup keeps its strip's best k keys per row), k_exh_merge selects each row's
// top k out of the strips' lists with the same streaming selection.  Bound: one pass over the float32 shard for ALL
// flagged rows together (10 GB at 10M x 256: ~2 ms) instead of 25 ms per row.  Both launches return at once when no
// row was recorded (~2 us each).  Rows beyond EXH_FB_MAX recorded rows are redone the old way by k_exh_merge.
#define EXH_FB_MAX 64
#define EXH_STRIPS 256
#define EXH_KEYS_PER_ROW 32768 // strips x k <= this: the strip count shrinks for k > 128

// streaming top-k of keys keyfn(j), j in [j0, j1), 0 = no key: leaves the best min(n, k) keys sorted descending in
// buf[0 ..) and returns their number.  (exhaustive_row's loop, generalised over where the keys come from.)
template <typename KeyFn>
__device__ __forceinline__ unsigned int exh_select(KeyFn &&keyfn, int64_t j0, int64_t j1, int k, unsigned long long *buf) {
    __shared__ unsigned int s_n;
    __shared__ unsigned long long s_thr;
    const int tid = threadIdx.x;
    __syncthreads();
    if (tid == 0) {
        s_n = 0;
        s_thr = 0ull;
    }
    __syncthreads();
    for (int64_t base = j0; base < j1; base += 256) {
        const int64_t j = base + tid;
        if (j < j1) {
            const unsigned long long key = keyfn(j);
            if (key > s_thr) {
                const unsigned int slot = atomicAdd(&s_n, 1u);
                buf[slot] = key;
            }
        }
        __syncthreads();
        if (s_n > EXH_BUF - 256) {
            const unsigned int n = s_n;
            for (int i = n + tid; i < EXH_BUF; i += 256) buf[i] = 0ull;
            bitonic_desc(buf, EXH_BUF);
            if (tid == 0) {
                s_n = (n < (unsigned int)k) ? n : k;
                if (n >= (unsigned int)k) s_thr = buf[k - 1];
            }
            __syncthreads();
        }
    }
    const unsigned int n = s_n;
    for (int i = n + tid; i < EXH_BUF; i += 256) buf[i] = 0ull;
    bitonic_desc(buf, EXH_BUF);
    __syncthreads();
    return n < (unsigned int)k ? n : (unsigned int)k;
}

__global__ void __launch_bounds__(256) k_exh_strips(const float *__restrict__ x, int d, const float *__restrict__ W,
                                                    const float *__restrict__ bias, int64_t n_local, int k, int n_strips,
                                                    const unsigned int *__restrict__ fb_count, const int32_t *__restrict__ fb_list,
                                                    unsigned long long *__restrict__ exh_keys) {
    __shared__ unsigned long long buf[EXH_BUF];
    __shared__ float xs[256];
    unsigned int nfb = *fb_count;
    if (nfb == 0u) return;
    if (nfb > EXH_FB_MAX) nfb = EXH_FB_MAX;
    const int strip = blockIdx.x;
    const int64_t per = (n_local + n_strips - 1) / n_strips;
    const int64_t j0 = (int64_t)strip * per, j1 = (j0 + per < n_local) ? j0 + per : n_local;
    for (unsigned int fi = blockIdx.y; fi < nfb; fi += gridDim.y) {
        const int row = fb_list[fi];
        __syncthreads();
        for (int i = threadIdx.x; i < d; i += 256) xs[i] = x[(size_t)row * d + i];
        __syncthreads();
        const unsigned int n = exh_select(
            [&](int64_t j) -> unsigned long long {
                const float e = irs_chain(xs, W + (size_t)j * d, bias[j], d);
                return ((unsigned long long)irs_fkey(e) << 32) | (0xFFFFFFFFu - (unsigned int)j);
            },
            j0, j1, k, buf);
        unsigned long long *o = exh_keys + ((size_t)fi * n_strips + strip) * k;
        for (int i = threadIdx.x; i < k; i += 256) o[i] = i < (int)n ? buf[i] : 0ull;
    }
}

// One workgroup per row: gather the row's bucketed candidates, validate the emission threshold,
// refine, re-score exactly, sort, write top-k.
__global__ void __launch_bounds__(256) k_refine(const float *__restrict__ x, int d, const float *__restrict__ W,
                                                const float *__restrict__ bias, const unsigned int *__restrict__ cnt,
                                                const unsigned long long *__restrict__ cand,
                                                const float *__restrict__ eps, const float *__restrict__ traw, int k,
                                                int64_t item_lo, int64_t n_local, float *__restrict__ val,
                                                int64_t *__restrict__ ids, int32_t *__restrict__ status,
                                                unsigned int *__restrict__ fb_count, int32_t *__restrict__ fb_list) {
    __shared__ __attribute__((aligned(16))) unsigned long long ckeys[IRS_CAND_CAP + 2];
    // the survivors are compacted into the candidates' own array (every thread holds its candidates in registers
    // across the barrier in between): up to IRS_CAND_CAP survivors fit, no second buffer, and a row with thousands
    // of near-equal scores around its k-th is re-scored here instead of being redone exhaustively
    unsigned long long *rkeys = ckeys;
    __shared__ unsigned int hist[256];
    __shared__ float xs[256];
    __shared__ unsigned int boff[IRS_CAND_BUCKETS + 1];
    __shared__ unsigned int s_prefix, s_k, s_nr, s_above, s_over;
    const int row = blockIdx.x, tid = threadIdx.x;
    // Every slot of the row's candidate buckets is valid memory and the slot addresses do not depend on the bucket
    // counts: the candidate loads, the row and the counts are all requested here, in one memory round trip (the
    // first form waited for the counts and their scan before it asked for the candidates).
    constexpr int PER = IRS_CAND_BUCKETS * IRS_CAND_SLOTS / 256;
    unsigned long long cv[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int idx = tid + 256 * j;
        cv[j] = cand[((size_t)row * IRS_CAND_BUCKETS + idx / IRS_CAND_SLOTS) * IRS_CAND_SLOTS + idx % IRS_CAND_SLOTS];
    }
    for (int i = tid; i < d; i += 256) xs[i] = x[(size_t)row * d + i];
    if (tid < IRS_CAND_BUCKETS) hist[tid] = cnt[(size_t)row * IRS_CAND_BUCKETS + tid];
    __syncthreads();
    if (tid < 64) { // exclusive scan of the (clamped) bucket counts by one wave (IRS_CAND_BUCKETS == 64)
        const unsigned int cb = hist[tid] & ~IRS_CAND_OVERFLOW;
        const unsigned int cl = cb > IRS_CAND_SLOTS ? IRS_CAND_SLOTS : cb;
        unsigned int incl = cl;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned int t = __shfl_up(incl, o, 64);
            if (tid >= o) incl += t;
        }
        boff[tid] = incl - cl;
        if (tid == 63) boff[IRS_CAND_BUCKETS] = incl;
        const unsigned long long ov = __ballot((hist[tid] & IRS_CAND_OVERFLOW) != 0u); // (set on the row's first counter)
        if (tid == 0) {
            s_over = ov ? 1u : 0u;
            s_prefix = 0;
            s_k = k;
            s_nr = 0;
            s_above = 0;
        }
    }
    __syncthreads();
    if (s_over) { // a bucket overflowed -> the row is redone exhaustively: recorded for the cooperative kernels, or here
        if (tid == 0) status[row] |= IRS_ROW_FALLBACK;
        if (fb_count) {
            if (tid == 0) fb_list[atomicAdd(fb_count, 1u)] = row;
            return;
        }
        exhaustive_row(x, d, W, bias, n_local, item_lo, k, row, val, ids, status, ckeys, xs, true);
        return;
    }
    const unsigned int c = boff[IRS_CAND_BUCKETS];
#pragma unroll
    for (int j = 0; j < PER; ++j) { // compaction of the live slots into ckeys, bucket after bucket
        const int idx = tid + 256 * j;
        const int b = idx / IRS_CAND_SLOTS, sl = idx % IRS_CAND_SLOTS;
        if ((unsigned int)sl < boff[b + 1] - boff[b]) ckeys[boff[b] + sl] = cv[j];
    }
    __syncthreads();
    // validation of the (possibly speculative) emission threshold: at least k items must score >= traw,
    // otherwise items between the true k-th score and the threshold may be missing
    const float t1 = traw[row];
    if (t1 > -INFINITY) {
        unsigned int mine = 0;
        for (int i = tid; i < (int)c; i += 256) mine += (irs_unkey((unsigned int)(ckeys[i] >> 32)) >= t1) ? 1u : 0u;
        if (mine) atomicAdd(&s_above, mine);
        __syncthreads();
        if (s_above < (unsigned int)k) {
            if (tid == 0) status[row] |= IRS_ROW_FALLBACK;
            if (fb_count) {
                if (tid == 0) fb_list[atomicAdd(fb_count, 1u)] = row;
                return;
            }
            exhaustive_row(x, d, W, bias, n_local, item_lo, k, row, val, ids, status, ckeys, xs, false);
            return;
        }
    }
    float thr2 = -INFINITY;
    if (c >= (unsigned int)k) {
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            hist[tid] = 0;
            __syncthreads();
            const unsigned int prefix = s_prefix;
            for (int i0 = 0; i0 < (int)c; i0 += 256) { // (trip count uniform over the workgroup)
                const int i = i0 + tid;
                const unsigned int key = i < (int)c ? (unsigned int)(ckeys[i] >> 32) : 0u;
                const bool match = i < (int)c && ((pass == 0) || ((key >> (shift + 8)) == (prefix >> (shift + 8))));
                const unsigned int bin = (key >> shift) & 255;
                // near-equal scores share their leading digits: when every matching lane of the wave holds the same
                // bin, one lane adds the count (hundreds of atomics on one LDS address serialise otherwise)
                const unsigned long long m = __ballot(match);
                if (m) {
                    const unsigned int first = __builtin_amdgcn_readlane(bin, __builtin_ctzll(m));
                    if (__ballot(match && bin == first) == m) {
                        if ((tid & 63) == __builtin_ctzll(m)) atomicAdd(&hist[first], (unsigned int)__popcll(m));
                    } else if (match)
                        atomicAdd(&hist[bin], 1u);
                }
            }
            __syncthreads();
            if (tid < 64) radix_find_bin(hist, s_k, tid, shift, &s_prefix, &s_k);
            __syncthreads();
        }
        thr2 = irs_unkey(s_prefix) - 2.0f * eps[row];
    }
    // survivors
    {
        constexpr int PERC = IRS_CAND_CAP / 256;
        unsigned int keep[PERC];
#pragma unroll
        for (int j = 0; j < PERC; ++j) {
            const int i = tid + 256 * j;
            keep[j] = 0xFFFFFFFFu;
            if (i < (int)c) {
                const unsigned long long key = ckeys[i];
                if (irs_unkey((unsigned int)(key >> 32)) >= thr2) keep[j] = (unsigned int)key; // low word = the item's local index (< 2^32 - 1)
            }
        }
        __syncthreads(); // every candidate has been read: the array is free for the compacted survivors
#pragma unroll
        for (int j = 0; j < PERC; ++j)
            if (keep[j] != 0xFFFFFFFFu) rkeys[atomicAdd(&s_nr, 1u)] = (unsigned long long)keep[j];
    }
    __syncthreads();
    const unsigned int nr = s_nr;
    // exact re-score
    for (int i = tid; i < (int)nr; i += 256) {
        unsigned int j = (unsigned int)rkeys[i];
        float e = irs_chain(xs, W + (size_t)j * d, bias[j], d);
        rkeys[i] = ((unsigned long long)irs_fkey(e) << 32) | (0xFFFFFFFFu - j);
    }
    if ((nr & 1) && tid == 0) rkeys[nr] = 0ull; // (the array has IRS_CAND_CAP + 2 slots)
    __syncthreads();
    // final order by rank counting as well (keys distinct): winners go straight to their slots
    {
        const ulonglong2 *r2 = reinterpret_cast<const ulonglong2 *>(rkeys);
        for (int i = tid; i < (int)nr; i += 256) {
            const unsigned long long me = rkeys[i];
            int above = 0;
            for (int j = 0; j < (int)(nr + 1) / 2; ++j) {
                const ulonglong2 v = r2[j];
                above += (v.x > me) + (v.y > me);
            }
            if (above < k) {
                val[(size_t)row * k + above] = irs_unkey((unsigned int)(me >> 32));
                ids[(size_t)row * k + above] = item_lo + (int64_t)(0xFFFFFFFFu - (unsigned int)me);
            }
        }
        for (int i = (int)nr + tid; i < k; i += 256) {
            val[(size_t)row * k + i] = -INFINITY;
            ids[(size_t)row * k + i] = -1;
        }
    }
    if (tid == 0 && (int)nr < k) status[row] |= IRS_ROW_FEWER_THAN_K;
}

// Small shard x few rows (the single-user latency path on an ml-1m-sized catalog): the five-kernel filter pipeline
// costs ~50 us of launches for ~2 us of work.  Here a workgroup scores 64 items of a row with the exact chain into a
// global key array; the LAST workgroup of the row to finish (arrival counter) selects the k-th largest key and sorts
// the survivors: one launch, same total order, same bits.  Every launch starts with cold caches and a lone CU fetches
// ~30-60 GB/s, so the work is spread thin (64 items = 32 KB of W per CU, staged into LDS with fully coalesced loads by
// all four waves; wave 0 then runs one item per lane) and the selection avoids LDS atomics on a handful of hot radix
// bins: the keys of the row sit in registers (16 per thread) and the threshold is found bit by bit with ballots.
#define DIRECT_MAX_ITEMS 4096 // keys of a row live in the candidate array ([M_pad][IRS_CAND_CAP] u64)
#define DIRECT_TILE 64
#define DIRECT_MAX_D 256
#ifdef IRS_DIRECT_TIMING
__device__ unsigned long long g_direct_t[16];
#define DSTAMP(i) do { if (threadIdx.x == 0) tstamp[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DSTAMP(i)
#endif
// MODE 0: as described (<= 32 rows: one launch).  More rows (<= 1024) take two launches of the same code: MODE 1 scores
// DIRECT_ROWS rows per workgroup against its W tile (the tile is staged once per 8 rows; wave w runs rows w and w + 4),
// MODE 2 is the selection alone, one workgroup per row -- the last-arriver scheme would hand all 8 rows of a group to
// the same workgroup, one after the other.
#define DIRECT_ROWS 8
template <int MODE>
__global__ void __launch_bounds__(256) k_topk_direct(const float *__restrict__ x, int d, const float *__restrict__ W,
                                                     const float *__restrict__ bias, int n_local, int64_t item_lo, int k,
                                                     unsigned long long *__restrict__ gkeys, unsigned int *__restrict__ arrive,
                                                     float *__restrict__ val, int64_t *__restrict__ ids,
                                                     int32_t *__restrict__ status, irs_path_args pa, int M) {
    __shared__ __attribute__((aligned(16))) float tile[DIRECT_TILE * (DIRECT_MAX_D + 4)]; // W tile, later the key array
    __shared__ __attribute__((aligned(16))) unsigned long long rkeys[IRS_REFINE_CAP + 2];
    __shared__ __attribute__((aligned(16))) float xs[(MODE == 1 ? DIRECT_ROWS : 1) * DIRECT_MAX_D];
    __shared__ __attribute__((aligned(16))) unsigned long long tmax[256];
    __shared__ unsigned int wsum[4];
    __shared__ unsigned int s_last, s_thr;
    const int row = MODE == 1 ? blockIdx.y * DIRECT_ROWS : blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef IRS_DIRECT_TIMING
    unsigned long long tstamp[10];
#endif
    DSTAMP(0);
    unsigned int *gk = reinterpret_cast<unsigned int *>(gkeys + (size_t)row * IRS_CAND_CAP);
    if constexpr (MODE != 2) {
    const bool vec = (d & 3) == 0;
    // row stride: 4 mod 8 floats (conflict-free ds_read_b128 of one row per lane), or odd when d is not a multiple
    // of 4 (the reference's default d = 30: scalar staging and ds_read_b32)
    const int S = vec ? ((d & 7) ? d : d + 4) : (d | 1);
    const int j0 = blockIdx.x * DIRECT_TILE;
    const int nj = min(DIRECT_TILE, n_local - j0);
    const int d4 = d >> 2;
    if (vec) {
        for (int i = tid; i < nj * d4; i += 256) { // the tile is nj * d consecutive floats of W
            const int jj = i / d4, c4 = i - jj * d4;
            *reinterpret_cast<float4 *>(tile + jj * S + 4 * c4) = reinterpret_cast<const float4 *>(W + (size_t)j0 * d)[i];
        }
    } else {
        for (int i = tid; i < nj * d; i += 256) {
            const int jj = i / d, c = i - jj * d;
            tile[jj * S + c] = W[(size_t)j0 * d + i];
        }
    }
    if constexpr (MODE == 1) {
        for (int i = tid; i < DIRECT_ROWS * d; i += 256) {
            const int rr = i / d, c = i - rr * d;
            xs[rr * DIRECT_MAX_D + c] = (row + rr < M) ? x[(size_t)(row + rr) * d + c] : 0.f;
        }
        __syncthreads();
        // two rows per wave against the same W row: one LDS read of W feeds two independent chains
        if (lane < nj) {
            const float b0 = bias[j0 + lane];
            float acc0 = b0, acc1 = b0;
            const float *x0 = xs + wave * DIRECT_MAX_D, *x1 = xs + (wave + 4) * DIRECT_MAX_D;
            if (vec) {
                const float4 *w4 = reinterpret_cast<const float4 *>(tile + lane * S);
                for (int c = 0; c < d4; ++c) { // the k-ascending fma chain of irs_chain, twice
                    const float4 wv = w4[c], xa = reinterpret_cast<const float4 *>(x0)[c], xb = reinterpret_cast<const float4 *>(x1)[c];
                    acc0 = __fmaf_rn(xa.x, wv.x, acc0), acc1 = __fmaf_rn(xb.x, wv.x, acc1);
                    acc0 = __fmaf_rn(xa.y, wv.y, acc0), acc1 = __fmaf_rn(xb.y, wv.y, acc1);
                    acc0 = __fmaf_rn(xa.z, wv.z, acc0), acc1 = __fmaf_rn(xb.z, wv.z, acc1);
                    acc0 = __fmaf_rn(xa.w, wv.w, acc0), acc1 = __fmaf_rn(xb.w, wv.w, acc1);
                }
            } else {
                const float *wr = tile + lane * S;
                for (int c = 0; c < d; ++c) acc0 = __fmaf_rn(x0[c], wr[c], acc0), acc1 = __fmaf_rn(x1[c], wr[c], acc1);
            }
            if (row + wave < M) gk[(size_t)wave * (IRS_CAND_CAP * 2) + j0 + lane] = irs_fkey(acc0);
            if (row + wave + 4 < M) gk[(size_t)(wave + 4) * (IRS_CAND_CAP * 2) + j0 + lane] = irs_fkey(acc1);
        }
        return;
    }
    for (int i = tid; i < d; i += 256) xs[i] = x[(size_t)row * d + i];
    __syncthreads();
    DSTAMP(1);
    // the row's score keys (32 bits per item, the item id is the position): 16 KB for the last workgroup to fetch
    if (wave == 0 && lane < nj) {
        float acc = bias[j0 + lane];
        if (vec) {
            const float4 *w4 = reinterpret_cast<const float4 *>(tile + lane * S);
            const float4 *x4 = reinterpret_cast<const float4 *>(xs);
            for (int c = 0; c < d4; ++c) { // the k-ascending fma chain of irs_chain
                const float4 wv = w4[c], xv = x4[c];
                acc = __fmaf_rn(xv.x, wv.x, acc);
                acc = __fmaf_rn(xv.y, wv.y, acc);
                acc = __fmaf_rn(xv.z, wv.z, acc);
                acc = __fmaf_rn(xv.w, wv.w, acc);
            }
        } else {
            const float *wr = tile + lane * S;
            for (int c = 0; c < d; ++c) acc = __fmaf_rn(xs[c], wr[c], acc);
        }
        gk[j0 + lane] = irs_fkey(acc);
    }
    DSTAMP(2);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); // the keys reach memory before the arrival is counted
    __syncthreads();
    DSTAMP(3);
    if (tid == 0) s_last = (atomicAdd(&arrive[row], 1u) == gridDim.x - 1) ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // and are read from memory, not from a stale L2 line
    DSTAMP(4);
    if (tid == 0) arrive[row] = 0u; // ready for the next call (graph replays included)
    } // MODE != 2
    if (tid == 0) status[row] = 0;
    constexpr int KPT = DIRECT_MAX_ITEMS / 256;
    unsigned long long key[KPT]; // this thread's items: 1024 q + 4 tid + e, key = (score key, ~id); 0 where there is none
#pragma unroll
    for (int q = 0; q < KPT / 4; ++q) {
        const int j = q * 1024 + 4 * tid;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (j < n_local) v = *reinterpret_cast<const uint4 *>(gk + j);
        key[4 * q + 0] = (j + 0 < n_local) ? ((unsigned long long)v.x << 32) | (0xFFFFFFFFu - (unsigned int)(j + 0)) : 0ull;
        key[4 * q + 1] = (j + 1 < n_local) ? ((unsigned long long)v.y << 32) | (0xFFFFFFFFu - (unsigned int)(j + 1)) : 0ull;
        key[4 * q + 2] = (j + 2 < n_local) ? ((unsigned long long)v.z << 32) | (0xFFFFFFFFu - (unsigned int)(j + 2)) : 0ull;
        key[4 * q + 3] = (j + 3 < n_local) ? ((unsigned long long)v.w << 32) | (0xFFFFFFFFu - (unsigned int)(j + 3)) : 0ull;
    }
    // A lower bound T <= (k-th largest score key) that keeps barely more than k keys, without a search: take each
    // thread's largest key; the k-th largest of those 256 maxima has at least k keys at or above it, and since a
    // thread's keys are an arbitrary 1/256 sample only ~1.1 k of the row's keys pass it.  The k-th largest maximum is
    // found by rank counting (keys are unique: the item id is part of them).
    unsigned int thr = 0u;
    DSTAMP(5);
    if (n_local >= 4 * k) { // thread t holds items 4t .. 4t+3 (+1024 q): then at least k threads hold a key (k <= 256);
                            // smaller rows (< 4k <= 1024 keys) keep every key and sort them all
        unsigned long long mx = 0ull;
#pragma unroll
        for (int i = 0; i < KPT; ++i) mx = key[i] > mx ? key[i] : mx;
        tmax[tid] = mx;
        __syncthreads();
        int above = 0;
        const ulonglong2 *t2 = reinterpret_cast<const ulonglong2 *>(tmax);
#pragma unroll 8
        for (int i = 0; i < 128; ++i) {
            const ulonglong2 v = t2[i];
            above += (v.x > mx) + (v.y > mx);
        }
        if (above == k - 1) s_thr = (unsigned int)(mx >> 32);
        __syncthreads();
        thr = s_thr;
    } else
        __syncthreads();
    DSTAMP(6);
    // survivors (ties at the threshold included) -> rkeys at slots from an exclusive scan of the per-thread counts
    int mine = 0;
#pragma unroll
    for (int i = 0; i < KPT; ++i) mine += (key[i] != 0ull && (unsigned int)(key[i] >> 32) >= thr) ? 1 : 0;
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    unsigned int slot = incl - mine;
    for (int w = 0; w < wave; ++w) slot += wsum[w];
    unsigned int nr = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    DSTAMP(7);
    if (nr <= IRS_REFINE_CAP) {
#pragma unroll
        for (int i = 0; i < KPT; ++i)
            if (key[i] != 0ull && (unsigned int)(key[i] >> 32) >= thr) rkeys[slot++] = key[i];
        if ((nr & 1) && tid == 0) rkeys[nr] = 0ull; // the rank loop below reads pairs
        __syncthreads();
    }
    if (nr <= 256) {
        // the usual case (~1.1 k survivors): rank by counting, write the winners straight out -- no sort stages
        DSTAMP(8);
        if (tid < (int)nr) {
            const unsigned long long me = rkeys[tid];
            int above = 0;
            const ulonglong2 *r2 = reinterpret_cast<const ulonglong2 *>(rkeys);
            for (int i = 0; i < (int)(nr + 1) / 2; ++i) {
                const ulonglong2 v = r2[i];
                above += (v.x > me) + (v.y > me);
            }
            if (above < k) {
                val[(size_t)row * k + above] = irs_unkey((unsigned int)(me >> 32));
                ids[(size_t)row * k + above] = item_lo + (int64_t)(0xFFFFFFFFu - (unsigned int)me);
            }
        }
        for (int i = (int)nr + tid; i < k; i += 256) {
            val[(size_t)row * k + i] = -INFINITY;
            ids[(size_t)row * k + i] = -1;
        }
    } else {
        const unsigned long long *sorted = rkeys;
        if (nr > IRS_REFINE_CAP) { // > 1024 - k exact ties at the boundary: sort the whole key array in LDS (the W tile is dead)
            unsigned long long *keys = reinterpret_cast<unsigned long long *>(tile);
#pragma unroll
            for (int i = 0; i < KPT; ++i) keys[i * 256 + tid] = key[i]; // all DIRECT_MAX_ITEMS slots, empty ones = 0
            bitonic_desc(keys, DIRECT_MAX_ITEMS);
            sorted = keys;
            nr = (unsigned int)n_local;
        } else {
            int n2 = 2;
            while (n2 < (int)nr) n2 <<= 1;
            for (int i = nr + tid; i < n2; i += 256) rkeys[i] = 0ull;
            bitonic_desc(rkeys, n2);
        }
        DSTAMP(8);
        for (int i = tid; i < k; i += 256) {
            if (i < (int)nr) {
                const unsigned long long kk = sorted[i];
                val[(size_t)row * k + i] = irs_unkey((unsigned int)(kk >> 32));
                ids[(size_t)row * k + i] = item_lo + (int64_t)(0xFFFFFFFFu - (unsigned int)kk);
            } else {
                val[(size_t)row * k + i] = -INFINITY;
                ids[(size_t)row * k + i] = -1;
            }
        }
    }
    if (tid == 0 && (int)nr < k) status[row] |= IRS_ROW_FEWER_THAN_K;
    // path search: the row's step (window filter of the ranked candidates, choice, window update) right here, by
    // the workgroup that has just ranked them -- one launch fewer per step
    if (pa.enabled) {
        __syncthreads(); // val / ids of this row are written
        if (wave == 0) irs_path_step_row(pa, row, lane, val, ids, k);
    }
#ifdef IRS_DIRECT_TIMING
    DSTAMP(9);
    if (tid == 0)
        for (int i = 0; i < 10; ++i) g_direct_t[i] = tstamp[i];
#endif
}

__global__ void __launch_bounds__(256) k_exhaustive(const float *__restrict__ x, int d, const float *__restrict__ W,
                                                    const float *__restrict__ bias, int64_t n_local, int64_t item_lo,
                                                    int k, int only_flagged, float *__restrict__ val,
                                                    int64_t *__restrict__ ids, int32_t *__restrict__ status) {
    __shared__ unsigned long long buf[EXH_BUF];
    __shared__ float xs[256];
    const int row = blockIdx.x;
    if (only_flagged && !(status[row] & IRS_ROW_FALLBACK)) return;
    exhaustive_row(x, d, W, bias, n_local, item_lo, k, row, val, ids, status, buf, xs, true);
}

// top k of every recorded row out of its strips' lists (k_exh_strips); recorded rows beyond EXH_FB_MAX are redone by
// one workgroup each over the whole shard (exhaustive_row)
__global__ void __launch_bounds__(256) k_exh_merge(const float *__restrict__ x, int d, const float *__restrict__ W,
                                                   const float *__restrict__ bias, int64_t n_local, int64_t item_lo, int k,
                                                   int n_strips, const unsigned int *__restrict__ fb_count,
                                                   const int32_t *__restrict__ fb_list, const unsigned long long *__restrict__ exh_keys,
                                                   float *__restrict__ val, int64_t *__restrict__ ids, int32_t *__restrict__ status) {
    __shared__ unsigned long long buf[EXH_BUF];
    __shared__ float xs[256];
    const unsigned int nfb = *fb_count;
    if (nfb == 0u) return;
    for (unsigned int fi = blockIdx.x; fi < nfb; fi += gridDim.x) {
        const int row = fb_list[fi];
        if (fi >= EXH_FB_MAX) {
            exhaustive_row(x, d, W, bias, n_local, item_lo, k, row, val, ids, status, buf, xs, true);
            continue;
        }
        const unsigned long long *src = exh_keys + (size_t)fi * n_strips * k;
        const unsigned int n = exh_select([&](int64_t j) -> unsigned long long { return src[j]; }, 0, (int64_t)n_strips * k, k, buf);
        for (int i = threadIdx.x; i < k; i += 256) {
            if (i < (int)n) {
                const unsigned long long kk = buf[i];
                val[(size_t)row * k + i] = irs_unkey((unsigned int)(kk >> 32));
                ids[(size_t)row * k + i] = item_lo + (int64_t)(0xFFFFFFFFu - (unsigned int)kk);
            } else {
                val[(size_t)row * k + i] = -INFINITY;
                ids[(size_t)row * k + i] = -1;
            }
        }
        if (threadIdx.x == 0 && (int)n < k) status[row] |= IRS_ROW_FEWER_THAN_K;
        __syncthreads();
    }
}

// exact scores at chosen items; -inf outside the shard
__global__ void __launch_bounds__(64) k_gather(const float *__restrict__ x, int d, const float *__restrict__ W,
                                               const float *__restrict__ bias, int64_t item_lo, int64_t n_local,
                                               const int64_t *__restrict__ ids0, int g, float *__restrict__ out) {
    __shared__ float xs[256];
    const int row = blockIdx.x;
    for (int i = threadIdx.x; i < d; i += 64) xs[i] = x[(size_t)row * d + i];
    __syncthreads();
    for (int i = threadIdx.x; i < g; i += 64) {
        int64_t j = ids0[(size_t)row * g + i] - item_lo;
        float e = -INFINITY;
        if (ids0[(size_t)row * g + i] >= 0 && j >= 0 && j < n_local) e = irs_chain(xs, W + (size_t)j * d, bias[j], d);
        out[(size_t)row * g + i] = e;
    }
}

// remove excluded items (history) that rank before the reference from count
__global__ void __launch_bounds__(64) k_count_excl(const float *__restrict__ x, int d, const float *__restrict__ W,
                                                   const float *__restrict__ bias, int64_t item_lo, int64_t n_local,
                                                   const float *__restrict__ ref_score, const int64_t *__restrict__ ref_id0,
                                                   const int64_t *__restrict__ excl, int n_excl,
                                                   unsigned long long *__restrict__ count) {
    __shared__ float xs[256];
    const int row = blockIdx.x;
    for (int i = threadIdx.x; i < d; i += 64) xs[i] = x[(size_t)row * d + i];
    __syncthreads();
    const float rs = ref_score[row];
    const int64_t rid = ref_id0[row];
    const int64_t *ex = excl + (size_t)row * n_excl;
    unsigned int sub = 0;
    for (int i = threadIdx.x; i < n_excl; i += 64) {
        int64_t gid = ex[i];
        int64_t j = gid - item_lo;
        if (gid < 0 || j < 0 || j >= n_local || gid == rid) continue;
        bool dup = false;
        for (int q = 0; q < i; ++q)
            if (ex[q] == gid) { dup = true; break; }
        if (dup) continue;
        float e = irs_chain(xs, W + (size_t)j * d, bias[j], d);
        if (e > rs || (e == rs && gid < rid)) ++sub;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sub += __shfl_xor(sub, off, 64);
    if (threadIdx.x == 0 && sub) atomicAdd(&count[row], (unsigned long long)(-(long long)sub));
}

__global__ void k_localize_ref(const int64_t *__restrict__ ref_id0, int64_t item_lo, int64_t *__restrict__ out, int M) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) out[i] = ref_id0[i] - item_lo; // may be negative / >= n_local: then only the score decides
}

// one workgroup per row: threads stride over the partial (max, sum) pairs with an online rescale, then a fixed tree
__global__ void __launch_bounds__(256) k_lse_reduce(const float *__restrict__ part, int slots, int M_pad, int M,
                                                    float *__restrict__ out_max, float *__restrict__ out_sum) {
    __shared__ float sm[4], ss[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    float m = -INFINITY, acc = 0.f;
    for (int s = tid; s < slots; s += 256) {
        const float2 p = *reinterpret_cast<const float2 *>(part + ((size_t)s * M_pad + row) * 2);
        if (p.x > -INFINITY) {
            const float nm = fmaxf(m, p.x);
            acc = acc * __expf(m - nm) + p.y * __expf(p.x - nm);
            m = nm;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float om = __shfl_xor(m, off, 64), oa = __shfl_xor(acc, off, 64);
        const float nm = fmaxf(m, om);
        acc = (nm > -INFINITY) ? acc * __expf(m - nm) + oa * __expf(om - nm) : 0.f;
        m = nm;
    }
    if ((tid & 63) == 0) {
        sm[tid >> 6] = m;
        ss[tid >> 6] = acc;
    }
    __syncthreads();
    if (tid == 0) {
        float fm = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3])), fa = 0.f;
        for (int w = 0; w < 4; ++w)
            if (sm[w] > -INFINITY) fa += ss[w] * __expf(sm[w] - fm);
        out_max[row] = fm;
        out_sum[row] = fa;
    }
}

// =============================== host side ===============================
static inline int ub_bf16(int KS) { return KS >= 16 ? 4 : 8; }
static inline int ub_f32(int KS) { return KS >= 16 ? 2 : KS >= 8 ? 4 : 8; }

// tiles per wave (a workgroup of the blocked kernels walks 4 of them) for a grid of about `target_wgs` workgroups
static void sweep_decompose(SweepArgs &a, int tile_begin, int tile_end, int n_ublocks_hint, int force_tpw, int stride = 1,
                            int target_wgs = 2048, int max_tpw = 16);

// Workgroups that can be resident at once (occupancy x CUs), per kernel instantiation.
template <typename K>
static int resident_workgroups(K kern, int threads, size_t lds) {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256 * per_cu;
    return prop.multiProcessorCount * per_cu;
}

template <int KS, int RT, int TPS, int NW, int MODE>
static void launch_ring(SweepArgs &a, hipStream_t s) {
    constexpr int NSLOT = 4; // two steps of DMA in flight behind the one being multiplied
    a.n_ublocks = (a.UT + NW * RT - 1) / (NW * RT);
    const size_t lds = (size_t)NSLOT * TPS * KS * 1024 + (size_t)NSLOT * TPS * 256 + (size_t)NW * (EMIT_Q * 12 + 16);
    auto kern = k_sweep_ring<KS, RT, TPS, NW, 2, NSLOT, MODE>;
    static std::atomic<int> slots_dev[IRS_MAX_DEVICES]; // per instantiation and device
    const int dev_ = irs_cur_dev();
    int slots = slots_dev[dev_].load(std::memory_order_acquire);
    if (!slots) {
        if (lds > 65536)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        slots = resident_workgroups(kern, NW * 64, lds);
        slots_dev[dev_].store(slots, std::memory_order_release);
    }
    if (MODE == MODE_EMIT) {
        // Equal strips, sized so that the grid is just under a whole number of rounds of resident workgroups: the
        // time of a sweep is rounds x strip length, and a grid of 2.06 rounds costs 3 (measured: 322 vs 264 us).  Four rounds of shorter strips balance a little
        // better than two or three of longer ones (lab: -4 % at both catalog shapes).
        const int nt = a.tile_end - a.tile_begin;
        int rounds = 4;
        while (rounds > 1 && (long long)nt * a.n_ublocks < (long long)rounds * slots * 16) --rounds; // >= 16 tiles per strip
        int strips = (int)((long long)rounds * slots / a.n_ublocks) & ~7; // strips of one XCD class are multiples of 8
        if (strips < 8) strips = 8;
        a.tiles_per_wg = (nt + strips - 1) / strips;
        if (a.tiles_per_wg < 1) a.tiles_per_wg = 1;
        a.n_strips = (nt + a.tiles_per_wg - 1) / a.tiles_per_wg;
        a.tile_stride = 1;
    } else
        a.tiles_per_wg = 0;
    dim3 grid(((a.n_strips + 7) / 8) * 8 * a.n_ublocks);
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, s, a);
}

// grid of the ring kernels' EMIT pass: equal strips, just under a whole number of rounds of resident workgroups
static void ring_emit_grid(SweepArgs &a, int slots) {
    const int nt = a.tile_end - a.tile_begin;
    int rounds = 4;
    while (rounds > 1 && (long long)nt * a.n_ublocks < (long long)rounds * slots * 16) --rounds; // >= 16 tiles per strip
    int strips = (int)((long long)rounds * slots / a.n_ublocks) & ~7; // strips of one XCD class are multiples of 8
    if (strips < 8) strips = 8;
    a.tiles_per_wg = (nt + strips - 1) / strips;
    if (a.tiles_per_wg < 1) a.tiles_per_wg = 1;
    a.n_strips = (nt + a.tiles_per_wg - 1) / a.tiles_per_wg;
    a.tile_stride = 1;
}

template <int KS, int RT16, int NW, int MODE>
static void launch_ring16(SweepArgs &a, hipStream_t s) {
    constexpr int NSLOT = 4;
    a.n_ublocks = (a.UT + NW * (RT16 / 2) - 1) / (NW * (RT16 / 2));
    const size_t lds = ring16_lds_bytes(KS, RT16, NW, NSLOT);
    auto kern = k_sweep_ring16<KS, RT16, NW, NSLOT, MODE>;
    static std::atomic<int> slots_dev[IRS_MAX_DEVICES]; // per instantiation and device
    const int dev_ = irs_cur_dev();
    int slots = slots_dev[dev_].load(std::memory_order_acquire);
    if (!slots) {
        if (lds > 65536)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        slots = resident_workgroups(kern, NW * 64, lds);
        slots_dev[dev_].store(slots, std::memory_order_release);
    }
    if (MODE == MODE_EMIT) ring_emit_grid(a, slots);
    else a.tiles_per_wg = 0;
    dim3 grid(((a.n_strips + 7) / 8) * 8 * a.n_ublocks);
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, s, a);
}

template <int MODE>
static int launch_sweep_bf16(irs_ctx *ctx, SweepArgs &a, hipStream_t s) {
    const int KS = ctx->KS, UB = ub_bf16(KS);
    // compute-bound regime (>= 256 rows), d_pad >= 32: the ring kernel on 16x16x32 MFMAs (sweep_variant 4: the 32x32x16
    // ring, development A/B only)
    if (a.UT >= 8 && KS >= 2 && ctx->sweep_variant == 0) {
        // 4 row tiles of 16 rows per wave, 256 rows per workgroup: ~170 registers, three workgroups per CU.  (Lab, 1M x 128
        // x 1024 rows: 209 us against 222 us with 8 row tiles per wave at two workgroups per CU, 274 us for the 32x32x16
        // ring; 1.25M x 256: 491 us against 569 us.)
        switch (KS) {
        case 2: launch_ring16<2, 4, 4, MODE>(a, s); break;
        case 4: launch_ring16<4, 4, 4, MODE>(a, s); break;
        case 8: launch_ring16<8, 4, 4, MODE>(a, s); break;
        case 16: launch_ring16<16, 4, 4, MODE>(a, s); break;
        default: IRS_FAIL(ctx, IRS_E_UNSUPPORTED, "unsupported d_pad %d", ctx->d_pad);
        }
        IRS_CHECK_HIP(ctx, hipGetLastError());
        return IRS_OK;
    }
    if (a.UT >= 8 && ctx->sweep_variant != 1 && ctx->sweep_variant != 3) {
        a.no_stagger = ctx->sweep_variant == 2;
        const int w512 = (a.UT + 15) / 16 * 16 - a.UT, w256 = (a.UT + 7) / 8 * 8 - a.UT;
        const bool big = w512 <= w256;
        switch (KS) {
        case 1: big ? launch_ring<1, 4, 1, 4, MODE>(a, s) : launch_ring<1, 2, 2, 4, MODE>(a, s); break;
        case 2: big ? launch_ring<2, 4, 1, 4, MODE>(a, s) : launch_ring<2, 2, 2, 4, MODE>(a, s); break;
        case 4: big ? launch_ring<4, 4, 1, 4, MODE>(a, s) : launch_ring<4, 2, 2, 4, MODE>(a, s); break;
        case 8: big ? launch_ring<8, 4, 1, 4, MODE>(a, s) : launch_ring<8, 2, 2, 4, MODE>(a, s); break;
        case 16: launch_ring<16, 2, 1, 4, MODE>(a, s); break; // 256 rows per workgroup: two row tiles per wave is all the register file holds
        default: IRS_FAIL(ctx, IRS_E_UNSUPPORTED, "unsupported d_pad %d", ctx->d_pad);
        }
        IRS_CHECK_HIP(ctx, hipGetLastError());
        return IRS_OK;
    }
    a.n_ublocks = (a.UT + UB - 1) / UB;
    dim3 grid(((a.n_strips + 7) / 8) * 8 * a.n_ublocks);
    if (a.UT >= UB && KS <= 8 && ctx->sweep_variant == 1) { // previous compute-bound form (kept for A/B measurements)
        const int ST = KS >= 16 ? 2 : 4; // as in the kernel
        const size_t lds_rs = (size_t)2 * ST * KS * 1024 + 2 * ST * 32 * 4 + EMIT_Q_BYTES;
#define R_(KS_, RT_) hipLaunchKernelGGL((k_sweep_bf16_rs<KS_, RT_, MODE>), grid, dim3(256), lds_rs, s, a)
        switch (KS) { // UB = 4 * RT must equal ub_bf16(KS)
        case 1: R_(1, 2); break;
        case 2: R_(2, 2); break;
        case 4: R_(4, 2); break;
        case 8: R_(8, 2); break;
        default: IRS_FAIL(ctx, IRS_E_UNSUPPORTED, "unsupported d_pad %d", ctx->d_pad);
        }
#undef R_
        IRS_CHECK_HIP(ctx, hipGetLastError());
        return IRS_OK;
    }
    size_t lds = (size_t)UB * KS * 1024 + EMIT_Q_BYTES;
#define L_(KS_, UB_)                                                                                           \
    hipLaunchKernelGGL((k_sweep_bf16<KS_, UB_, MODE>), grid, dim3(256), lds, s, a)
    switch (KS) {
    case 1: L_(1, 8); break;
    case 2: L_(2, 8); break;
    case 4: L_(4, 8); break;
    case 8: L_(8, 8); break;
    case 16: L_(16, 4); break;
    default: IRS_FAIL(ctx, IRS_E_UNSUPPORTED, "unsupported d_pad %d", ctx->d_pad);
    }
#undef L_
    IRS_CHECK_HIP(ctx, hipGetLastError());
    return IRS_OK;
}

template <int MODE>
static int launch_sweep_f32(irs_ctx *ctx, SweepArgs &a, hipStream_t s) {
    const int KS = ctx->KS, UB = ub_f32(KS);
    a.n_ublocks = (a.UT + UB - 1) / UB;
    dim3 grid(((a.n_strips + 7) / 8) * 8 * a.n_ublocks);
    size_t lds = (size_t)UB * KS * 2048 + EMIT_Q_BYTES;
    const bool vec = (a.d % 8 == 0) && (a.d == KS * 16) && ((((uintptr_t)a.w32) & 15) == 0);
#define L_(KS_, UB_)                                                                                           \
    do {                                                                                                       \
        if (vec) hipLaunchKernelGGL((k_sweep_f32<KS_, UB_, MODE, true>), grid, dim3(256), lds, s, a);          \
        else hipLaunchKernelGGL((k_sweep_f32<KS_, UB_, MODE, false>), grid, dim3(256), lds, s, a);             \
    } while (0)
    switch (KS) {
    case 1: L_(1, 8); break;
    case 2: L_(2, 8); break;
    case 4: L_(4, 8); break;
    case 8: L_(8, 4); break;
    case 16: L_(16, 2); break;
    default: IRS_FAIL(ctx, IRS_E_UNSUPPORTED, "unsupported d_pad %d", ctx->d_pad);
    }
#undef L_
    IRS_CHECK_HIP(ctx, hipGetLastError());
    return IRS_OK;
}

static void sweep_common(irs_ctx *ctx, SweepArgs &a, const float *xrows, int M) {
    memset(&a, 0, sizeof(a));
    a.wp = ctx->wp;
    a.w32 = ctx->proj_w;
    a.bias = ctx->bias_pad;
    a.xb = ctx->xb;
    a.x32 = xrows;
    a.M = M;
    a.M_pad = (M + 31) & ~31;
    a.UT = a.M_pad / 32;
    a.d = ctx->dims.d;
    a.n_local = ctx->n_local;
    a.cap = IRS_CAND_CAP;
}

// choose tiles per wave so that the grid has >= ~target_wgs workgroups when the catalog allows it
static void sweep_decompose(SweepArgs &a, int tile_begin, int tile_end, int n_ublocks_hint, int force_tpw, int stride,
                            int target_wgs, int max_tpw) {
    a.tile_begin = tile_begin;
    a.tile_end = tile_end;
    a.tile_stride = stride;
    int nt = (tile_end - tile_begin + stride - 1) / stride;
    int tpw = force_tpw;
    if (tpw <= 0) {
        long long t = ((long long)nt * n_ublocks_hint) / (4LL * target_wgs);
        tpw = (int)(t < 1 ? 1 : t > max_tpw ? max_tpw : t);
    }
    a.tiles_per_wave = tpw;
    a.n_strips = (nt + 4 * tpw - 1) / (4 * tpw);
    if (a.n_strips < 1) a.n_strips = 1;
}

// the double-buffered log-sum-exp kernel loads 16-byte fragments: d = d_pad and aligned operands
static bool lse_fast_ok(const irs_ctx *ctx, const SweepArgs &a) {
    return ctx->KS >= 2 && a.d == ctx->KS * 16 && ((((uintptr_t)a.w32) & 15) == 0) && ((((uintptr_t)a.x32) & 15) == 0);
}

static bool lse_ring_ok(const irs_ctx *ctx, const SweepArgs &a) {
    return a.UT == 1 && (ctx->KS == 16 || ctx->KS == 8) && !ctx->lse_no_ring && lse_fast_ok(ctx, a);
}
// waves (= pairs of (max, sum) partial slots) a log-sum-exp sweep may use: the ring form runs 8 waves on every CU
static int lse_wave_budget(const irs_ctx *ctx, const SweepArgs &a) {
    return lse_ring_ok(ctx, a) ? IRS_LSE_SLOTS_RING / 2 : ctx->lse_slots / 2;
}

template <bool EMIT>
static int launch_lse_f32(irs_ctx *ctx, SweepArgs &a, hipStream_t s) {
    const int KS = ctx->KS;
    // at most 32 rows at d_pad = 256: the rows' fragments in registers (k_lse_f32<.., XREG>): 2.5M x 256 x 32 rows 686 -> 644 us.
    // (At d_pad = 128 the same form, with only 16 KB per wave in flight behind the tile being multiplied, was slower: 181 vs 151 us at 1M x 128.)
    if (lse_ring_ok(ctx, a)) { // at most 32 rows (a beam search's): the LDS-ring form
        a.n_ublocks = 1;
        constexpr int SPW = LSE_RING_NW / 4; // strips per workgroup
        const int nwg = (a.n_strips + SPW - 1) / SPW;
        dim3 grid(((nwg + 7) / 8) * 8);
        const size_t lds = (size_t)LSE_RING_NW * LSE_RING_WAVE_B + (size_t)LSE_RING_NW * (EMIT_Q * 12 + 16);
        IRS_ONCE_PER_DEVICE({
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_lse_ring<16, EMIT, LSE_RING_NW, LSE_RING_CK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_lse_ring<8, EMIT, LSE_RING_NW, LSE_RING_CK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        });
        if (KS == 16) hipLaunchKernelGGL((k_lse_ring<16, EMIT, LSE_RING_NW, LSE_RING_CK>), grid, dim3(64 * LSE_RING_NW), lds, s, a);
        else hipLaunchKernelGGL((k_lse_ring<8, EMIT, LSE_RING_NW, LSE_RING_CK>), grid, dim3(64 * LSE_RING_NW), lds, s, a);
        IRS_CHECK_HIP(ctx, hipGetLastError());
        return IRS_OK;
    }
    const bool xreg = a.UT == 1 && KS == 16;
    const int UB = xreg ? 1 : ub_f32(KS);
    a.n_ublocks = (a.UT + UB - 1) / UB;
    dim3 grid(((a.n_strips + 7) / 8) * 8 * a.n_ublocks);
    const size_t lds = (size_t)UB * KS * 2048 + EMIT_Q_BYTES;
#define L_(KS_, UB_, XR_)                                                                                                 \
    do {                                                                                                                  \
        IRS_ONCE_PER_DEVICE((void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_lse_f32<KS_, UB_, EMIT, XR_>),     \
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));             \
        hipLaunchKernelGGL((k_lse_f32<KS_, UB_, EMIT, XR_>), grid, dim3(256), lds, s, a);                                 \
    } while (0)
    switch (KS) {
    case 2: L_(2, 8, false); break;
    case 4: L_(4, 8, false); break;
    case 8: L_(8, 4, false); break;
    case 16: if (xreg) L_(16, 1, true); else L_(16, 2, false); break;
    default: IRS_FAIL(ctx, IRS_E_UNSUPPORTED, "unsupported d_pad %d", ctx->d_pad);
    }
#undef L_
    IRS_CHECK_HIP(ctx, hipGetLastError());
    return IRS_OK;
}

int irs_launch_pack_w(irs_ctx *ctx, hipStream_t s) {
    IRS_CHECK_HIP(ctx, hipMemsetAsync(ctx->wnorm_max, 0, 4 * sizeof(float), s));
    int64_t rows = (int64_t)ctx->n_tiles * 32;
    hipLaunchKernelGGL(k_pack_w, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, ctx->proj_w, ctx->proj_b,
                       ctx->n_local, ctx->n_tiles, ctx->dims.d, ctx->KS, ctx->wp,
                       ctx->bias_pad, reinterpret_cast<unsigned int *>(ctx->wnorm_max));
    IRS_CHECK_HIP(ctx, hipGetLastError());
    return IRS_OK;
}

bool irs_topk_is_direct(const irs_ctx *ctx, int M, int k) {
    return M <= 1024 && ctx->n_local <= DIRECT_MAX_ITEMS && k <= 256 && ctx->dims.d <= DIRECT_MAX_D;
}

// lse_max / lse_sum non-null: the rows' log-sum-exp over the shard as well (beam search).  On the swept path with the
// bf16 filter it comes out of the SAME pass as the candidates: pre-pass and threshold on the bf16 catalog sample as
// usual, then one sweep of the float32 catalog that emits against the threshold and accumulates (max, sum exp).
int irs_launch_topk(irs_ctx *ctx, const float *xrows, int M, int k, int sweep, float *val, int64_t *ids0,
                    int32_t *status, hipStream_t s, const irs_path_args *path, float *lse_max, float *lse_sum, int carry) {
    if (path && !irs_topk_is_direct(ctx, M, k)) IRS_FAIL(ctx, IRS_E_STATE, "fused path step needs the one-launch top-k");
    if (irs_topk_is_direct(ctx, M, k)) {
        // latency path on a small shard: one kernel, no fallback needed
        irs_prof_begin(ctx, IRS_PROF_REFINE, s);
        const unsigned int tiles = (unsigned)((ctx->n_local + DIRECT_TILE - 1) / DIRECT_TILE);
        unsigned int *arrive = reinterpret_cast<unsigned int *>(ctx->step_ctr) + 8;
        const irs_path_args pa = path ? *path : irs_path_args{};
        if (M <= 32)
            hipLaunchKernelGGL(k_topk_direct<0>, dim3(tiles, M), dim3(256), 0, s, xrows, ctx->dims.d, ctx->proj_w, ctx->proj_b,
                               (int)ctx->n_local, ctx->shard.item_lo, k, ctx->cand, arrive, val, ids0, status, pa, M);
        else {
            hipLaunchKernelGGL(k_topk_direct<1>, dim3(tiles, (M + DIRECT_ROWS - 1) / DIRECT_ROWS), dim3(256), 0, s, xrows, ctx->dims.d,
                               ctx->proj_w, ctx->proj_b, (int)ctx->n_local, ctx->shard.item_lo, k, ctx->cand, arrive, val, ids0,
                               status, irs_path_args{}, M);
            hipLaunchKernelGGL(k_topk_direct<2>, dim3(1, M), dim3(256), 0, s, xrows, ctx->dims.d, ctx->proj_w, ctx->proj_b,
                               (int)ctx->n_local, ctx->shard.item_lo, k, ctx->cand, arrive, val, ids0, status, pa, M);
        }
        irs_prof_end(ctx, IRS_PROF_REFINE, s, 2.0 * ctx->dims.d * (double)M * (double)ctx->n_local, 0.0);
        IRS_CHECK_HIP(ctx, hipGetLastError());
        return lse_max ? irs_launch_lse(ctx, xrows, M, lse_max, lse_sum, s) : IRS_OK;
    }
    SweepArgs a;
    sweep_common(ctx, a, xrows, M);
    const int M_pad = a.M_pad;
    const bool fused_lse = lse_max && sweep == IRS_SWEEP_BF16 && lse_fast_ok(ctx, a);
    const int d = ctx->dims.d;
    const int nt = ctx->n_tiles;
    int rc;
    static_assert(IRS_CAND_BUCKETS == 64, "k_prep_x resets one bucket counter per lane");
    // big shards: rows that need the exhaustive path are recorded by k_refine and redone cooperatively (see k_exh_strips)
    const bool coop_fb = ctx->n_local >= IRS_COOP_FALLBACK_MIN_ITEMS && ctx->exh_keys != nullptr;
    // Threshold carry (round 5): inside a path search the rows of step t + 1 are the rows of step t one item later -- a row's k-th
    // score moves by ~0.02 per step while the emission threshold sits ~0.2-0.4 below it (profiles/r05/thr_drift_probe.txt) -- so the
    // pre-pass and the threshold selection run on the first step and then every carry_period-th one; in between the thresholds of
    // the previous step are reused (k_prep_x re-derives the validation level for the new rows' eps).  k_refine validates as always:
    // a threshold that no longer fits costs that row the exhaustive path, never a wrong list.
    // (Only where the threshold is the speculative one of a 1/8 or 1/16 sample -- ~3k-4k candidates per row: shards of 262144 items and
    //  more.  On a smaller shard the pre-pass sees every tile and the threshold is the tight k-th group maximum: the next step's rows
    //  would miss it often, and the pre-pass is cheap there anyway.)
    const bool reuse_thr = carry && ctx->carry_period > 0 && sweep == IRS_SWEEP_BF16 && !fused_lse && ctx->thr_valid && ctx->thr_M == M &&
                           ctx->thr_k == k && ctx->thr_age + 1 < ctx->carry_period && nt >= 2 * 8 * 1024;
    if (sweep == IRS_SWEEP_BF16) {
        // |approx - exact| <= eps[row] for every item of the shard: see k_prep_x
        const float acc_factor = (float)(ctx->d_pad + 8) * 2.384185791015625e-07f; // (d_pad + 8) 2^-22
        hipLaunchKernelGGL(k_prep_x, dim3((M_pad + 3) / 4), dim3(256), 0, s, xrows, M, M_pad, d, ctx->KS,
                           ctx->xb, ctx->eps, ctx->wnorm_max, acc_factor, ctx->cand_cnt, status, coop_fb ? ctx->fb_count : nullptr,
                           reuse_thr ? ctx->thr : nullptr, reuse_thr ? ctx->ref_tmp : nullptr);
    } else {
        IRS_CHECK_HIP(ctx, hipMemsetAsync(status, 0, sizeof(int32_t) * M, s));
        IRS_CHECK_HIP(ctx, hipMemsetAsync(ctx->cand_cnt, 0, sizeof(unsigned int) * (size_t)M_pad * IRS_CAND_BUCKETS, s));
        hipLaunchKernelGGL(k_zero_eps, dim3((M_pad + 255) / 256), dim3(256), 0, s, ctx->eps, M_pad);
    }
    IRS_CHECK_HIP(ctx, hipGetLastError());
    const int UBh = (sweep == IRS_SWEEP_BF16) ? ub_bf16(ctx->KS) : ub_f32(ctx->KS);
    const int nub = (a.UT + UBh - 1) / UBh;

    if (!reuse_thr) {
    // pre-pass over a strided sample of item tiles: >= 1024 tiles (32768 items) or 1/8 of the shard; 1/16 where that still is
    // >= 1024 tiles (shards of 524288 items and more; round 4).  The threshold is then the ceil(4k/16) = 25th largest sampled
    // group maximum per row (r_sel below) instead of the 38th of twice as many: ~4k emitted items per row instead of ~3k (a third
    // more refine work), for a fallback probability of 1.5e-9 per row -- 25 of the first k - 1 catalog items in threshold order
    // would have to fall into the 1/16 sample -- and the pre-pass costs half (10M x 256: 0.5 -> 0.25 ms)
    int nt0 = nt >= 16 * 1024 ? nt / 16 : nt / 8;
    if (nt0 < 1024) nt0 = 1024;
    if (nt0 > nt) nt0 = nt;
    const int stride = nt / nt0; // >= 1; sampled tiles 0, stride, 2 stride, ...
    nt0 = (nt + stride - 1) / stride;
    int tpw0 = (nt0 + 1023) / 1024; // -> at most ~2048 group maxima per row
    sweep_decompose(a, 0, nt, nub, tpw0, stride);
    int n_waves0 = a.n_strips * 4;
    int G = 2 * n_waves0;
    if (G > SEL2_G) IRS_FAIL(ctx, IRS_E_UNSUPPORTED, "pre-pass groups %d > %d", G, SEL2_G);
    a.gm = ctx->gm;
    a.n_groups = G;
    irs_prof_begin(ctx, IRS_PROF_SWEEP, s);
    if (sweep == IRS_SWEEP_BF16) rc = launch_sweep_bf16<MODE_PRE>(ctx, a, s);
    else rc = launch_sweep_f32<MODE_PRE>(ctx, a, s);
    irs_prof_end(ctx, IRS_PROF_SWEEP, s, 2.0 * d * (double)M * nt0 * 32.0,
                 (double)nt0 * 32.0 * ctx->d_pad * (sweep == IRS_SWEEP_BF16 ? 2.0 : 4.0));
    if (rc) return rc;
    // Emission threshold = r-th largest sampled group maximum (minus 2 eps).  r = k is rigorous (>= k items
    // score above it).  With a 1/stride sample, r ~ 3k/stride targets ~3k emitted items per row instead of
    // ~k*stride; k_refine validates that >= k items really lie above it and otherwise hands the row to the
    // exhaustive kernel, so the speculation can cost time but never correctness.
    // (A row falls back when fewer than k items lie above the threshold, i.e. when r_sel of the catalog's k - 1 best items
    //  landed in the 1/stride sample: Binomial(k - 1, 1/stride) >= r_sel.  k = 100: 6e-11 per row at stride 8 with r = 38;
    //  1.1e-5 at stride 16 with r = 19 -- one row in 100 000, seen in the soak run -- so the 1/16 sample targets 4k emitted
    //  items, r = 25: 1.5e-9.)
    int r_sel = k;
    if (stride > 1) {
        r_sel = ((stride >= 16 ? 4 : 3) * k + stride - 1) / stride;
        if (r_sel < 8) r_sel = 8;
        if (r_sel > k) r_sel = k;
    }
    hipLaunchKernelGGL(k_select_thr_bits, dim3((M_pad + 3) / 4), dim3(256), 0, s, ctx->gm, G, M, M_pad, r_sel, ctx->eps,
                       ctx->thr, ctx->ref_tmp);
    IRS_CHECK_HIP(ctx, hipGetLastError());

    ctx->thr_valid = (sweep == IRS_SWEEP_BF16), ctx->thr_M = M, ctx->thr_k = k, ctx->thr_age = 0;
    } else
        ++ctx->thr_age;
    // emission sweep over the whole shard
    a.thr = ctx->thr;
    a.cnt = ctx->cand_cnt;
    a.cand = ctx->cand;
    int lse_slots = 0;
    if (fused_lse) { // bounded number of (max, sum) partials: tiles per wave from the slot budget, like irs_launch_lse
        const int max_waves = lse_wave_budget(ctx, a);
        int tpw = (nt + max_waves - 1) / max_waves;
        if (tpw < 1) tpw = 1;
        sweep_decompose(a, 0, nt, 1, tpw);
        lse_slots = a.n_strips * 4 * 2;
        if (lse_slots > 2 * max_waves) IRS_FAIL(ctx, IRS_E_UNSUPPORTED, "lse slots %d > %d", lse_slots, 2 * max_waves);
        a.lse_part = ctx->lse_part;
    } else
        sweep_decompose(a, 0, nt, nub, 0);
    const double emit_bytes = (double)nt * 32.0 * ctx->d_pad * ((sweep == IRS_SWEEP_BF16 && !fused_lse) ? 2.0 : 4.0);
    irs_prof_begin(ctx, IRS_PROF_SWEEP, s);
    irs_prof_begin(ctx, IRS_PROF_SWEEP_EMIT, s); // (one family is enabled at a time)
    if (fused_lse) rc = launch_lse_f32<true>(ctx, a, s);
    else if (sweep == IRS_SWEEP_BF16) rc = launch_sweep_bf16<MODE_EMIT>(ctx, a, s);
    else rc = launch_sweep_f32<MODE_EMIT>(ctx, a, s);
    irs_prof_end(ctx, IRS_PROF_SWEEP, s, 2.0 * d * (double)M * (double)ctx->n_local, emit_bytes);
    irs_prof_end(ctx, IRS_PROF_SWEEP_EMIT, s, 2.0 * d * (double)M * (double)ctx->n_local, emit_bytes);
    if (rc) return rc;
    if (fused_lse) {
        hipLaunchKernelGGL(k_lse_reduce, dim3(M), dim3(256), 0, s, ctx->lse_part, lse_slots, M_pad, M, lse_max, lse_sum);
        IRS_CHECK_HIP(ctx, hipGetLastError());
    }

    irs_prof_begin(ctx, IRS_PROF_REFINE, s);
    // big shards: rows that need the exhaustive path are recorded by k_refine and redone cooperatively (see k_exh_strips);
    // on small shards one workgroup walks the shard faster than two more launches cost
    const bool coop = coop_fb;
    if (coop && sweep != IRS_SWEEP_BF16) IRS_CHECK_HIP(ctx, hipMemsetAsync(ctx->fb_count, 0, sizeof(unsigned int), s)); // (bf16: k_prep_x reset it)
    hipLaunchKernelGGL(k_refine, dim3(M), dim3(256), 0, s, xrows, d, ctx->proj_w, ctx->proj_b, ctx->cand_cnt, ctx->cand,
                       ctx->eps, ctx->ref_tmp, k, ctx->shard.item_lo, ctx->n_local, val, ids0, status,
                       coop ? ctx->fb_count : nullptr, ctx->fb_list);
    if (coop) {
        int ns = EXH_STRIPS;
        while (ns > 1 && (long long)ns * k > EXH_KEYS_PER_ROW) ns >>= 1;
        hipLaunchKernelGGL(k_exh_strips, dim3(ns, 4), dim3(256), 0, s, xrows, d, ctx->proj_w, ctx->proj_b, ctx->n_local, k, ns,
                           ctx->fb_count, ctx->fb_list, ctx->exh_keys);
        hipLaunchKernelGGL(k_exh_merge, dim3(EXH_FB_MAX), dim3(256), 0, s, xrows, d, ctx->proj_w, ctx->proj_b, ctx->n_local,
                           ctx->shard.item_lo, k, ns, ctx->fb_count, ctx->fb_list, ctx->exh_keys, val, ids0, status);
    }
    irs_prof_end(ctx, IRS_PROF_REFINE, s, 0.0, 0.0); // (rows flagged IRS_ROW_FALLBACK were redone exhaustively: inside k_refine or by the two kernels behind it)
    IRS_CHECK_HIP(ctx, hipGetLastError());
    if (lse_max && !fused_lse) return irs_launch_lse(ctx, xrows, M, lse_max, lse_sum, s);
    return IRS_OK;
}

// exhaustive exact top-k for every row (tests / yard-stick): sweep = -1 through the C ABI
int irs_launch_topk_exhaustive(irs_ctx *ctx, const float *xrows, int M, int k, float *val, int64_t *ids0,
                               int32_t *status, hipStream_t s) {
    IRS_CHECK_HIP(ctx, hipMemsetAsync(status, 0, sizeof(int32_t) * M, s));
    hipLaunchKernelGGL(k_exhaustive, dim3(M), dim3(256), 0, s, xrows, ctx->dims.d, ctx->proj_w, ctx->proj_b,
                       ctx->n_local, ctx->shard.item_lo, k, 0, val, ids0, status);
    IRS_CHECK_HIP(ctx, hipGetLastError());
    return IRS_OK;
}

int irs_launch_gather(irs_ctx *ctx, const float *xrows, int M, const int64_t *ids0, int g, float *out, hipStream_t s) {
    hipLaunchKernelGGL(k_gather, dim3(M), dim3(64), 0, s, xrows, ctx->dims.d, ctx->proj_w, ctx->proj_b,
                       ctx->shard.item_lo, ctx->n_local, ids0, g, out);
    IRS_CHECK_HIP(ctx, hipGetLastError());
    return IRS_OK;
}

int irs_launch_count_before(irs_ctx *ctx, const float *xrows, int M, const float *ref_score, const int64_t *ref_id0,
                            const int64_t *excl, int n_excl, int64_t *count, hipStream_t s) {
    SweepArgs a;
    sweep_common(ctx, a, xrows, M);
    IRS_CHECK_HIP(ctx, hipMemsetAsync(count, 0, sizeof(int64_t) * M, s));
    int64_t *ref_local = reinterpret_cast<int64_t *>(ctx->cand); // scratch: [M] int64
    hipLaunchKernelGGL(k_localize_ref, dim3((M + 255) / 256), dim3(256), 0, s, ref_id0, ctx->shard.item_lo, ref_local, M);
    const int nub = (a.UT + ub_f32(ctx->KS) - 1) / ub_f32(ctx->KS);
    sweep_decompose(a, 0, ctx->n_tiles, nub, 0);
    a.ref_score = ref_score;
    a.ref_id = ref_local;
    a.count = reinterpret_cast<unsigned long long *>(count);
    irs_prof_begin(ctx, IRS_PROF_SWEEP, s);
    int rc = launch_sweep_f32<MODE_COUNT>(ctx, a, s);
    irs_prof_end(ctx, IRS_PROF_SWEEP, s, 2.0 * ctx->dims.d * (double)M * (double)ctx->n_local,
                 (double)ctx->n_local * ctx->dims.d * 4.0);
    if (rc) return rc;
    if (excl && n_excl > 0) {
        hipLaunchKernelGGL(k_count_excl, dim3(M), dim3(64), 0, s, xrows, ctx->dims.d, ctx->proj_w, ctx->proj_b,
                           ctx->shard.item_lo, ctx->n_local, ref_score, ref_id0, excl, n_excl,
                           reinterpret_cast<unsigned long long *>(count));
    }
    IRS_CHECK_HIP(ctx, hipGetLastError());
    return IRS_OK;
}

int irs_launch_dense(irs_ctx *ctx, const float *xrows, int M, float *out, int64_t ld, hipStream_t s) {
    SweepArgs a;
    sweep_common(ctx, a, xrows, M);
    const int nub = (a.UT + ub_f32(ctx->KS) - 1) / ub_f32(ctx->KS);
    sweep_decompose(a, 0, ctx->n_tiles, nub, 0);
    a.dense = out;
    a.ld = ld;
    irs_prof_begin(ctx, IRS_PROF_SWEEP, s);
    int rc = launch_sweep_f32<MODE_DENSE>(ctx, a, s);
    irs_prof_end(ctx, IRS_PROF_SWEEP, s, 2.0 * ctx->dims.d * (double)M * (double)ctx->n_local,
                 (double)ctx->n_local * (ctx->dims.d + (double)M) * 4.0);
    return rc;
}

// ---- projection + cross entropy without the [M, n_item] logits (training side: reference influentialRS.py:252-310,
// evaluator.py:53-92).  The float32 sweeps read project.weight where the caller bound it, so an optimizer step
// that updates it in place is seen at once; the one derived operand they need, the padded bias, is refreshed here.
__global__ void k_refresh_bias(const float *__restrict__ b, int64_t n_local, int64_t n_pad, float *__restrict__ bias_pad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_pad) bias_pad[i] = i < n_local ? b[i] : -INFINITY;
}

int irs_launch_refresh_bias(irs_ctx *ctx, hipStream_t s) {
    const int64_t n_pad = (int64_t)ctx->n_tiles * 32;
    hipLaunchKernelGGL(k_refresh_bias, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, s, ctx->proj_b, ctx->n_local, n_pad,
                       ctx->bias_pad);
    IRS_CHECK_HIP(ctx, hipGetLastError());
    return IRS_OK;
}

// labels0 global 0-based (-1: row ignored) -> local; ignored rows get the sentinel the sweep tests for
__global__ void k_ce_localize(const int64_t *__restrict__ labels0, int64_t item_lo, int64_t *__restrict__ out, int M) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) out[i] = labels0[i] < 0 ? INT64_MIN : labels0[i] - item_lo; // labels >= n_item match no column (reported by irs_ce_forward)
}

// loss[0] = sum over valid rows of (lse - label score), loss[1] = number of valid rows, loss[2] = number of labels
// >= n_item (nn.CrossEntropyLoss raises on those; such a row is left out of [0] and [1]) (one workgroup; M is a batch)
__global__ void __launch_bounds__(256) k_ce_reduce(const float *__restrict__ lse, const float *__restrict__ lab_score,
                                                   const int64_t *__restrict__ labels0, int M, int64_t n_item,
                                                   double *__restrict__ out) {
    __shared__ double ssum[4], scnt[4], sbad[4];
    double acc = 0.0, cnt = 0.0, bad = 0.0;
    for (int i = threadIdx.x; i < M; i += 256) {
        const int64_t l = labels0[i];
        if (l >= n_item) bad += 1.0;
        else if (l >= 0) {
            acc += (double)lse[i] - (double)lab_score[i];
            cnt += 1.0;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        acc += __shfl_xor(acc, off, 64);
        cnt += __shfl_xor(cnt, off, 64);
        bad += __shfl_xor(bad, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        ssum[threadIdx.x >> 6] = acc;
        scnt[threadIdx.x >> 6] = cnt;
        sbad[threadIdx.x >> 6] = bad;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = (ssum[0] + ssum[1]) + (ssum[2] + ssum[3]);
        out[1] = (scnt[0] + scnt[1]) + (scnt[2] + scnt[3]);
        out[2] = (sbad[0] + sbad[1]) + (sbad[2] + sbad[3]);
    }
}

__global__ void k_lse_combine(const float *__restrict__ mx, const float *__restrict__ sm, float *__restrict__ lse, int M) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) lse[i] = mx[i] + logf(sm[i]);
}

int irs_launch_lse_combine(irs_ctx *ctx, const float *mx, const float *sm, float *lse, int M, hipStream_t s) {
    hipLaunchKernelGGL(k_lse_combine, dim3((M + 255) / 256), dim3(256), 0, s, mx, sm, lse, M);
    IRS_CHECK_HIP(ctx, hipGetLastError());
    return IRS_OK;
}

int irs_launch_ce_reduce(irs_ctx *ctx, const float *lse, const float *lab_score, const int64_t *labels0, int M, double *out,
                         hipStream_t s) {
    hipLaunchKernelGGL(k_ce_reduce, dim3(1), dim3(256), 0, s, lse, lab_score, labels0, M, ctx->dims.n_item, out);
    IRS_CHECK_HIP(ctx, hipGetLastError());
    return IRS_OK;
}

int irs_launch_ce_grad(irs_ctx *ctx, const float *xrows, const int64_t *labels0, const float *lse, int M, float scale,
                       float *out, int64_t ld, hipStream_t s) {
    SweepArgs a;
    sweep_common(ctx, a, xrows, M);
    int64_t *lab_local = reinterpret_cast<int64_t *>(ctx->cand); // scratch: [M] int64
    hipLaunchKernelGGL(k_ce_localize, dim3((M + 255) / 256), dim3(256), 0, s, labels0, ctx->shard.item_lo, lab_local, M);
    const int nub = (a.UT + ub_f32(ctx->KS) - 1) / ub_f32(ctx->KS);
    sweep_decompose(a, 0, ctx->n_tiles, nub, 0);
    a.dense = out;
    a.ld = ld;
    a.ce_lse = lse;
    a.ce_label = lab_local;
    a.ce_scale = scale;
    irs_prof_begin(ctx, IRS_PROF_SWEEP, s);
    int rc = launch_sweep_f32<MODE_CEGRAD>(ctx, a, s);
    irs_prof_end(ctx, IRS_PROF_SWEEP, s, 2.0 * ctx->dims.d * (double)M * (double)ctx->n_local,
                 (double)ctx->n_local * (ctx->dims.d + (double)M) * 4.0);
    return rc;
}

int irs_launch_lse(irs_ctx *ctx, const float *xrows, int M, float *out_max, float *out_sum, hipStream_t s) {
    SweepArgs a;
    sweep_common(ctx, a, xrows, M);
    const int nub = (a.UT + ub_f32(ctx->KS) - 1) / ub_f32(ctx->KS);
    // bounded number of partial slots: tiles per wave from the slot budget
    int max_waves = lse_wave_budget(ctx, a);
    int tpw = (ctx->n_tiles + max_waves - 1) / max_waves;
    if (tpw < 1) tpw = 1;
    sweep_decompose(a, 0, ctx->n_tiles, nub, tpw);
    int slots = a.n_strips * 4 * 2;
    if (slots > 2 * max_waves) IRS_FAIL(ctx, IRS_E_UNSUPPORTED, "lse slots %d > %d", slots, 2 * max_waves);
    a.lse_part = ctx->lse_part;
    irs_prof_begin(ctx, IRS_PROF_SWEEP, s);
    int rc = IRS_OK;
    if (lse_fast_ok(ctx, a)) rc = launch_lse_f32<false>(ctx, a, s);
    else rc = launch_sweep_f32<MODE_LSE>(ctx, a, s);
    irs_prof_end(ctx, IRS_PROF_SWEEP, s, 2.0 * ctx->dims.d * (double)M * (double)ctx->n_local,
                 (double)ctx->n_local * ctx->dims.d * 4.0);
    if (rc) return rc;
    hipLaunchKernelGGL(k_lse_reduce, dim3(M), dim3(256), 0, s, ctx->lse_part, slots, a.M_pad, M, out_max,
                       out_sum);
    IRS_CHECK_HIP(ctx, hipGetLastError());
    return IRS_OK;
}
