/*
 * irs_hip.h -- C ABI of libirs_hip.so: the MI355X (gfx950) implementation of
 * InfluentialRS's sequential next-item scoring and persuasion-path search.
 *
 * The reference (JackShDr/InfluentialRS) has NO native/FFI interface for this
 * path: its boundary is the Python class API of model/influentialRS.py
 * (SURVEY.md section 8b, row B1).  This ABI is therefore NEW and sits below
 * that class API; each entry point names the reference call site whose
 * arithmetic it replaces.  The Python front-end (influentialrs_amd/model/)
 * keeps the reference's class/method signatures and calls these through
 * ctypes; INTEGRATION.md shows the binding a maintainer of the reference
 * would add.
 *
 * Conventions
 *  - every function returns 0 on success, a negative IRS_E_* code otherwise;
 *    no C++ exception crosses the ABI; irs_last_error() gives the message.
 *  - every pointer marked "dev" is a device (HBM) pointer owned by the caller
 *    (e.g. torch.Tensor.data_ptr()); the library never frees it.
 *  - nothing is allocated after irs_create(): scratch lives in a caller-owned
 *    workspace sized by irs_workspace_bytes() and bound by irs_bind_workspace().
 *  - all work is enqueued on the caller's stream (a hipStream_t passed as
 *    void*); no call synchronises the device unless documented.
 *  - one context <-> one device, one stream at a time (not re-entrant).  Kernels
 *    are launched on the CURRENT HIP device: the caller makes the device that
 *    owns the context's pointers current around every call (the Python engine
 *    does: influentialrs_amd/engine.py Engine._call).
 *  - item ids crossing the ABI in "ids0" arguments are 0-based GLOBAL catalog
 *    positions (reference item id = ids0 + 1, influentialRS.py:376,422;
 *    0 is the pad id in sequences, which carry 1-based ids like the reference).
 *  - total order used by every selection: score descending, then id
 *    ascending (torch's own tie order is unspecified).
 *  - indices are the caller's responsibility, as with any device gather: sequence entries in [0, n_item]
 *    (0 = pad), user ids in [0, n_user), label / candidate ids0 in [0, n_item) or negative where documented;
 *    nn.Embedding's IndexError has no device-side counterpart here.  Shapes and buffer sizes ARE checked
 *    (IRS_E_INVALID).
 */
#ifndef IRS_HIP_H
#define IRS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IRS_ABI_VERSION 1

/* error codes */
#define IRS_OK 0
#define IRS_E_INVALID (-1)     /* bad argument / shape */
#define IRS_E_STATE (-2)       /* weights or workspace not bound */
#define IRS_E_HIP (-3)         /* HIP runtime error (message has the hipError) */
#define IRS_E_UNSUPPORTED (-4) /* shape outside the kernels' envelope */

/* attention mask flavours */
#define IRS_MASK_IRN 0    /* InfluentialNet as called: allowed = r_u[b], last column = 1.0
                             (_generate_square_subsequent_mask, influentialRS.py:120-155,
                             called at :183-184) + key padding (:171) */
#define IRS_MASK_CAUSAL 1 /* SampleNet: 0 / -inf causal (uRS.py:47-50) + key padding (:53) */

/* scoring precision of the catalog sweep */
#define IRS_SWEEP_BF16 0 /* bf16 MFMA filter with a proven error bound, then exact fp32
                            re-scoring of the survivors: results identical to IRS_SWEEP_F32 */
#define IRS_SWEEP_F32 1  /* fp32 MFMA sweep (exact k-ordered fma chain) */
#define IRS_SWEEP_EXHAUSTIVE 2 /* irs_score_topk only: every item through the exact VALU chain,
                                  one workgroup per row (fallback kernel; yard-stick in tests) */

/* per-row status bits written by the selection kernels */
#define IRS_ROW_OK 0
#define IRS_ROW_FALLBACK 1     /* candidate buffers overflowed; row was re-done by the exhaustive exact kernel */
#define IRS_ROW_NO_CANDIDATE 2 /* every one of the k candidates is in the window
                                  (the reference raises IndexError at influentialRS.py:429) */
#define IRS_ROW_FEWER_THAN_K 4 /* catalog shard has fewer than k items; tail filled with (-inf, -1) */

#define IRS_MAX_SAMPLE_K 8 /* sampled path steps draw among at most this many survivors (reference default: 3);
                              larger sample_k -> IRS_E_UNSUPPORTED */

typedef struct irs_ctx irs_ctx;

typedef struct irs_dims {
    int64_t n_item;   /* catalog size N (global) */
    int64_t n_user;   /* rows of user_embedder (0 for SampleNet) */
    int32_t d;        /* emb_dim (even, <= 256) */
    int32_t max_len;  /* L (<= 256) */
    int32_t n_heads;  /* H, divides d; d / H <= 64 */
    int32_t ffn_dim;  /* F */
    int32_t n_layers;
    int32_t u_dim;    /* u_emb_dim (0 for SampleNet) */
    int32_t mask_mode;/* IRS_MASK_* */
    int32_t max_rows; /* upper bound on scored rows per call */
    int32_t max_k;    /* upper bound on k of irs_score_topk (reference: 100) */
    int32_t max_seqs; /* upper bound on sequences per irs_decode call (0 -> max_rows) */
} irs_dims;

/* item-dimension shard held by this context: rows
 * [item_lo, item_hi) of project.weight / project.bias. */
typedef struct irs_shard {
    int32_t rank;
    int32_t world;
    int64_t item_lo;
    int64_t item_hi;
} irs_shard;

int irs_abi_version(void);
const char *irs_last_error(const irs_ctx *ctx); /* ctx may be NULL: last create() error */

int irs_create(irs_ctx **out, const irs_dims *dims, const irs_shard *shard);
void irs_destroy(irs_ctx *ctx);

/* ---- weights -----------------------------------------------------------
 * Bind one float32 device tensor by its reference state_dict key
 * (item_embedder.weight | word_embedder.weight,
 * user_embedder.weight, pos_embedder.pe, user_mask_layer.{weight,bias},
 * project.{weight,bias}, decoder.layers.<l>.{self_attn,multihead_attn}.
 * {in_proj_weight,in_proj_bias,out_proj.weight,out_proj.bias},
 * decoder.layers.<l>.linear{1,2}.{weight,bias}, decoder.layers.<l>.norm{1,2,3}.
 * {weight,bias}); a leading "module." (nn.DataParallel, pipeline.py:43-44) is
 * accepted.  project.weight / project.bias are the LOCAL shard rows
 * ([item_hi-item_lo, d] / [item_hi-item_lo]).  numel is checked. */
int irs_bind_weight(irs_ctx *ctx, const char *name, const float *dev_ptr, int64_t numel);

/* Bytes of the derived-weight arena (bf16 MFMA-fragment-packed copy of the
 * project shard, padded bias, per-layer cross-attention constants). */
size_t irs_derived_bytes(const irs_ctx *ctx);
/* Build the derived weights into the caller's arena (async on stream).
 * Must be called after all weights are bound and again after any weight
 * changes in place. */
int irs_finalize_weights(irs_ctx *ctx, void *dev_arena, size_t bytes, void *stream);

size_t irs_workspace_bytes(const irs_ctx *ctx);
int irs_bind_workspace(irs_ctx *ctx, void *dev_ws, size_t bytes);

/* ---- decoder (replaces InfluentialNet.decoding, influentialRS.py:157-200;
 *      SampleNet.decoding, uRS.py:52-64) --------------------------------- */

/* r_u[b] = user_mask_layer(user_embedder(user[b]))  (influentialRS.py:180). */
int irs_pif(irs_ctx *ctx, const int64_t *dev_user, int32_t B, float *dev_r_u, void *stream);

/* Full decoder over B sequences.
 *  dev_seq   int64 [B, L] row-major, 1-based item ids, 0 = pad (not modified)
 *  dev_user  int64 [B] (ignored for IRS_MASK_CAUSAL, may be NULL)
 *  dev_x     float [B, L, d] out, may be NULL
 *  dev_pos   int32 [B] row to extract per sequence, may be NULL
 *  dev_xrows float [B, d] out: x[b, pos[b], :], may be NULL (with dev_pos)
 *  dev_r_u   float [B] out, may be NULL */
int irs_decode(irs_ctx *ctx, const int64_t *dev_seq, const int64_t *dev_user, int32_t B, float *dev_x,
               const int32_t *dev_pos, float *dev_xrows, float *dev_r_u, void *stream);

/* ---- scoring against the catalog shard (replaces `project`,
 *      influentialRS.py:83/214, uRS.py:45/68, and the selections on it) ---
 * Exact score of row m and local item j:
 *   e = fmaf(x[d-1], W[j][d-1], ... fmaf(x[0], W[j][0], b[j]))   (float32, k ascending)
 * identical bit for bit to oracle/oracle_score.c. */

/* top-k per row (replaces softmax + topk(100), influentialRS.py:418-421; the
 * softmax is monotone, so ids follow the logits).
 *  dev_xrows float [M, d]
 *  dev_val   float [M, k] out, exact scores, descending
 *  dev_ids0  int64 [M, k] out, global 0-based ids (-1 where the shard has < k items)
 *  dev_status int32 [M] out, IRS_ROW_* bits */
int irs_score_topk(irs_ctx *ctx, const float *dev_xrows, int32_t M, int32_t k, int32_t sweep, float *dev_val,
                   int64_t *dev_ids0, int32_t *dev_status, void *stream);
/* The same call for rows that are the PREVIOUS irs_score_topk[_carry] call's rows one path-search step later (same M, k, IRS_SWEEP_BF16):
 * the pre-pass over the catalog sample and the per-row threshold selection are skipped and the previous call's emission thresholds
 * reused (refreshed every IRS_THR_CARRY-th call, default 8); irs_generate_paths[_sharded] do this between their own steps.  Results are
 * exact whatever the rows are: a threshold that no longer fits is detected by the same validation as always and that row takes the
 * exhaustive path (IRS_ROW_FALLBACK) -- rows unrelated to the previous call's only cost time. */
int irs_score_topk_carry(irs_ctx *ctx, const float *dev_xrows, int32_t M, int32_t k, int32_t sweep, float *dev_val,
                         int64_t *dev_ids0, int32_t *dev_status, void *stream);

/* Exact scores at chosen items (replaces prob_dict[j][end][item-1],
 * evaluator.py:203-205, and the label lookup of influentialRS.py:386-388).
 *  dev_ids0 int64 [M, g] global 0-based ids; entries outside this shard (or <0)
 *  produce -inf so that an elementwise max over shards assembles the answer. */
int irs_score_gather(irs_ctx *ctx, const float *dev_xrows, int32_t M, const int64_t *dev_ids0, int32_t g,
                     float *dev_out, void *stream);

/* count[m] = #{ local j not in excl[m] : (e_j, j) ranks before (ref_score[m], ref_id0[m]) }
 * (replaces sort + history filter + nonzero, influentialRS.py:375-389,
 * evaluator.py:121-131,266-286; rank = 1 + sum of count over shards).
 *  dev_excl_ids0 int64 [M, n_excl] global 0-based ids, -1 = unused slot; duplicates allowed. */
int irs_score_count_before(irs_ctx *ctx, const float *dev_xrows, int32_t M, const float *dev_ref_score,
                           const int64_t *dev_ref_id0, const int64_t *dev_excl_ids0, int32_t n_excl,
                           int64_t *dev_count, void *stream);

/* Dense logits of the shard: out[m, j] for j in [0, n_local) with leading
 * dimension ld (>= n_local); the API-compatible `forward()` output
 * (influentialRS.py:214).  HBM-write bound by construction. */
int irs_score_dense(irs_ctx *ctx, const float *dev_xrows, int32_t M, float *dev_out, int64_t ld, void *stream);

/* Row-wise (max, sum exp(e - max)) over the shard (replaces Softmax /
 * LogSoftmax over N, influentialRS.py:418, evaluator.py:195, and
 * CrossEntropyLoss, evaluator.py:321).  Combine shards on the host side
 * (max of maxes, rescaled sum). */
int irs_score_lse(irs_ctx *ctx, const float *dev_xrows, int32_t M, float *dev_max, float *dev_sumexp, void *stream);

/* irs_score_topk and irs_score_lse of the same rows out of ONE call: what a beam-search step needs (candidates +
 * the log-softmax normaliser, influentialRS.py:418-421 with LogSoftmax instead of Softmax).  On the swept path
 * with IRS_SWEEP_BF16 the threshold still comes from the bf16 pre-pass, but candidates and (max, sum exp) come
 * out of a single pass over the float32 catalog; results equal the two separate calls' (ids and values bit for
 * bit; the pair (max, sum) to float32 rounding). */
int irs_score_topk_lse(irs_ctx *ctx, const float *dev_xrows, int32_t M, int32_t k, int32_t sweep, float *dev_val,
                       int64_t *dev_ids0, int32_t *dev_status, float *dev_max, float *dev_sumexp, void *stream);

/* Training side: CrossEntropyLoss(project(x)[valid], label - 1) of IRSNN.train_batch /
 * get_loss_on_eval_data (influentialRS.py:252-310) and Evaluator.train_batch (evaluator.py:53-92) WITHOUT the
 * [M, n_item] logits the reference materialises.  One device holds the whole catalog.  project.weight / bias are
 * read where they were bound: an optimizer step that updates them in place needs no re-finalisation for these two.
 * They do mark the context's derived catalog (bf16 copy, filter norms) as possibly stale: the entry points that
 * filter through it (irs_score_topk / _topk_lse with IRS_SWEEP_BF16, irs_generate_paths, irs_beam_search) return
 * IRS_E_STATE until irs_finalize_weights has run again.
 *  irs_ce_forward:  dev_labels0 int64 [M] 0-based, -1 = row ignored (a pad target);
 *                   dev_lse float [M] = log sum_j exp(logit_mj); dev_label_score float [M];
 *                   dev_loss double [3] = { sum over valid rows of (lse - label score), number of valid rows,
 *                   number of labels >= n_item (nn.CrossEntropyLoss raises on those; they count in neither sum) }.
 *  irs_ce_grad_logits: dL/dlogits of rows [0, M) (a chunk the caller sizes: M x ld floats), written ONCE by the
 *                   fp32-MFMA sweep's epilogue: scale * (exp(logit - lse) - [item == label]); ignored rows 0.
 *                   The caller finishes with two plain GEMMs (dX = G W, dW = G^T X) and a column sum (db). */
int irs_ce_forward(irs_ctx *ctx, const float *dev_xrows, const int64_t *dev_labels0, int32_t M, float *dev_lse,
                   float *dev_label_score, double *dev_loss, void *stream);
int irs_ce_grad_logits(irs_ctx *ctx, const float *dev_xrows, const int64_t *dev_labels0, const float *dev_lse, int32_t M,
                       float scale, float *dev_out, int64_t ld, void *stream);

/* Merge W per-shard top-k lists (after the RCCL all-gather) into
 * the global top-k with the same total order.
 *  dev_val_in float [W, M, k], dev_ids_in int64 [W, M, k] (ids -1 ignored) */
int irs_merge_topk(irs_ctx *ctx, const float *dev_val_in, const int64_t *dev_ids_in, int32_t W, int32_t M, int32_t k,
                   float *dev_val, int64_t *dev_ids0, void *stream);

/* Wire format of a per-shard list for the exchange step (SURVEY 8e budgets 8 bytes per entry): ONE unsigned 64-bit
 * key per entry = (order key of the float32 score << 32) | (0xFFFFFFFF - global id0); 0 = no entry (id -1).
 * Unsigned key order IS the selection order (score descending, id ascending), so a merge is a sort of keys.
 * irs_pack_topk: n = M * k entries -> dev_keys uint64 [n].
 * irs_merge_topk_keys: dev_keys_in uint64 [W, M, k] (what an all-gather / all-to-all of packed lists delivers)
 *                      -> the merged top-k as (dev_val float [M, k], dev_ids0 int64 [M, k]); W * k <= 2048. */
int irs_pack_topk(irs_ctx *ctx, const float *dev_val, const int64_t *dev_ids0, int64_t n, uint64_t *dev_keys, void *stream);
int irs_merge_topk_keys(irs_ctx *ctx, const uint64_t *dev_keys_in, int32_t W, int32_t M, int32_t k, float *dev_val,
                        int64_t *dev_ids0, void *stream);

/* ---- evaluation batch construction on the device (replaces the per-user Python of
 *      DataProvider.get_random_evaluate_data, data_provider.py:398-449, and
 *      DataLoaderEvalIRS._collate_fn, data_provider.py:591-617) -------------
 * User b's events are dev_items[dev_offsets[b] .. dev_offsets[b+1]) (1-based ids, oldest first).
 *   label[b]  = the last event; history = all events before it; raw window = its last raw_len items
 *   target[b] = dev_targets_in[b] when given, else a uniformly random member of [1, n_item] (or of
 *               dev_pool[0 .. n_pool), the reference's `popular_item` restriction) that is absent from the raw
 *               window -- device counter RNG keyed by (seed, b): distributional parity with random.sample
 *   seq[b]    = [0 .. 0, last (L - gap_len - 1) raw items, gap_len zeros, target],  L = max_len
 *  dev_seq int64 [B, L], dev_target / dev_label int64 [B] out
 *  dev_raw int64 [B, raw_len] out, right-aligned zero-padded raw windows, dev_raw_n int32 [B] their lengths (both may be NULL)
 *  dev_status int32 [B] in/out (may be NULL): IRS_ROW_NO_CANDIDATE is OR-ed in when no target was found
 * Needs L - gap_len - 1 >= 1 (the reference's slice arithmetic is only meaningful there). */
int irs_build_eval_batch(irs_ctx *ctx, const int64_t *dev_items, const int64_t *dev_offsets, int32_t B, int32_t raw_len,
                         int32_t gap_len, const int64_t *dev_targets_in, const int64_t *dev_pool, int64_t n_pool,
                         uint64_t seed, int64_t *dev_seq, int64_t *dev_target, int64_t *dev_label, int64_t *dev_raw,
                         int32_t *dev_raw_n, int32_t *dev_status, void *stream);

/* ---- one step of the persuasion-path search (replaces the per-row body of
 *      IRSNN.get_seq_in_batch, influentialRS.py:419-450) ------------------
 * For each row: drop candidates present in seq[b, :hep[b]+1], take the first
 * survivor (greedy) or draw one of the first sample_k survivors with
 * probability proportional to exp(val) (sample != 0; device counter RNG,
 * distributional parity only), record it in paths[b, step], then grow the
 * window (hep < L-2) or shift it left keeping the target at [L-1].
 *  dev_seq  int64 [B, L] in/out     dev_hep int32 [B] in/out
 *  dev_val  float [B, k], dev_ids0 int64 [B, k]: merged top-k (descending)
 *  dev_paths float [B, path_ld] out (ids stored as float32 like the reference, :407)
 *  dev_status int32 [B] in/out: IRS_ROW_NO_CANDIDATE is OR-ed in */
int irs_path_step(irs_ctx *ctx, int64_t *dev_seq, int32_t *dev_hep, int32_t B, const float *dev_val,
                  const int64_t *dev_ids0, int32_t k, int32_t step, float *dev_paths, int32_t path_ld, int32_t sample,
                  int32_t sample_k, uint64_t seed, int32_t *dev_status, void *stream);

/* Whole greedy/sampled path generation on ONE device holding the full catalog
 * (world == 1): max_path_len x { decode, top-k, path step } enqueued on the
 * stream; with use_graph != 0 one step is captured once into a hipGraph and
 * replayed.  dev_seq is the working window (modified); dev_hep int32 [B]
 * initial history end positions (L - gap_len - 2, per row). */
int irs_generate_paths(irs_ctx *ctx, int64_t *dev_seq, const int64_t *dev_user, int32_t *dev_hep, int32_t B,
                       int32_t max_path_len, int32_t k, int32_t sweep, int32_t sample, int32_t sample_k,
                       uint64_t seed, int32_t use_graph, float *dev_paths, int32_t *dev_status, void *stream);

/* ---- beam search over persuasion paths (BUILD-DEFINED: the reference has no beam
 *      search -- SURVEY fact 4; BASELINE.json config 5).  Beam width W <= 32, path
 *      length P <= 64.  W == 1 is the greedy search of irs_generate_paths id for id.
 * One step for B users x W beams given each beam row's merged top-k (descending) and,
 * for W > 1, its row-wise (max, sum exp) over the whole catalog:
 *   candidates = first W window-survivors of every live beam, scored
 *   cum + (val - max - log(sumexp)); best W by (score desc, parent asc, rank asc) survive.
 * State is ping-ponged: *_in -> *_out ([B, W, ...] row-major; cum = -inf marks a dead beam). */
int irs_beam_step(irs_ctx *ctx, const int64_t *dev_seq_in, const int32_t *dev_hep_in, const double *dev_cum_in,
                  const float *dev_paths_in, const float *dev_val, const int64_t *dev_ids0, const float *dev_lse_max,
                  const float *dev_lse_sum, int32_t B, int32_t W, int32_t k, int32_t step, int32_t P,
                  int64_t *dev_seq_out, int32_t *dev_hep_out, double *dev_cum_out, float *dev_paths_out,
                  int32_t *dev_status, void *stream);

/* Whole beam search on ONE device holding the full catalog: P x { decode B*W windows, top-k,
 * log-sum-exp, beam step } on the stream; use_graph != 0 captures a two-step hipGraph once and
 * replays it.  Needs max_seqs >= B*W and max_rows >= B*W.
 *  dev_seq0 int64 [B, L], dev_user int64 [B], dev_hep0 int32 [B]  (not modified)
 *  dev_paths float [B, W, P] out (beam 0 = best), dev_scores double [B, W] out,
 *  dev_seq_final int64 [B, W, L] out (may be NULL), dev_status int32 [B] out */
int irs_beam_search(irs_ctx *ctx, const int64_t *dev_seq0, const int64_t *dev_user, const int32_t *dev_hep0, int32_t B,
                    int32_t W, int32_t P, int32_t k, int32_t sweep, int32_t use_graph, float *dev_paths,
                    double *dev_scores, int64_t *dev_seq_final, int32_t *dev_status, void *stream);

/* ---- multi-GPU: the exchange steps and the sharded search loops below the ABI (section 8 row B2's
 *      `allgather_merge(ctx, comm, ...)`).  One process per GPU; rank r holds item rows [item_lo, item_hi) (irs_shard).
 * A communicator is either RCCL (librccl.so is dlopen()ed on first use: ncclCommInitRank over a 128-byte unique id the
 * caller distributes, e.g. with its torch.distributed store) or a set of caller-supplied collectives (the CPU
 * rehearsal tests run gloo through it).  Every collective is enqueued on the caller's stream: decode -> row
 * all-gather -> shard sweep -> pack -> key exchange -> merge -> path / beam step is ONE stream-ordered sequence over
 * buffers of the context's workspace, nothing is allocated per call, and with RCCL the whole step can be captured
 * into a hipGraph (use_graph).  The reference has no counterpart (nn.DataParallel only, pipeline.py:43-44). */
typedef struct irs_comm irs_comm;
#define IRS_COMM_ID_BYTES 128
#define IRS_REDUCE_SUM 0
#define IRS_REDUCE_MAX 1
/* all-gather: every rank contributes bytes_per_rank bytes, receives world * bytes_per_rank (rank-major);
 * all-to-all: slice j of send (bytes_per_rank bytes) goes to rank j, slice i of recv came from rank i;
 * all-reduce: in place over `count` float32.  Device pointers; ordered against `stream`; return 0 on success. */
typedef int (*irs_allgather_fn)(void *user, const void *dev_send, void *dev_recv, size_t bytes_per_rank, void *stream);
typedef int (*irs_alltoall_fn)(void *user, const void *dev_send, void *dev_recv, size_t bytes_per_rank, void *stream);
typedef int (*irs_allreduce_f32_fn)(void *user, float *dev_buf, size_t count, int op, void *stream);

int irs_comm_unique_id(void *out_id128);                        /* RCCL: ncclGetUniqueId (call on one rank) */
int irs_comm_init_rccl(irs_comm **out, const void *id128, int32_t rank, int32_t world); /* ncclCommInitRank on the CURRENT device */
int irs_comm_init_callbacks(irs_comm **out, int32_t rank, int32_t world, void *user, irs_allgather_fn allgather,
                            irs_alltoall_fn alltoall, irs_allreduce_f32_fn allreduce);
void irs_comm_destroy(irs_comm *comm);
const char *irs_comm_last_error(void);
int irs_comm_is_rccl(const irs_comm *comm);
/* how irs_exchange_topk moves the keys: 0 = caller-supplied callbacks, 1 = ncclAllToAll (an RCCL extension), 2 = grouped
 * ncclSend / ncclRecv (libraries without the extension; IRS_RCCL_NO_ALLTOALL=1 in the environment at irs_comm_init_rccl
 * forces it, so that the fallback can be tested on a library that has the extension) */
int irs_comm_exchange_kind(const irs_comm *comm);
/* NCCL version code of the librccl.so in use (0 before the first irs_comm_unique_id / irs_comm_init_rccl).  The library is
 * refused at load time unless its major version is the one csrc/comm.hip was compiled against (rccl.h: 2.x). */
int irs_comm_rccl_version(void);

/* rows decoded data-parallel -> all rows on every rank, rank-major: dev_rows_all float [world * B, d]. */
int irs_allgather_rows(irs_ctx *ctx, irs_comm *comm, const float *dev_rows_local, int32_t B, float *dev_rows_all, void *stream);
/* packed per-shard lists of ALL rows (dev_keys_send uint64 [world, B, k], rank-major rows: what irs_pack_topk makes of
 * this rank's irs_score_topk over the gathered rows) -> the world's lists of THIS rank's B rows (dev_keys_recv
 * uint64 [world, B, k], shard-major): one all-to-all of M * k * 8 bytes per rank (SURVEY 8e's budget). */
int irs_exchange_topk(irs_ctx *ctx, irs_comm *comm, const uint64_t *dev_keys_send, uint64_t *dev_keys_recv, int32_t B,
                      int32_t k, void *stream);

/* irs_generate_paths over an item-sharded catalog (replaces the loop of IRSNN.get_seq_in_batch, influentialRS.py:412-450):
 * this rank's B users, per step { decode, row all-gather, sweep of the local shard for all world * B rows, pack,
 * all-to-all of keys, merge, path step }.  Needs max_rows >= world * B, max_seqs >= B; every rank calls it with the
 * same B / max_path_len / k / sweep.  Results equal the single-device irs_generate_paths bit for bit. */
int irs_generate_paths_sharded(irs_ctx *ctx, irs_comm *comm, int64_t *dev_seq, const int64_t *dev_user, int32_t *dev_hep,
                               int32_t B, int32_t max_path_len, int32_t k, int32_t sweep, int32_t sample, int32_t sample_k,
                               uint64_t seed, int32_t use_graph, float *dev_paths, int32_t *dev_status, void *stream);

/* irs_beam_search over an item-sharded catalog.  split_decode == 0: this rank's OWN B users (rows = B * W per rank;
 * all-gather of rows, all-to-all of keys, all-reduce of the rows' (max, sum exp)); needs max_rows >= world * B * W.
 * split_decode != 0 (BASELINE configs[4]: ONE user's beams spread over the node): every rank passes the SAME B users
 * and keeps the same beam state; per step rank r decodes rows [r R/world, (r+1) R/world) of the R = B * W beam windows
 * (R a multiple of world), the rows are all-gathered, every rank sweeps its shard for all R rows, the packed lists are
 * all-gathered and merged on every rank, and the (deterministic) beam step runs replicated; needs max_rows >= R,
 * max_k * world <= 2048.  Outputs as irs_beam_search (identical on every rank when split_decode). */
int irs_beam_search_sharded(irs_ctx *ctx, irs_comm *comm, const int64_t *dev_seq0, const int64_t *dev_user,
                            const int32_t *dev_hep0, int32_t B, int32_t W, int32_t P, int32_t k, int32_t sweep,
                            int32_t split_decode, int32_t use_graph, float *dev_paths, double *dev_scores,
                            int64_t *dev_seq_final, int32_t *dev_status, void *stream);

/* Opt-in overlap of irs_generate_paths_sharded's collectives with its compute (round 5; greedy choice only): the step's users run
 * as TWO micro-batches, the row all-gather and the key all_to_all of one on a side stream (chained by events) while the other
 * decodes / sweeps / merges on the caller's stream -- per step one all-gather and one all_to_all leave the critical path.  Same
 * results bit for bit (rows are independent).  Off by default: two micro-batches cost decoder efficiency on small batches, and
 * the build loop has one GPU (the overlap itself has never been measured).  Stream launches only (use_graph is ignored). */
int irs_set_sharded_overlap(irs_ctx *ctx, int32_t on);
int irs_get_sharded_overlap(const irs_ctx *ctx);

/* State of the sharded loops' step capture (tests / diagnostics): bit 0 = a captured step is held, bit 1 = capture was
 * attempted and refused by the collective library (plain stream launches from then on). */
int irs_sharded_graph_state(const irs_ctx *ctx);

/* ---- decoder GEMM arithmetic -------------------------------------------
 * The throughput path's fused layer kernel (d = 128, ffn = 256: out-projection, both layer norms, feed-forward, the
 * next layer's q | k | v) multiplies in one of three ways; all accumulate in float32 and agree to ~1e-6 relative on the
 * decoder rows (tests/test_gpu_decoder_path.py holds the bound):
 *   IRS_GEMM_H3  (default since round 4) two float16 planes per operand, three products: see the constant below;
 *   IRS_GEMM_X6  every float32 operand is split exactly into three bf16 planes (h + m + l) and the six
 *                leading products hh, hm, mh, hl, lh, mm are summed on v_mfma_f32_32x32x16_bf16: float32-grade products
 *                at 6/16 of the float32-MFMA instruction time;
 *   IRS_GEMM_F32 v_mfma_f32_32x32x2f32.
 * The initial mode is IRS_GEMM_H3 unless the environment holds IRS_DECODER_GEMM=x6 or =f32 when the context is created
 * (any other value than h3 / x6 / f32 fails irs_create; the same holds for IRS_ATTN_GEMM).
 * Changing the mode drops the context's captured steps (they are re-captured on the next graph call).
 * Attention of the throughput path at head dim 32 behind a split-precision layer kernel (environment IRS_ATTN_GEMM, read at
 * creation): default "h3" -- scores on the float32 matrix chain, O^T += V^T P^T on float16 plane pairs, the V section of a
 * q | k | v row being WRITTEN as [32 f16 h | 32 f16 l] per (token, head) by the layer kernel (same 128 bytes; rows within 6e-6
 * of the float32 attention's); "f32": the float32-MFMA attention on float32 q | k | v rows.  (Round 5 removed the variants that
 * lost twice -- split-bf16 attention "x6", the persistent work-list attention IRS_ATTN_PERSIST, the one-wave d = 256 layer kernel
 * IRS_X6D_MIN_ROWS: their records are under profiles/r03, profiles/r04 and in HISTORY.md.)  The q | k | v buffer is internal to
 * irs_decode: no caller sees the plane format. */
#define IRS_GEMM_F32 0
#define IRS_GEMM_X6 1
/* IRS_GEMM_H3 (round 4): two FLOAT16 planes per float32 operand (h = f16(x), l = f16(x - h)) and the three leading products hh,
 * hl, lh on v_mfma_f32_32x32x16_f16 -- half the matrix instructions of IRS_GEMM_X6.  Error model (round 5): a value's two planes
 * carry it to 2^-22 RELATIVE while l is a normal float16 (|x| >= 2^-3) and to 2^-25 ABSOLUTE below that (l subnormal).  The weight
 * operands -- O(0.05) in a trained model -- are therefore packed times 2^8 (exact; the epilogues fold the 2^-8 into their bias
 * multiply-add), which puts their absolute floor at 2^-33; the activation operands are O(1) LayerNorm outputs, hidden units and
 * attention outputs, where 2^-25 absolute is 2^-25 of the sum they enter.  Needs 2^8 |weights|, |activations| < 65504 (float16):
 * checked at finalisation, see irs_h3_range_bound. */
#define IRS_GEMM_H3 2
int irs_set_decoder_gemm(irs_ctx *ctx, int32_t mode);
int irs_get_decoder_gemm(const irs_ctx *ctx);           /* the selected mode (a get / set round trip restores it) */
int irs_get_decoder_gemm_effective(const irs_ctx *ctx); /* the mode that runs (IRS_GEMM_X6 where IRS_GEMM_H3 fails its range bound) */
/* Sequence-resident decoder layers (round 5; reference model/influentialRS.py:183-193 is ONE nn.TransformerDecoder call): with
 * IRS_GEMM_H3 at d = 128, 4 heads, ffn 256, L <= 256, rows-only decodes of a throughput batch run every layer but the last as ONE
 * launch -- q | k | v from x, K / V of a head in LDS, attention, out-projection, feed-forward, layer norms -- on whole sequences
 * per workgroup (k_block_x6<.., SEQ>), instead of a layer kernel + an attention kernel exchanging q | k | v rows through HBM.
 * The launch covers layers 0 .. n - 2 (x stays in registers from layer to layer) and the q | k | v + attention of the last layer
 * for the consumed token's block: no q | k | v row ever reaches HBM.  GEMM arithmetic as in the default kernels; attention scores as
 * three float16 plane products (the default kernels: float32 MFMAs): rows within 2e-5 of theirs, as close to the float32 kernels.
 * mode: 0 never, 1 whenever the shape allows, 2 (default) from 384 sequences per call up, where it measured 8-12 % ahead of the
 * two-kernel path.  Environment IRS_DECODER_SEQ=0 / 1 / auto at creation or this call.
 * irs_decoder_seq_last: 1 when the last irs_decode took this path. */
int irs_set_decoder_seq(irs_ctx *ctx, int32_t mode);
int irs_get_decoder_seq(const irs_ctx *ctx);
int irs_decoder_seq_last(const irs_ctx *ctx);
/* (tests / lab) device address of a decoder workspace buffer: 0 x (fragment-major), 1 attention output (fragment-major), 2 / 3 the
 * sequence-resident plan's tile -> sequence / tile index, 4 image row of a sequence, 5 tile-order consumed row, 6 workgroup count,
 * 7 / 8 packed offset / count per sequence, 9 q | k | v rows, 10 packed consumed row.  Null for an unknown index. */
void *irs_debug_ptr(const irs_ctx *ctx, int32_t which);
/* Float16 planes overflow at 65504.  irs_finalize_weights bounds every operand of the float16-plane kernels from the bound
 * weights (embedded tokens, LayerNorm outputs, hidden activations, V rows, the weights themselves) and keeps half the range as
 * margin: a model whose bound is 32752 or more runs IRS_GEMM_X6 (no range limit) wherever IRS_GEMM_H3 is selected, with float32
 * V rows -- irs_get_decoder_gemm_effective reports the mode that RUNS (irs_get_decoder_gemm the selected one).  irs_h3_range_bound returns the bound (-1 before finalisation).
 * (The reference has no counterpart: model/influentialRS.py:67-74 multiplies in float32 throughout.) */
float irs_h3_range_bound(const irs_ctx *ctx);

/* ---- measurement hooks (bench.py only) ---------------------------------
 * While enabled, every launch of the named kernel family is bracketed by HIP
 * events on the launch stream; irs_prof_read() synchronises those events and
 * returns launches and total milliseconds since the last reset. */
#define IRS_PROF_NONE 0
#define IRS_PROF_LINEAR 1  /* decoder GEMM family (fused layer kernel, layer-0 QKV) */
#define IRS_PROF_ATTN 2    /* decoder attention */
#define IRS_PROF_SWEEP 3   /* catalog sweep (pre-pass + emit) */
#define IRS_PROF_REFINE 4  /* candidate refine / exact re-score / sort */
#define IRS_PROF_SWEEP_EMIT 5 /* the emission sweep of irs_score_topk alone (the kernel the bf16 MFMA roofline is quoted on) */
#define IRS_PROF_LAYER 6   /* the fused layer kernel with the next layer's q | k | v (k_block_x6 / k_block) alone: the
                            * launches the decoder roofline is quoted on; flops are the algorithmic (float32-product) ones */
int irs_prof_enable(irs_ctx *ctx, int32_t family);
int irs_prof_read(irs_ctx *ctx, int32_t *launches, double *total_ms, double *total_flops, double *total_bytes);

#ifdef __cplusplus
}
#endif
#endif /* IRS_HIP_H */
