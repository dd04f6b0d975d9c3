// C ABI of libirs_hip.so (include/irs_hip.h): context, weight binding, workspace
// planning, entry points, hipGraph-captured path generation, measurement hooks.
#include <stdlib.h>

#include <new>

#include "irs_internal.h"

int irs_launch_topk_exhaustive(irs_ctx *ctx, const float *xrows, int M, int k, float *val, int64_t *ids0,
                               int32_t *status, hipStream_t s);

static char g_create_err[512] = "";

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

extern "C" int irs_abi_version(void) { return IRS_ABI_VERSION; }

extern "C" const char *irs_last_error(const irs_ctx *ctx) { return ctx ? ctx->err : g_create_err; }

extern "C" int irs_create(irs_ctx **out, const irs_dims *dims, const irs_shard *shard) {
    if (!out || !dims) {
        snprintf(g_create_err, sizeof(g_create_err), "irs_create: null argument");
        return IRS_E_INVALID;
    }
    const irs_dims &D = *dims;
#define BAD(...)                                                   \
    do {                                                           \
        snprintf(g_create_err, sizeof(g_create_err), __VA_ARGS__); \
        return IRS_E_INVALID;                                      \
    } while (0)
    if (D.n_item < 1) BAD("n_item must be >= 1");
    if (D.d < 2 || D.d > 256 || (D.d & 1)) BAD("emb_dim must be even and in [2, 256] (got %d)", D.d);
    if (D.max_len < 2 || D.max_len > 256) BAD("max_len must be in [2, 256] (got %d)", D.max_len);
    if (D.n_heads < 1 || D.d % D.n_heads) BAD("n_heads must divide emb_dim");
    if (D.d / D.n_heads > 64) BAD("head dim > 64 unsupported");
    if (D.n_layers < 1 || D.n_layers > IRS_MAX_LAYERS) BAD("n_layers out of range");
    if (D.ffn_dim < 1) BAD("ffn_dim must be >= 1");
    if (D.mask_mode != IRS_MASK_IRN && D.mask_mode != IRS_MASK_CAUSAL) BAD("bad mask_mode");
    if (D.mask_mode == IRS_MASK_IRN && (D.u_dim < 1 || D.n_user < 1)) BAD("IRN mask needs user embeddings");
    if (D.max_rows < 1) BAD("max_rows must be >= 1");
    if (D.max_k < 1 || D.max_k > 1024) BAD("max_k must be in [1, 1024]");
    irs_shard sh;
    if (shard) sh = *shard;
    else {
        sh.rank = 0;
        sh.world = 1;
        sh.item_lo = 0;
        sh.item_hi = D.n_item;
    }
    if (sh.item_lo < 0 || sh.item_hi > D.n_item || sh.item_lo >= sh.item_hi) BAD("bad item shard [%lld, %lld)", (long long)sh.item_lo, (long long)sh.item_hi);
    if (sh.item_hi - sh.item_lo > 0x7FFFFFE0LL) BAD("shard too large");
    {   // the launchers' per-device caches (MaxDynamicSharedMemorySize attributes, occupancy) are keyed by device ordinal
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && (dev < 0 || dev >= IRS_MAX_DEVICES))
            BAD("device ordinal %d: contexts are supported on devices 0 .. %d", dev, IRS_MAX_DEVICES - 1);
    }
#undef BAD
    irs_ctx *c = new (std::nothrow) irs_ctx();
    if (!c) {
        snprintf(g_create_err, sizeof(g_create_err), "out of host memory");
        return IRS_E_INVALID;
    }
    memset(c, 0, sizeof(*c));
    c->dims = D;
    c->shard = sh;
    c->n_local = sh.item_hi - sh.item_lo;
    int dp = 16;
    while (dp < D.d) dp <<= 1;
    c->d_pad = dp;
    c->KS = dp / 16;
    c->n_tiles = (int)((c->n_local + 31) / 32);
    c->max_rows = D.max_rows;
    c->max_seqs = D.max_seqs > 0 ? D.max_seqs : D.max_rows;
    c->m_pad_max = (c->max_rows + 31) & ~31;
    c->lse_slots = 2048;
    {   // decoder GEMMs of the throughput path: split-bf16 MFMAs (k_block_x6, the default) or float32 MFMAs (k_block);
        // the environment variable sets the initial mode, irs_set_decoder_gemm() changes it on a live context
        // (an unrecognised value is an error, not a silent default: a typo would otherwise select another arithmetic)
        const char *e = getenv("IRS_DECODER_GEMM");
        if (e && strcmp(e, "f32") && strcmp(e, "x6") && strcmp(e, "h3")) {
            snprintf(g_create_err, sizeof(g_create_err), "IRS_DECODER_GEMM=%s (expected h3, x6 or f32)", e);
            delete c;
            return IRS_E_INVALID;
        }
        c->use_x6 = e ? (strcmp(e, "f32") == 0 ? IRS_GEMM_F32 : strcmp(e, "x6") == 0 ? IRS_GEMM_X6 : IRS_GEMM_H3) : IRS_GEMM_H3;
        const char *ea = getenv("IRS_ATTN_GEMM");
        if (ea && strcmp(ea, "f32") && strcmp(ea, "h3")) {
            snprintf(g_create_err, sizeof(g_create_err), "IRS_ATTN_GEMM=%s (expected h3 or f32)", ea);
            delete c;
            return IRS_E_INVALID;
        }
        c->use_attn_h3 = ea ? (strcmp(ea, "h3") == 0) : 1; // (IRS_ATTN_GEMM=f32: float32 K / V rows and the float32-MFMA attention)
        const char *es = getenv("IRS_DECODER_SEQ");
        c->use_seq = es ? (strcmp(es, "auto") == 0 ? 2 : (atoi(es) != 0 ? 1 : 0)) : 2;
        const char *ec = getenv("IRS_THR_CARRY"); // steps between two pre-pass + threshold selections inside a path search (0 / 1: every step)
        c->carry_period = ec ? atoi(ec) : 8;
        if (c->carry_period < 0 || c->carry_period > 1024) c->carry_period = 8;
        const char *eo = getenv("IRS_SHARDED_OVERLAP");
        c->sh_overlap = eo ? (atoi(eo) != 0 ? 1 : 0) : 0; // (irs_set_sharded_overlap: off by default)
        const char *er = getenv("IRS_LSE_RING");
        c->lse_no_ring = er ? (strcmp(er, "0") == 0) : 0;
    }
    *out = c;
    return IRS_OK;
}

extern "C" void irs_destroy(irs_ctx *ctx) {
    if (!ctx) return;
    if (ctx->graph_exec) hipGraphExecDestroy(ctx->graph_exec);
    if (ctx->beam_graph) hipGraphExecDestroy(ctx->beam_graph);
    if (ctx->sh_graph) hipGraphExecDestroy(ctx->sh_graph);
    if (ctx->sh_side) {
        for (int i = 0; i < 8; ++i) (void)hipEventDestroy(ctx->sh_ev[i]);
        (void)hipStreamDestroy(ctx->sh_side);
    }
    if (ctx->prof_ev) {
        for (int i = 0; i < ctx->prof_cap; ++i) {
            hipEventDestroy(ctx->prof_ev[i].a);
            hipEventDestroy(ctx->prof_ev[i].b);
        }
        free(ctx->prof_ev);
    }
    delete ctx;
}

// ------------------------------------------------------------------ weights
static bool ends_with(const char *s, const char *suf) {
    size_t ls = strlen(s), lf = strlen(suf);
    return ls >= lf && strcmp(s + ls - lf, suf) == 0;
}

extern "C" int irs_bind_weight(irs_ctx *ctx, const char *name, const float *p, int64_t numel) {
    if (!ctx || !name || !p) return IRS_E_INVALID;
    if (strncmp(name, "module.", 7) == 0) name += 7;
    const irs_dims &D = ctx->dims;
    const int64_t d = D.d, F = D.ffn_dim;
    const float **slot = nullptr;
    int64_t want = -1;
    if (!strcmp(name, "item_embedder.weight") || !strcmp(name, "word_embedder.weight")) {
        slot = &ctx->item_emb;
        want = (D.n_item + 1) * d;
    } else if (!strcmp(name, "user_embedder.weight")) {
        slot = &ctx->user_emb;
        want = D.n_user * D.u_dim;
    } else if (!strcmp(name, "pos_embedder.pe")) {
        slot = &ctx->pe;
        want = -2; // >= max_len * d
        if (numel < (int64_t)D.max_len * d) IRS_FAIL(ctx, IRS_E_INVALID, "pos_embedder.pe too small");
    } else if (!strcmp(name, "user_mask_layer.weight")) {
        slot = &ctx->um_w;
        want = D.u_dim;
    } else if (!strcmp(name, "user_mask_layer.bias")) {
        slot = &ctx->um_b;
        want = 1;
    } else if (!strcmp(name, "project.weight")) {
        slot = &ctx->proj_w;
        want = ctx->n_local * d;
    } else if (!strcmp(name, "project.bias")) {
        slot = &ctx->proj_b;
        want = ctx->n_local;
    } else if (!strncmp(name, "decoder.layers.", 15)) {
        char *end = nullptr;
        long l = strtol(name + 15, &end, 10);
        if (l < 0 || l >= D.n_layers || !end || *end != '.') IRS_FAIL(ctx, IRS_E_INVALID, "bad layer in '%s'", name);
        const char *rest = end + 1;
        irs_layer_w &w = ctx->layer[l];
        struct { const char *n; const float **s; int64_t numel; } tab[] = {
            {"self_attn.in_proj_weight", &w.sa_in_w, 3 * d * d}, {"self_attn.in_proj_bias", &w.sa_in_b, 3 * d},
            {"self_attn.out_proj.weight", &w.sa_out_w, d * d},   {"self_attn.out_proj.bias", &w.sa_out_b, d},
            {"multihead_attn.in_proj_weight", &w.ca_in_w, 3 * d * d}, {"multihead_attn.in_proj_bias", &w.ca_in_b, 3 * d},
            {"multihead_attn.out_proj.weight", &w.ca_out_w, d * d},   {"multihead_attn.out_proj.bias", &w.ca_out_b, d},
            {"linear1.weight", &w.l1_w, F * d}, {"linear1.bias", &w.l1_b, F},
            {"linear2.weight", &w.l2_w, d * F}, {"linear2.bias", &w.l2_b, d},
            {"norm1.weight", &w.n1_w, d}, {"norm1.bias", &w.n1_b, d},
            {"norm2.weight", &w.n2_w, d}, {"norm2.bias", &w.n2_b, d},
            {"norm3.weight", &w.n3_w, d}, {"norm3.bias", &w.n3_b, d},
        };
        for (auto &e : tab)
            if (!strcmp(rest, e.n)) {
                slot = e.s;
                want = e.numel;
            }
    }
    if (!slot) IRS_FAIL(ctx, IRS_E_INVALID, "unknown weight '%s'", name);
    if (want >= 0 && numel != want) IRS_FAIL(ctx, IRS_E_INVALID, "weight '%s': numel %lld, expected %lld", name, (long long)numel, (long long)want);
    *slot = p;
    ctx->finalized = false;
    (void)ends_with;
    return IRS_OK;
}

static int check_bound(irs_ctx *ctx) {
    const irs_dims &D = ctx->dims;
    if (!ctx->item_emb || !ctx->pe || !ctx->proj_w || !ctx->proj_b) IRS_FAIL(ctx, IRS_E_STATE, "embedding / pe / project weights not bound");
    if (D.mask_mode == IRS_MASK_IRN && (!ctx->user_emb || !ctx->um_w || !ctx->um_b)) IRS_FAIL(ctx, IRS_E_STATE, "user weights not bound");
    for (int l = 0; l < D.n_layers; ++l) {
        const irs_layer_w &w = ctx->layer[l];
        const float *all[] = {w.sa_in_w, w.sa_in_b, w.sa_out_w, w.sa_out_b, w.ca_in_b, w.ca_out_w, w.ca_out_b, w.l1_w,
                              w.l1_b, w.l2_w, w.l2_b, w.n1_w, w.n1_b, w.n2_w, w.n2_b, w.n3_w, w.n3_b};
        for (auto p : all)
            if (!p) IRS_FAIL(ctx, IRS_E_STATE, "decoder layer %d has unbound weights", l);
    }
    return IRS_OK;
}

static size_t derived_plan(const irs_ctx *ctx, size_t *o_wp, size_t *o_bias, size_t *o_cl, size_t *o_wn, size_t *o_wf, size_t *o_x6 = nullptr) {
    size_t off = 0;
    *o_wp = off;
    off = align_up(off + (size_t)ctx->n_tiles * ctx->KS * 1024, 256);
    *o_bias = off;
    off = align_up(off + (size_t)ctx->n_tiles * 32 * sizeof(float), 256);
    *o_cl = off;
    off = align_up(off + (size_t)ctx->dims.n_layers * ctx->dims.d * sizeof(float), 256);
    *o_wn = off;
    off = align_up(off + 256, 256);
    *o_wf = off;
    off = align_up(off + irs_small_frag_floats(ctx) * sizeof(float), 256);
    if (o_x6) *o_x6 = off;
    off = align_up(off + irs_x6_bytes(ctx), 256);
    return off;
}

extern "C" size_t irs_derived_bytes(const irs_ctx *ctx) {
    size_t a, b, c, d, e;
    return ctx ? derived_plan(ctx, &a, &b, &c, &d, &e) : 0;
}

extern "C" int irs_finalize_weights(irs_ctx *ctx, void *arena, size_t bytes, void *stream) {
    if (!ctx || !arena) return IRS_E_INVALID;
    int rc = check_bound(ctx);
    if (rc) return rc;
    size_t o_wp, o_bias, o_cl, o_wn, o_wf, o_x6;
    size_t need = derived_plan(ctx, &o_wp, &o_bias, &o_cl, &o_wn, &o_wf, &o_x6);
    if (bytes < need) IRS_FAIL(ctx, IRS_E_INVALID, "derived arena too small: %zu < %zu", bytes, need);
    if (((uintptr_t)arena) & 255) IRS_FAIL(ctx, IRS_E_INVALID, "derived arena must be 256-byte aligned");
    char *base = (char *)arena;
    ctx->wp = (uint4 *)(base + o_wp);
    ctx->bias_pad = (float *)(base + o_bias);
    ctx->c_l = (float *)(base + o_cl);
    ctx->wnorm_max = (float *)(base + o_wn);
    ctx->w_frag16 = irs_small_frag_floats(ctx) ? (float *)(base + o_wf) : nullptr;
    ctx->w_x6 = irs_x6_bytes(ctx) ? (uint4 *)(base + o_x6) : nullptr;
    hipStream_t s = (hipStream_t)stream;
    if ((rc = irs_launch_pack_w(ctx, s))) return rc;
    if ((rc = irs_launch_cross_const(ctx, s))) return rc;
    if ((rc = irs_launch_pack_small(ctx, s))) return rc;
    if ((rc = irs_launch_pack_x6(ctx, s))) return rc;
    // float16 planes (IRS_GEMM_H3, the V planes of the attention) need every operand below 65504: bound them from the weights,
    // keep half the range as margin; a model outside it runs the split-bf16 kernels (no range limit) -- the one host
    // synchronisation of finalisation
    ctx->h3_ok = true, ctx->h3_bound = 0.f;
    if (ctx->w_x6) {
        float *st = ctx->wnorm_max + 8, hst[8];
        if ((rc = irs_launch_h3_range(ctx, st, s))) return rc;
        IRS_CHECK_HIP(ctx, hipMemcpyAsync(hst, st, sizeof(hst), hipMemcpyDeviceToHost, s));
        IRS_CHECK_HIP(ctx, hipStreamSynchronize(s));
        ctx->h3_bound = irs_h3_operand_bound(ctx, hst);
        ctx->h3_ok = ctx->h3_bound < 32752.f; // (a NaN or infinite statistic fails the comparison)
    }
    ctx->finalized = true;
    ctx->proj_stale = false;
    ctx->thr_valid = 0; // (carried emission thresholds belong to the catalog they were selected on)
    if (ctx->sh_graph) {
        hipGraphExecDestroy(ctx->sh_graph);
        ctx->sh_graph = nullptr;
    }
    if (ctx->graph_exec) {
        hipGraphExecDestroy(ctx->graph_exec);
        ctx->graph_exec = nullptr;
    }
    if (ctx->beam_graph) {
        hipGraphExecDestroy(ctx->beam_graph);
        ctx->beam_graph = nullptr;
    }
    return IRS_OK;
}

// ------------------------------------------------------------------ workspace
struct ws_plan {
    size_t x, y, xf, yf, qkv, qkv_b1, ao, h, ru, xb, eps, thr, gm, cnt, cand, lse, ref, xrows, tval, tids, status, step, pos;
    size_t bseq[2], bhep[2], bcum[2], bpaths[2], buser, lmax, lsum, tokrow, scnt, soff, sqrow, spadq, mdev, tseq, tidx, srow0, qtile, nwg, xlocal, ksend, krecv, gmax, fbcount, fblist, exhkeys, total;
};

static void workspace_plan(const irs_ctx *ctx, ws_plan *p) {
    const irs_dims &D = ctx->dims;
    const size_t RL = (size_t)ctx->max_seqs * D.max_len;
    const size_t mp = ctx->m_pad_max;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off = align_up(off + bytes, 256);
        return o;
    };
    p->x = take(RL * D.d * 4);
    p->y = take(RL * D.d * 4);
    // (the sequence-resident layer kernel pads every sequence to whole 32-token tiles and every workgroup to 8 tiles: up to 256 rows
    //  per sequence in the fragment-major images)
    const bool seq_shape = D.d == 128 && D.ffn_dim == 256 && D.n_heads == 4 && D.max_len <= 256 && D.n_layers > 1;
    const size_t RLs = seq_shape ? (size_t)ctx->max_seqs * 256 : RL;
    const size_t RLf = (((RLs > RL ? RLs : RL) + 127) / 128) * 128; // 128-token tiles x 128 padded columns
    const size_t fcols = D.d > 128 ? (size_t)((D.d + 31) / 32) * 32 : 128; // (d = 256: eight column tiles per token tile)
    p->xf = take(RLf * fcols * 4);
    p->yf = take(RLf * fcols * 4);
    p->qkv = take(RL * 3 * D.d * 4);
    p->qkv_b1 = take((RL < 256 ? RL : 256) * 3 * D.d * 4);
    p->ao = take(RL * D.d * 4);
    p->h = take(RL * D.ffn_dim * 4);
    p->ru = take((size_t)ctx->max_seqs * 4);
    p->xb = take(mp * ctx->d_pad * 2);
    p->eps = take(mp * 4);
    p->thr = take(mp * 4);
    p->gm = take((size_t)IRS_MAX_GROUPS * mp * 4);
    p->cnt = take(mp * (size_t)IRS_CAND_BUCKETS * 4);
    p->cand = take(mp * (size_t)IRS_CAND_CAP * 8);
    {
        size_t lse_b = (size_t)ctx->lse_slots * mp * 8, ring_b = (size_t)IRS_LSE_SLOTS_RING * 32 * 8; // (the ring form: <= 32 rows, more slots)
        p->lse = take(lse_b > ring_b ? lse_b : ring_b);
    }
    p->ref = take(mp * 4);
    p->xrows = take((size_t)ctx->max_rows * D.d * 4);
    p->tval = take((size_t)ctx->max_rows * D.max_k * 4);
    p->tids = take((size_t)ctx->max_rows * D.max_k * 8);
    p->status = take((size_t)ctx->max_rows * 4);
    p->step = take(256);
    p->pos = take((size_t)ctx->max_seqs * 4);
    for (int i = 0; i < 2; ++i) {
        p->bseq[i] = take((size_t)ctx->max_seqs * D.max_len * 8);
        p->bhep[i] = take((size_t)ctx->max_seqs * 4);
        p->bcum[i] = take((size_t)ctx->max_seqs * 8);
        p->bpaths[i] = take((size_t)ctx->max_seqs * IRS_MAX_PATH * 4);
    }
    p->buser = take((size_t)ctx->max_seqs * 8);
    p->lmax = take((size_t)ctx->max_rows * 4);
    p->lsum = take((size_t)ctx->max_rows * 4);
    p->tokrow = take(RL * 4);
    p->scnt = take((size_t)ctx->max_seqs * 4);
    p->soff = take((size_t)ctx->max_seqs * 4);
    p->sqrow = take((size_t)ctx->max_seqs * 4);
    p->spadq = take((size_t)ctx->max_seqs * 4);
    p->mdev = take(256);
    p->tseq = seq_shape ? take((size_t)ctx->max_seqs * 16 * 4) : 0;
    p->tidx = seq_shape ? take((size_t)ctx->max_seqs * 16 * 4) : 0;
    p->srow0 = seq_shape ? take((size_t)ctx->max_seqs * 4) : 0;
    p->qtile = seq_shape ? take((size_t)ctx->max_seqs * 4) : 0;
    p->nwg = seq_shape ? take(256) : 0;
    p->xlocal = take((size_t)ctx->max_seqs * D.d * 4);
    p->ksend = take((size_t)ctx->max_rows * D.max_k * 8); // exchange buffers of the item-sharded loops (comm.hip)
    p->krecv = take((size_t)ctx->max_rows * D.max_k * 8);
    p->gmax = take((size_t)ctx->max_rows * 4);
    p->fbcount = take(256);
    p->fblist = take((size_t)ctx->max_rows * 4);
    p->exhkeys = ctx->n_local >= IRS_COOP_FALLBACK_MIN_ITEMS ? take((size_t)IRS_EXH_SCRATCH_KEYS * 8) : 0;
    p->total = off;
}

extern "C" size_t irs_workspace_bytes(const irs_ctx *ctx) {
    if (!ctx) return 0;
    ws_plan p;
    workspace_plan(ctx, &p);
    return p.total;
}

extern "C" int irs_bind_workspace(irs_ctx *ctx, void *ws, size_t bytes) {
    if (!ctx || !ws) return IRS_E_INVALID;
    ws_plan p;
    workspace_plan(ctx, &p);
    if (bytes < p.total) IRS_FAIL(ctx, IRS_E_INVALID, "workspace too small: %zu < %zu", bytes, p.total);
    if (((uintptr_t)ws) & 255) IRS_FAIL(ctx, IRS_E_INVALID, "workspace must be 256-byte aligned");
    char *b = (char *)ws;
    ctx->ws = b;
    ctx->ws_bytes = bytes;
    ctx->act_x = (float *)(b + p.x);
    ctx->act_y = (float *)(b + p.y);
    ctx->act_xf = (float *)(b + p.xf);
    ctx->act_yf = (float *)(b + p.yf);
    ctx->act_qkv = (float *)(b + p.qkv);
    ctx->act_qkv_b1 = (float *)(b + p.qkv_b1);
    ctx->act_ao = (float *)(b + p.ao);
    ctx->act_h = (float *)(b + p.h);
    ctx->act_ru = (float *)(b + p.ru);
    ctx->xb = (uint4 *)(b + p.xb);
    ctx->eps = (float *)(b + p.eps);
    ctx->thr = (float *)(b + p.thr);
    ctx->thr_valid = 0;
    ctx->gm = (float *)(b + p.gm);
    ctx->cand_cnt = (unsigned int *)(b + p.cnt);
    ctx->cand = (unsigned long long *)(b + p.cand);
    ctx->lse_part = (float *)(b + p.lse);
    ctx->ref_tmp = (float *)(b + p.ref);
    ctx->xrows = (float *)(b + p.xrows);
    ctx->top_val = (float *)(b + p.tval);
    ctx->top_ids = (int64_t *)(b + p.tids);
    ctx->row_status = (int32_t *)(b + p.status);
    ctx->step_ctr = (int32_t *)(b + p.step); // [0..1] step pair of the path loops, [8..40) arrival counters of k_topk_direct
    IRS_CHECK_HIP(ctx, hipMemset(ctx->step_ctr, 0, 256));
    ctx->pos_tmp = (int32_t *)(b + p.pos);
    for (int i = 0; i < 2; ++i) {
        ctx->bm_seq[i] = (int64_t *)(b + p.bseq[i]);
        ctx->bm_hep[i] = (int32_t *)(b + p.bhep[i]);
        ctx->bm_cum[i] = (double *)(b + p.bcum[i]);
        ctx->bm_paths[i] = (float *)(b + p.bpaths[i]);
    }
    ctx->bm_user = (int64_t *)(b + p.buser);
    ctx->lse_max = (float *)(b + p.lmax);
    ctx->lse_sum = (float *)(b + p.lsum);
    ctx->tok_row = (int32_t *)(b + p.tokrow);
    ctx->seq_cnt = (int32_t *)(b + p.scnt);
    ctx->seq_off = (int32_t *)(b + p.soff);
    ctx->seq_qrow = (int32_t *)(b + p.sqrow);
    ctx->seq_padq = (int32_t *)(b + p.spadq);
    ctx->m_dev = (int32_t *)(b + p.mdev);
    {
        const irs_dims &D_ = ctx->dims;
        const bool seq_shape = D_.d == 128 && D_.ffn_dim == 256 && D_.n_heads == 4 && D_.max_len <= 256 && D_.n_layers > 1;
        ctx->tile_seq = seq_shape ? (int32_t *)(b + p.tseq) : nullptr;
        ctx->tile_idx = seq_shape ? (int32_t *)(b + p.tidx) : nullptr;
        ctx->seq_row0 = seq_shape ? (int32_t *)(b + p.srow0) : nullptr;
        ctx->qrow_tile = seq_shape ? (int32_t *)(b + p.qtile) : nullptr;
        ctx->n_wg_dev = seq_shape ? (int32_t *)(b + p.nwg) : nullptr;
    }
    ctx->x_local = (float *)(b + p.xlocal);
    ctx->keys_send = (uint64_t *)(b + p.ksend);
    ctx->keys_recv = (uint64_t *)(b + p.krecv);
    ctx->lse_gmax = (float *)(b + p.gmax);
    ctx->fb_count = (unsigned int *)(b + p.fbcount);
    ctx->fb_list = (int32_t *)(b + p.fblist);
    ctx->exh_keys = ctx->n_local >= IRS_COOP_FALLBACK_MIN_ITEMS ? (unsigned long long *)(b + p.exhkeys) : nullptr;
    if (ctx->sh_graph) {
        hipGraphExecDestroy(ctx->sh_graph);
        ctx->sh_graph = nullptr;
    }
    if (ctx->beam_graph) {
        hipGraphExecDestroy(ctx->beam_graph);
        ctx->beam_graph = nullptr;
    }
    if (ctx->graph_exec) {
        hipGraphExecDestroy(ctx->graph_exec);
        ctx->graph_exec = nullptr;
    }
    return IRS_OK;
}

static void drop_graphs(irs_ctx *ctx) { // captured steps hold kernel choices and buffer addresses
    if (ctx->sh_graph) {
        hipGraphExecDestroy(ctx->sh_graph);
        ctx->sh_graph = nullptr;
    }
    if (ctx->beam_graph) {
        hipGraphExecDestroy(ctx->beam_graph);
        ctx->beam_graph = nullptr;
    }
    if (ctx->graph_exec) {
        hipGraphExecDestroy(ctx->graph_exec);
        ctx->graph_exec = nullptr;
    }
}

extern "C" int irs_set_decoder_gemm(irs_ctx *ctx, int32_t mode) {
    if (!ctx) return IRS_E_INVALID;
    if (mode != IRS_GEMM_F32 && mode != IRS_GEMM_X6 && mode != IRS_GEMM_H3)
        IRS_FAIL(ctx, IRS_E_INVALID, "decoder GEMM mode %d (IRS_GEMM_F32, IRS_GEMM_X6 or IRS_GEMM_H3)", mode);
    if (ctx->use_x6 != mode) {
        ctx->use_x6 = mode;
        drop_graphs(ctx);
    }
    return IRS_OK;
}

extern "C" int irs_set_decoder_seq(irs_ctx *ctx, int32_t mode) {
    if (!ctx) return IRS_E_INVALID;
    if (mode < 0 || mode > 2) IRS_FAIL(ctx, IRS_E_INVALID, "decoder seq mode %d (0 off, 1 on, 2 auto)", mode);
    if (ctx->use_seq != mode) {
        ctx->use_seq = mode;
        drop_graphs(ctx);
    }
    return IRS_OK;
}
extern "C" int irs_get_decoder_seq(const irs_ctx *ctx) { return ctx ? ctx->use_seq : IRS_E_INVALID; }
extern "C" int irs_decoder_seq_last(const irs_ctx *ctx) { return ctx ? (ctx->seq_last ? 1 : 0) : IRS_E_INVALID; }
// (lab / tests: device addresses of decoder workspace buffers, so that a test can look at what a decode left behind)
extern "C" void *irs_debug_ptr(const irs_ctx *ctx, int32_t which) {
    if (!ctx) return nullptr;
    switch (which) {
    case 0: return ctx->act_xf;
    case 1: return ctx->act_yf;
    case 2: return ctx->tile_seq;
    case 3: return ctx->tile_idx;
    case 4: return ctx->seq_row0;
    case 5: return ctx->qrow_tile;
    case 6: return ctx->n_wg_dev;
    case 7: return ctx->seq_off;
    case 8: return ctx->seq_cnt;
    case 9: return ctx->act_qkv;
    case 10: return ctx->seq_qrow;
    default: return nullptr;
    }
}

// the SELECTED mode (what irs_set_decoder_gemm stored: a get / set round trip restores it) ...
extern "C" int irs_get_decoder_gemm(const irs_ctx *ctx) {
    if (!ctx) return IRS_E_INVALID;
    return ctx->use_x6;
}
// ... and the mode that RUNS: a model whose weights fail finalisation's float16 range bound runs IRS_GEMM_X6 where
// IRS_GEMM_H3 is selected
extern "C" int irs_get_decoder_gemm_effective(const irs_ctx *ctx) {
    if (!ctx) return IRS_E_INVALID;
    return (ctx->use_x6 == IRS_GEMM_H3 && ctx->finalized && !ctx->h3_ok) ? IRS_GEMM_X6 : ctx->use_x6;
}
extern "C" float irs_h3_range_bound(const irs_ctx *ctx) { return ctx && ctx->finalized ? ctx->h3_bound : -1.f; }

static int ready(irs_ctx *ctx) {
    if (!ctx) return IRS_E_INVALID;
    if (!ctx->finalized) IRS_FAIL(ctx, IRS_E_STATE, "weights not finalized (irs_finalize_weights)");
    if (!ctx->ws) IRS_FAIL(ctx, IRS_E_STATE, "workspace not bound (irs_bind_workspace)");
    return IRS_OK;
}

// entry points that filter through the bf16 catalog copy and its norms: a training entry point since the last
// irs_finalize_weights means project.* may have moved under them (the filter's |approx - exact| <= eps would not hold)
static int ready_filter(irs_ctx *ctx, int sweep) {
    int rc = ready(ctx);
    if (rc) return rc;
    if (ctx->proj_stale && sweep == IRS_SWEEP_BF16)
        IRS_FAIL(ctx, IRS_E_STATE, "project.* may have changed since irs_finalize_weights (irs_ce_forward / irs_ce_grad_logits "
                                   "ran): call irs_finalize_weights before filtering through the bf16 catalog");
    return IRS_OK;
}

// ------------------------------------------------------------------ entry points
extern "C" int irs_pif(irs_ctx *ctx, const int64_t *user, int32_t B, float *r_u, void *stream) {
    int rc = ready(ctx);
    if (rc) return rc;
    if (B < 1 || !r_u) IRS_FAIL(ctx, IRS_E_INVALID, "irs_pif: bad arguments");
    if (ctx->dims.mask_mode == IRS_MASK_IRN && !user) IRS_FAIL(ctx, IRS_E_INVALID, "irs_pif: user is null");
    return irs_launch_pif(ctx, user, B, r_u, (hipStream_t)stream);
}

extern "C" int irs_decode(irs_ctx *ctx, const int64_t *seq, const int64_t *user, int32_t B, float *x, const int32_t *pos,
                          float *xrows, float *r_u, void *stream) {
    int rc = ready(ctx);
    if (rc) return rc;
    if (!seq || B < 1) IRS_FAIL(ctx, IRS_E_INVALID, "irs_decode: bad arguments");
    if (B > ctx->max_seqs) IRS_FAIL(ctx, IRS_E_INVALID, "irs_decode: B=%d exceeds max_seqs=%d", B, ctx->max_seqs);
    if (ctx->dims.mask_mode == IRS_MASK_IRN && !user) IRS_FAIL(ctx, IRS_E_INVALID, "irs_decode: user is null");
    if ((pos == nullptr) != (xrows == nullptr)) IRS_FAIL(ctx, IRS_E_INVALID, "irs_decode: pos and xrows go together");
    return irs_launch_decode(ctx, seq, user, B, x, pos, xrows, r_u, (hipStream_t)stream);
}

static int check_rows(irs_ctx *ctx, const char *fn, const void *xrows, int M) {
    if (!xrows || M < 1) IRS_FAIL(ctx, IRS_E_INVALID, "%s: bad arguments", fn);
    if (M > ctx->max_rows) IRS_FAIL(ctx, IRS_E_INVALID, "%s: M=%d exceeds max_rows=%d", fn, M, ctx->max_rows);
    return IRS_OK;
}

extern "C" int irs_score_topk_carry(irs_ctx *ctx, const float *xrows, int32_t M, int32_t k, int32_t sweep, float *val,
                                    int64_t *ids0, int32_t *status, void *stream) {
    int rc = ready_filter(ctx, sweep);
    if (rc) return rc;
    if ((rc = check_rows(ctx, "irs_score_topk_carry", xrows, M))) return rc;
    if (k < 1 || k > ctx->dims.max_k || !val || !ids0 || !status) IRS_FAIL(ctx, IRS_E_INVALID, "irs_score_topk_carry: bad k / outputs");
    if (sweep != IRS_SWEEP_BF16) IRS_FAIL(ctx, IRS_E_INVALID, "irs_score_topk_carry: the bf16 filter only");
    return irs_launch_topk(ctx, xrows, M, k, sweep, val, ids0, status, (hipStream_t)stream, nullptr, nullptr, nullptr, 1);
}

extern "C" int irs_score_topk(irs_ctx *ctx, const float *xrows, int32_t M, int32_t k, int32_t sweep, float *val,
                              int64_t *ids0, int32_t *status, void *stream) {
    int rc = ready_filter(ctx, sweep);
    if (rc) return rc;
    if ((rc = check_rows(ctx, "irs_score_topk", xrows, M))) return rc;
    if (k < 1 || k > ctx->dims.max_k || !val || !ids0 || !status) IRS_FAIL(ctx, IRS_E_INVALID, "irs_score_topk: bad k / outputs");
    if (sweep == IRS_SWEEP_EXHAUSTIVE) return irs_launch_topk_exhaustive(ctx, xrows, M, k, val, ids0, status, (hipStream_t)stream);
    if (sweep != IRS_SWEEP_BF16 && sweep != IRS_SWEEP_F32) IRS_FAIL(ctx, IRS_E_INVALID, "irs_score_topk: bad sweep");
    return irs_launch_topk(ctx, xrows, M, k, sweep, val, ids0, status, (hipStream_t)stream);
}

extern "C" int irs_score_gather(irs_ctx *ctx, const float *xrows, int32_t M, const int64_t *ids0, int32_t g, float *out,
                                void *stream) {
    int rc = ready(ctx);
    if (rc) return rc;
    if ((rc = check_rows(ctx, "irs_score_gather", xrows, M))) return rc;
    if (!ids0 || g < 1 || !out) IRS_FAIL(ctx, IRS_E_INVALID, "irs_score_gather: bad arguments");
    return irs_launch_gather(ctx, xrows, M, ids0, g, out, (hipStream_t)stream);
}

extern "C" int irs_score_count_before(irs_ctx *ctx, const float *xrows, int32_t M, const float *ref_score,
                                      const int64_t *ref_id0, const int64_t *excl, int32_t n_excl, int64_t *count,
                                      void *stream) {
    int rc = ready(ctx);
    if (rc) return rc;
    if ((rc = check_rows(ctx, "irs_score_count_before", xrows, M))) return rc;
    if (!ref_score || !ref_id0 || !count || n_excl < 0) IRS_FAIL(ctx, IRS_E_INVALID, "irs_score_count_before: bad arguments");
    return irs_launch_count_before(ctx, xrows, M, ref_score, ref_id0, excl, n_excl, count, (hipStream_t)stream);
}

extern "C" int irs_score_dense(irs_ctx *ctx, const float *xrows, int32_t M, float *out, int64_t ld, void *stream) {
    int rc = ready(ctx);
    if (rc) return rc;
    if ((rc = check_rows(ctx, "irs_score_dense", xrows, M))) return rc;
    if (!out || ld < ctx->n_local) IRS_FAIL(ctx, IRS_E_INVALID, "irs_score_dense: bad out / ld");
    return irs_launch_dense(ctx, xrows, M, out, ld, (hipStream_t)stream);
}

extern "C" int irs_score_lse(irs_ctx *ctx, const float *xrows, int32_t M, float *omax, float *osum, void *stream) {
    int rc = ready(ctx);
    if (rc) return rc;
    if ((rc = check_rows(ctx, "irs_score_lse", xrows, M))) return rc;
    if (!omax || !osum) IRS_FAIL(ctx, IRS_E_INVALID, "irs_score_lse: null outputs");
    return irs_launch_lse(ctx, xrows, M, omax, osum, (hipStream_t)stream);
}

extern "C" int irs_score_topk_lse(irs_ctx *ctx, const float *xrows, int32_t M, int32_t k, int32_t sweep, float *val,
                                  int64_t *ids0, int32_t *status, float *omax, float *osum, void *stream) {
    int rc = ready_filter(ctx, sweep);
    if (rc) return rc;
    if ((rc = check_rows(ctx, "irs_score_topk_lse", xrows, M))) return rc;
    if (k < 1 || k > ctx->dims.max_k || !val || !ids0 || !status || !omax || !osum)
        IRS_FAIL(ctx, IRS_E_INVALID, "irs_score_topk_lse: bad k / outputs");
    if (sweep != IRS_SWEEP_BF16 && sweep != IRS_SWEEP_F32) IRS_FAIL(ctx, IRS_E_INVALID, "irs_score_topk_lse: bad sweep");
    return irs_launch_topk(ctx, xrows, M, k, sweep, val, ids0, status, (hipStream_t)stream, nullptr, omax, osum);
}

// ---- projection + cross entropy, training side
extern "C" int irs_ce_forward(irs_ctx *ctx, const float *xrows, const int64_t *labels0, int32_t M, float *lse,
                              float *label_score, double *loss, void *stream) {
    int rc = ready(ctx);
    if (rc) return rc;
    if ((rc = check_rows(ctx, "irs_ce_forward", xrows, M))) return rc;
    if (!labels0 || !lse || !label_score || !loss) IRS_FAIL(ctx, IRS_E_INVALID, "irs_ce_forward: null arguments");
    if (ctx->shard.world != 1) IRS_FAIL(ctx, IRS_E_UNSUPPORTED, "irs_ce_forward needs the whole catalog on one device");
    hipStream_t s = (hipStream_t)stream;
    ctx->proj_stale = true;
    if ((rc = irs_launch_refresh_bias(ctx, s))) return rc;
    if ((rc = irs_launch_lse(ctx, xrows, M, ctx->lse_max, ctx->lse_sum, s))) return rc;
    if ((rc = irs_launch_lse_combine(ctx, ctx->lse_max, ctx->lse_sum, lse, M, s))) return rc;
    if ((rc = irs_launch_gather(ctx, xrows, M, labels0, 1, label_score, s))) return rc;
    return irs_launch_ce_reduce(ctx, lse, label_score, labels0, M, loss, s);
}

extern "C" int irs_ce_grad_logits(irs_ctx *ctx, const float *xrows, const int64_t *labels0, const float *lse, int32_t M,
                                  float scale, float *out, int64_t ld, void *stream) {
    int rc = ready(ctx);
    if (rc) return rc;
    if ((rc = check_rows(ctx, "irs_ce_grad_logits", xrows, M))) return rc;
    if (!labels0 || !lse || !out || ld < ctx->n_local) IRS_FAIL(ctx, IRS_E_INVALID, "irs_ce_grad_logits: bad arguments");
    if (ctx->shard.world != 1) IRS_FAIL(ctx, IRS_E_UNSUPPORTED, "irs_ce_grad_logits needs the whole catalog on one device");
    hipStream_t s = (hipStream_t)stream;
    ctx->proj_stale = true;
    if ((rc = irs_launch_refresh_bias(ctx, s))) return rc;
    return irs_launch_ce_grad(ctx, xrows, labels0, lse, M, scale, out, ld, s);
}

extern "C" int irs_merge_topk(irs_ctx *ctx, const float *val_in, const int64_t *ids_in, int32_t W, int32_t M, int32_t k,
                              float *val, int64_t *ids0, void *stream) {
    if (!ctx) return IRS_E_INVALID;
    if (!val_in || !ids_in || !val || !ids0 || W < 1 || M < 1 || k < 1) IRS_FAIL(ctx, IRS_E_INVALID, "irs_merge_topk: bad arguments");
    return irs_launch_merge(ctx, val_in, ids_in, W, M, k, val, ids0, (hipStream_t)stream);
}

extern "C" int irs_pack_topk(irs_ctx *ctx, const float *val, const int64_t *ids0, int64_t n, uint64_t *keys, void *stream) {
    if (!ctx) return IRS_E_INVALID;
    if (!val || !ids0 || !keys || n < 1) IRS_FAIL(ctx, IRS_E_INVALID, "irs_pack_topk: bad arguments");
    return irs_launch_pack_topk(ctx, val, ids0, n, keys, (hipStream_t)stream);
}

extern "C" int irs_merge_topk_keys(irs_ctx *ctx, const uint64_t *keys_in, int32_t W, int32_t M, int32_t k, float *val,
                                   int64_t *ids0, void *stream) {
    if (!ctx) return IRS_E_INVALID;
    if (!keys_in || !val || !ids0 || W < 1 || M < 1 || k < 1) IRS_FAIL(ctx, IRS_E_INVALID, "irs_merge_topk_keys: bad arguments");
    return irs_launch_merge_keys(ctx, keys_in, W, M, k, val, ids0, (hipStream_t)stream);
}

extern "C" int irs_build_eval_batch(irs_ctx *ctx, const int64_t *items, const int64_t *offsets, int32_t B, int32_t raw_len,
                                    int32_t gap_len, const int64_t *targets_in, const int64_t *pool, int64_t n_pool,
                                    uint64_t seed, int64_t *seq, int64_t *target, int64_t *label, int64_t *raw,
                                    int32_t *raw_n, int32_t *status, void *stream) {
    if (!ctx) return IRS_E_INVALID;
    if (!items || !offsets || !seq || !target || !label || B < 1 || raw_len < 1 || gap_len < 0 || (pool && n_pool < 1))
        IRS_FAIL(ctx, IRS_E_INVALID, "irs_build_eval_batch: bad arguments");
    return irs_launch_build_eval_batch(ctx, items, offsets, B, raw_len, gap_len, targets_in, pool, n_pool, seed, seq, target,
                                       label, raw, raw_n, status, (hipStream_t)stream);
}

extern "C" int irs_path_step(irs_ctx *ctx, int64_t *seq, int32_t *hep, int32_t B, const float *val, const int64_t *ids0,
                             int32_t k, int32_t step, float *paths, int32_t path_ld, int32_t sample, int32_t sample_k,
                             uint64_t seed, int32_t *status, void *stream) {
    if (!ctx) return IRS_E_INVALID;
    if (!seq || !hep || !val || !ids0 || !paths || !status || B < 1 || k < 1 || step < 0 || step >= path_ld)
        IRS_FAIL(ctx, IRS_E_INVALID, "irs_path_step: bad arguments");
    if (sample && (sample_k < 1 || sample_k > IRS_MAX_SAMPLE_K))
        IRS_FAIL(ctx, IRS_E_UNSUPPORTED, "irs_path_step: sample_k must be in [1, %d]", IRS_MAX_SAMPLE_K);
    return irs_launch_path_step(ctx, seq, hep, B, val, ids0, k, step, nullptr, paths, path_ld, sample, sample_k, seed,
                                status, (hipStream_t)stream);
}

// one search step on one device: decode -> rows at hep -> top-k -> choose/update
static bool step_merged(const irs_ctx *ctx, int B) { // see irs_launch_decode: the single-workgroup plan kernel runs
    return B <= 64 && ctx->dims.max_len >= 4;
}

static int enqueue_step(irs_ctx *ctx, int64_t *seq, const int64_t *user, int32_t *hep, int B, int k, int sweep,
                        int sample, int sample_k, uint64_t seed, float *paths, int path_ld, int32_t *status,
                        hipStream_t s, int carry = 0) {
    int rc;
    const bool merged = step_merged(ctx, B);
    ctx->step_pair = merged ? ctx->step_ctr : nullptr;
    rc = irs_launch_decode(ctx, seq, user, B, nullptr, hep, ctx->xrows, nullptr, s);
    ctx->step_pair = nullptr;
    if (rc) return rc;
    // small shard, few rows: the workgroup that ranks a row's candidates also takes the row's path step
    if (merged && sweep != IRS_SWEEP_EXHAUSTIVE && irs_topk_is_direct(ctx, B, k)) {
        const irs_path_args pa{seq, hep, ctx->dims.max_len, paths, path_ld, sample, sample_k, (unsigned long long)seed, status,
                               ctx->step_ctr, 0, ctx->step_ctr + 1, 1};
        return irs_launch_topk(ctx, ctx->xrows, B, k, sweep, ctx->top_val, ctx->top_ids, ctx->row_status, s, &pa);
    }
    if ((rc = irs_launch_topk(ctx, ctx->xrows, B, k, sweep, ctx->top_val, ctx->top_ids, ctx->row_status, s, nullptr, nullptr, nullptr, carry)))
        return rc;
    if ((rc = irs_launch_path_step(ctx, seq, hep, B, ctx->top_val, ctx->top_ids, k, 0, ctx->step_ctr, paths, path_ld,
                                   sample, sample_k, seed, status, s, merged ? ctx->step_ctr + 1 : nullptr)))
        return rc;
    return merged ? IRS_OK : irs_launch_inc(ctx, ctx->step_ctr, s);
}

extern "C" int irs_generate_paths(irs_ctx *ctx, int64_t *seq, const int64_t *user, int32_t *hep, int32_t B,
                                  int32_t max_path_len, int32_t k, int32_t sweep, int32_t sample, int32_t sample_k,
                                  uint64_t seed, int32_t use_graph, float *paths, int32_t *status, void *stream) {
    int rc = ready_filter(ctx, sweep);
    if (rc) return rc;
    if (ctx->shard.world != 1) IRS_FAIL(ctx, IRS_E_UNSUPPORTED, "irs_generate_paths needs the whole catalog on one device");
    if (!seq || !hep || !paths || !status || B < 1 || max_path_len < 1) IRS_FAIL(ctx, IRS_E_INVALID, "irs_generate_paths: bad arguments");
    if (B > ctx->max_seqs || B > ctx->max_rows) IRS_FAIL(ctx, IRS_E_INVALID, "irs_generate_paths: B too large");
    if (k < 1 || k > ctx->dims.max_k) IRS_FAIL(ctx, IRS_E_INVALID, "irs_generate_paths: bad k");
    if (sweep != IRS_SWEEP_BF16 && sweep != IRS_SWEEP_F32) IRS_FAIL(ctx, IRS_E_INVALID, "irs_generate_paths: bad sweep");
    if (sample && (sample_k < 1 || sample_k > IRS_MAX_SAMPLE_K))
        IRS_FAIL(ctx, IRS_E_UNSUPPORTED, "irs_generate_paths: sample_k must be in [1, %d]", IRS_MAX_SAMPLE_K);
    hipStream_t s = (hipStream_t)stream;
    IRS_CHECK_HIP(ctx, hipMemsetAsync(ctx->step_ctr, 0, 2 * sizeof(int32_t), s));
    IRS_CHECK_HIP(ctx, hipMemsetAsync(status, 0, sizeof(int32_t) * B, s));
    if (!use_graph) {
        for (int i = 0; i < max_path_len; ++i) // (steps behind the first may reuse the previous step's emission thresholds)
            if ((rc = enqueue_step(ctx, seq, user, hep, B, k, sweep, sample, sample_k, seed, paths, max_path_len, status, s, i > 0)))
                return rc;
        return IRS_OK;
    }
    bool reuse = ctx->graph_exec && ctx->graph_B == B && ctx->graph_P == max_path_len && ctx->graph_k == k && ctx->graph_sweep == sweep &&
                 ctx->graph_sample == sample && ctx->graph_sample_k == sample_k && ctx->graph_seq == seq &&
                 ctx->graph_user == user && ctx->graph_hep == hep && ctx->graph_paths == paths &&
                 ctx->graph_status == status && ctx->graph_seed == seed && ctx->prof_family == IRS_PROF_NONE;
    if (!reuse) {
        if (ctx->graph_exec) {
            hipGraphExecDestroy(ctx->graph_exec);
            ctx->graph_exec = nullptr;
        }
        int saved_prof = ctx->prof_family;
        ctx->prof_family = IRS_PROF_NONE; // event records are not captured
        hipStream_t cs;
        IRS_CHECK_HIP(ctx, hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
        hipGraph_t graph = nullptr;
        hipError_t e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
        if (e == hipSuccess) {
            rc = enqueue_step(ctx, seq, user, hep, B, k, sweep, sample, sample_k, seed, paths, max_path_len, status, cs);
            hipError_t e2 = hipStreamEndCapture(cs, &graph);
            if (rc == IRS_OK && e2 != hipSuccess) e = e2;
        }
        ctx->prof_family = saved_prof;
        if (e != hipSuccess || rc != IRS_OK || !graph) {
            hipStreamDestroy(cs);
            if (graph) hipGraphDestroy(graph);
            if (rc) return rc;
            IRS_FAIL(ctx, IRS_E_HIP, "graph capture failed: %s", hipGetErrorString(e));
        }
        e = hipGraphInstantiate(&ctx->graph_exec, graph, nullptr, nullptr, 0);
        hipGraphDestroy(graph);
        hipStreamDestroy(cs);
        if (e != hipSuccess) {
            ctx->graph_exec = nullptr;
            IRS_FAIL(ctx, IRS_E_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
        }
        ctx->graph_B = B;
        ctx->graph_P = max_path_len;
        ctx->graph_k = k;
        ctx->graph_sweep = sweep;
        ctx->graph_sample = sample;
        ctx->graph_sample_k = sample_k;
        ctx->graph_seq = seq;
        ctx->graph_user = (void *)user;
        ctx->graph_hep = hep;
        ctx->graph_paths = paths;
        ctx->graph_status = status;
        ctx->graph_seed = seed;
    }
    for (int i = 0; i < max_path_len; ++i) IRS_CHECK_HIP(ctx, hipGraphLaunch(ctx->graph_exec, s));
    return IRS_OK;
}

// ------------------------------------------------------------------ beam search
extern "C" int irs_beam_step(irs_ctx *ctx, const int64_t *seq_in, const int32_t *hep_in, const double *cum_in,
                             const float *paths_in, const float *val, const int64_t *ids0, const float *lse_max,
                             const float *lse_sum, int32_t B, int32_t W, int32_t k, int32_t step, int32_t P,
                             int64_t *seq_out, int32_t *hep_out, double *cum_out, float *paths_out, int32_t *status,
                             void *stream) {
    if (!ctx) return IRS_E_INVALID;
    if (!seq_in || !hep_in || !cum_in || !paths_in || !val || !ids0 || !seq_out || !hep_out || !cum_out || !paths_out ||
        !status || B < 1 || W < 1 || k < 1 || P < 1 || step < 0 || step >= P)
        IRS_FAIL(ctx, IRS_E_INVALID, "irs_beam_step: bad arguments");
    if (W > 1 && (!lse_max || !lse_sum)) IRS_FAIL(ctx, IRS_E_INVALID, "irs_beam_step: W > 1 needs the row log-sum-exp");
    return irs_launch_beam_step(ctx, seq_in, hep_in, cum_in, paths_in, val, ids0, W > 1 ? lse_max : nullptr,
                                W > 1 ? lse_sum : nullptr, B, W, k, step, nullptr, P, seq_out, hep_out, cum_out,
                                paths_out, status, (hipStream_t)stream);
}

static int enqueue_beam_step(irs_ctx *ctx, int in, int B, int W, int k, int sweep, int P, int32_t *status, hipStream_t s) {
    const int out = in ^ 1, rows = B * W;
    int rc;
    if ((rc = irs_launch_decode(ctx, ctx->bm_seq[in], ctx->bm_user, rows, nullptr, ctx->bm_hep[in], ctx->xrows, nullptr, s))) return rc;
    // W > 1: top-k and log-sum-exp out of one call (one pass over the float32 catalog on the swept path)
    if ((rc = irs_launch_topk(ctx, ctx->xrows, rows, k, sweep, ctx->top_val, ctx->top_ids, ctx->row_status, s, nullptr,
                              W > 1 ? ctx->lse_max : nullptr, W > 1 ? ctx->lse_sum : nullptr)))
        return rc;
    if ((rc = irs_launch_beam_step(ctx, ctx->bm_seq[in], ctx->bm_hep[in], ctx->bm_cum[in], ctx->bm_paths[in], ctx->top_val,
                                   ctx->top_ids, W > 1 ? ctx->lse_max : nullptr, W > 1 ? ctx->lse_sum : nullptr, B, W, k, 0,
                                   ctx->step_ctr, P, ctx->bm_seq[out], ctx->bm_hep[out], ctx->bm_cum[out],
                                   ctx->bm_paths[out], status, s)))
        return rc;
    return irs_launch_inc(ctx, ctx->step_ctr, s);
}

extern "C" int irs_beam_search(irs_ctx *ctx, const int64_t *seq0, const int64_t *user, const int32_t *hep0, int32_t B,
                               int32_t W, int32_t P, int32_t k, int32_t sweep, int32_t use_graph, float *paths,
                               double *scores, int64_t *seq_final, int32_t *status, void *stream) {
    int rc = ready_filter(ctx, sweep);
    if (rc) return rc;
    if (ctx->shard.world != 1) IRS_FAIL(ctx, IRS_E_UNSUPPORTED, "irs_beam_search needs the whole catalog on one device");
    if (!seq0 || !hep0 || !paths || !scores || !status || B < 1) IRS_FAIL(ctx, IRS_E_INVALID, "irs_beam_search: bad arguments");
    if (W < 1 || W > 32) IRS_FAIL(ctx, IRS_E_INVALID, "irs_beam_search: beam width must be in [1, 32]");
    if (P < 1 || P > IRS_MAX_PATH) IRS_FAIL(ctx, IRS_E_INVALID, "irs_beam_search: path length must be in [1, %d]", IRS_MAX_PATH);
    if ((int64_t)B * W > ctx->max_seqs || (int64_t)B * W > ctx->max_rows)
        IRS_FAIL(ctx, IRS_E_INVALID, "irs_beam_search: B*W=%d exceeds max_seqs=%d / max_rows=%d", B * W, ctx->max_seqs, ctx->max_rows);
    if (k < 1 || k > ctx->dims.max_k) IRS_FAIL(ctx, IRS_E_INVALID, "irs_beam_search: bad k");
    if (sweep != IRS_SWEEP_BF16 && sweep != IRS_SWEEP_F32) IRS_FAIL(ctx, IRS_E_INVALID, "irs_beam_search: bad sweep");
    if (ctx->dims.mask_mode == IRS_MASK_IRN && !user) IRS_FAIL(ctx, IRS_E_INVALID, "irs_beam_search: user is null");
    hipStream_t s = (hipStream_t)stream;
    IRS_CHECK_HIP(ctx, hipMemsetAsync(ctx->step_ctr, 0, 2 * sizeof(int32_t), s));
    IRS_CHECK_HIP(ctx, hipMemsetAsync(status, 0, sizeof(int32_t) * B, s));
    if ((rc = irs_launch_beam_init(ctx, seq0, user, hep0, B, W, P, ctx->bm_seq[0], ctx->bm_user, ctx->bm_hep[0],
                                   ctx->bm_cum[0], ctx->bm_paths[0], s)))
        return rc;
    int done = 0;
    if (use_graph && P >= 2 && ctx->prof_family == IRS_PROF_NONE) {
        // NOTE: `status` is baked into the captured graph, so it is part of the cache key via its address below
        static_assert(sizeof(void *) == 8, "64-bit only");
        bool reuse = ctx->beam_graph && ctx->beam_B == B && ctx->beam_W == W && ctx->beam_k == k &&
                     ctx->beam_sweep == sweep && ctx->beam_P == P && ctx->beam_status == (void *)status;
        if (!reuse) {
            if (ctx->beam_graph) {
                hipGraphExecDestroy(ctx->beam_graph);
                ctx->beam_graph = nullptr;
            }
            hipStream_t cs;
            IRS_CHECK_HIP(ctx, hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
            hipGraph_t graph = nullptr;
            hipError_t e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
            if (e == hipSuccess) {
                rc = enqueue_beam_step(ctx, 0, B, W, k, sweep, P, status, cs);
                if (rc == IRS_OK) rc = enqueue_beam_step(ctx, 1, B, W, k, sweep, P, status, cs);
                hipError_t e2 = hipStreamEndCapture(cs, &graph);
                if (rc == IRS_OK && e2 != hipSuccess) e = e2;
            }
            if (e == hipSuccess && rc == IRS_OK && graph) e = hipGraphInstantiate(&ctx->beam_graph, graph, nullptr, nullptr, 0);
            if (graph) hipGraphDestroy(graph);
            hipStreamDestroy(cs);
            if (rc) return rc;
            if (e != hipSuccess) {
                ctx->beam_graph = nullptr;
                IRS_FAIL(ctx, IRS_E_HIP, "beam graph capture failed: %s", hipGetErrorString(e));
            }
            ctx->beam_B = B;
            ctx->beam_W = W;
            ctx->beam_k = k;
            ctx->beam_sweep = sweep;
            ctx->beam_P = P;
            ctx->beam_status = status;
        }
        for (; done + 2 <= P; done += 2) IRS_CHECK_HIP(ctx, hipGraphLaunch(ctx->beam_graph, s));
    }
    for (; done < P; ++done)
        if ((rc = enqueue_beam_step(ctx, done & 1, B, W, k, sweep, P, status, s))) return rc;
    const int fin = P & 1;
    const size_t rows = (size_t)B * W;
    IRS_CHECK_HIP(ctx, hipMemcpyAsync(paths, ctx->bm_paths[fin], rows * P * sizeof(float), hipMemcpyDeviceToDevice, s));
    IRS_CHECK_HIP(ctx, hipMemcpyAsync(scores, ctx->bm_cum[fin], rows * sizeof(double), hipMemcpyDeviceToDevice, s));
    if (seq_final)
        IRS_CHECK_HIP(ctx, hipMemcpyAsync(seq_final, ctx->bm_seq[fin], rows * ctx->dims.max_len * sizeof(int64_t), hipMemcpyDeviceToDevice, s));
    return IRS_OK;
}

// ------------------------------------------------------------------ profiling hooks
void irs_prof_begin(irs_ctx *ctx, int family, hipStream_t s) {
    if (ctx->prof_family != family) return;
    if (ctx->prof_n == ctx->prof_cap) {
        int ncap = ctx->prof_cap ? ctx->prof_cap * 2 : 256;
        irs_prof_ev *n = (irs_prof_ev *)realloc(ctx->prof_ev, sizeof(irs_prof_ev) * ncap);
        if (!n) return;
        for (int i = ctx->prof_cap; i < ncap; ++i) {
            hipEventCreate(&n[i].a);
            hipEventCreate(&n[i].b);
        }
        ctx->prof_ev = n;
        ctx->prof_cap = ncap;
    }
    hipEventRecord(ctx->prof_ev[ctx->prof_n].a, s);
}

void irs_prof_end(irs_ctx *ctx, int family, hipStream_t s, double flops, double bytes) {
    if (ctx->prof_family != family) return;
    if (ctx->prof_n >= ctx->prof_cap) return;
    hipEventRecord(ctx->prof_ev[ctx->prof_n].b, s);
    ctx->prof_n++;
    ctx->prof_flops += flops;
    ctx->prof_bytes += bytes;
}

extern "C" int irs_prof_enable(irs_ctx *ctx, int32_t family) {
    if (!ctx) return IRS_E_INVALID;
    ctx->prof_family = family;
    ctx->prof_n = 0;
    ctx->prof_flops = 0;
    ctx->prof_bytes = 0;
    return IRS_OK;
}

extern "C" int irs_prof_read(irs_ctx *ctx, int32_t *launches, double *total_ms, double *total_flops, double *total_bytes) {
    if (!ctx) return IRS_E_INVALID;
    double ms = 0;
    for (int i = 0; i < ctx->prof_n; ++i) {
        IRS_CHECK_HIP(ctx, hipEventSynchronize(ctx->prof_ev[i].b));
        float t = 0;
        IRS_CHECK_HIP(ctx, hipEventElapsedTime(&t, ctx->prof_ev[i].a, ctx->prof_ev[i].b));
        ms += t;
    }
    if (launches) *launches = ctx->prof_n;
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = ctx->prof_flops;
    if (total_bytes) *total_bytes = ctx->prof_bytes;
    ctx->prof_n = 0;
    ctx->prof_flops = 0;
    ctx->prof_bytes = 0;
    return IRS_OK;
}
