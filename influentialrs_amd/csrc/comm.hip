// Multi-GPU exchange below the C ABI (include/irs_hip.h, "multi-GPU" section):
// a communicator (RCCL through dlopen, or caller-supplied collectives), the two exchange steps, and the item-sharded
// greedy and beam search loops as ONE stream-ordered sequence per step over buffers of the context's workspace.
// The reference's loop being replaced: IRSNN.get_seq_in_batch, /root/reference/model/influentialRS.py:412-450; its
// only multi-GPU construct is nn.DataParallel (pipeline.py:43-44), which this does not resemble: rows are
// data-parallel, the catalog is item-sharded, and what crosses xGMI per step is M x d floats of rows and M x k packed
// 64-bit keys per rank (latency-bound payloads).
#include <dlfcn.h>
#include <stdlib.h>

#include <new>

#include "irs_internal.h"

// ---- RCCL: types and constants come from rccl.h (compile time); the LIBRARY (570 MB) is loaded on demand with dlopen and
//      its functions are reached through the pointers below, so libirs_hip.so carries no link-time dependency on it.
#include <rccl/rccl.h>
typedef ncclComm_t irs_nccl_comm_t;
typedef ncclUniqueId irs_nccl_uid;
enum { IRS_NCCL_SUM = ncclSum, IRS_NCCL_MAX = ncclMax, IRS_NCCL_INT8 = ncclInt8, IRS_NCCL_FLOAT32 = ncclFloat32 };
static_assert(sizeof(ncclUniqueId) == IRS_COMM_ID_BYTES, "include/irs_hip.h: IRS_COMM_ID_BYTES must equal NCCL_UNIQUE_ID_BYTES");
static_assert(NCCL_MAJOR == 2, "the call signatures below are the NCCL 2.x ABI");

struct irs_rccl_api {
    void *lib;
    int version;
    int (*GetVersion)(int *);
    int (*GetUniqueId)(irs_nccl_uid *);
    int (*CommInitRank)(irs_nccl_comm_t *, int, irs_nccl_uid, int);
    int (*CommDestroy)(irs_nccl_comm_t);
    int (*AllGather)(const void *, void *, size_t, int, irs_nccl_comm_t, hipStream_t);
    int (*AllReduce)(const void *, void *, size_t, int, int, irs_nccl_comm_t, hipStream_t);
    int (*AllToAll)(const void *, void *, size_t, int, irs_nccl_comm_t, hipStream_t); // RCCL extension (may be absent)
    int (*Send)(const void *, size_t, int, int, irs_nccl_comm_t, hipStream_t);
    int (*Recv)(void *, size_t, int, int, irs_nccl_comm_t, hipStream_t);
    int (*GroupStart)(void);
    int (*GroupEnd)(void);
    const char *(*GetErrorString)(int);
};

static irs_rccl_api g_rccl;
static char g_comm_err[512] = "";

struct irs_comm {
    int rank, world;
    bool rccl;
    bool p2p_alltoall; // exchange through grouped ncclSend / ncclRecv (the library lacks ncclAllToAll, or IRS_RCCL_NO_ALLTOALL=1)
    irs_nccl_comm_t nccl;
    void *user;
    irs_allgather_fn allgather;
    irs_alltoall_fn alltoall;
    irs_allreduce_f32_fn allreduce;
};

extern "C" const char *irs_comm_last_error(void) { return g_comm_err; }

static int load_rccl() {
    if (g_rccl.lib) return IRS_OK;
    // One HIP runtime per process: a host that has already loaded an RCCL (PyTorch ships its own librccl.so, linked
    // against its own libamdhip64) must get THAT copy -- RTLD_NOLOAD first -- and only a process without one loads the
    // system library.  IRS_RCCL_PATH overrides.
    const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void *lib = nullptr;
    if (const char *env = getenv("IRS_RCCL_PATH")) lib = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
    for (int i = 0; i < 2 && !lib; ++i) lib = dlopen(names[i], RTLD_NOW | RTLD_NOLOAD);
    for (const char *n : names)
        if (lib || (lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!lib) {
        snprintf(g_comm_err, sizeof(g_comm_err), "dlopen(librccl.so) failed: %s", dlerror());
        return IRS_E_STATE;
    }
#define SYM(field, name, required)                                                        \
    do {                                                                                  \
        *(void **)(&g_rccl.field) = dlsym(lib, name);                                     \
        if (required && !g_rccl.field) {                                                  \
            snprintf(g_comm_err, sizeof(g_comm_err), "librccl.so lacks %s", name);        \
            dlclose(lib);                                                                 \
            return IRS_E_STATE;                                                           \
        }                                                                                 \
    } while (0)
    SYM(GetVersion, "ncclGetVersion", true);
    SYM(GetUniqueId, "ncclGetUniqueId", true);
    SYM(CommInitRank, "ncclCommInitRank", true);
    SYM(CommDestroy, "ncclCommDestroy", true);
    SYM(AllGather, "ncclAllGather", true);
    SYM(AllReduce, "ncclAllReduce", true);
    SYM(AllToAll, "ncclAllToAll", false);
    SYM(Send, "ncclSend", true);
    SYM(Recv, "ncclRecv", true);
    SYM(GroupStart, "ncclGroupStart", true);
    SYM(GroupEnd, "ncclGroupEnd", true);
    SYM(GetErrorString, "ncclGetErrorString", true);
#undef SYM
    // the library actually loaded (possibly the host process's own copy) must speak the ABI this file was compiled against:
    // the enum values and call signatures are stable within NCCL major version 2
    int ver = 0;
    if (g_rccl.GetVersion(&ver) != 0 || ver / 10000 != NCCL_MAJOR) {
        snprintf(g_comm_err, sizeof(g_comm_err), "librccl.so reports NCCL version code %d; this library was built against %d.%d (major %d required)",
                 ver, NCCL_MAJOR, NCCL_MINOR, NCCL_MAJOR);
        memset(&g_rccl, 0, sizeof(g_rccl));
        dlclose(lib);
        return IRS_E_STATE;
    }
    g_rccl.version = ver;
    g_rccl.lib = lib;
    return IRS_OK;
}

#define NCCL_OK_OR_FAIL(expr, what)                                                                          \
    do {                                                                                                     \
        int r_ = (expr);                                                                                     \
        if (r_ != 0) {                                                                                       \
            snprintf(g_comm_err, sizeof(g_comm_err), "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?"); \
            return IRS_E_HIP;                                                                                \
        }                                                                                                    \
    } while (0)

extern "C" int irs_comm_unique_id(void *out_id128) {
    if (!out_id128) return IRS_E_INVALID;
    int rc = load_rccl();
    if (rc) return rc;
    irs_nccl_uid id;
    NCCL_OK_OR_FAIL(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
    memcpy(out_id128, id.internal, IRS_COMM_ID_BYTES);
    return IRS_OK;
}

extern "C" int irs_comm_init_rccl(irs_comm **out, const void *id128, int32_t rank, int32_t world) {
    if (!out || !id128 || world < 1 || rank < 0 || rank >= world) {
        snprintf(g_comm_err, sizeof(g_comm_err), "irs_comm_init_rccl: bad arguments");
        return IRS_E_INVALID;
    }
    int rc = load_rccl();
    if (rc) return rc;
    irs_comm *c = new (std::nothrow) irs_comm();
    if (!c) return IRS_E_INVALID;
    memset(c, 0, sizeof(*c));
    c->rank = rank;
    c->world = world;
    c->rccl = true;
    // ncclAllToAll is an RCCL extension: without it -- or on request, so that the fallback can be exercised on a library that
    // has it -- the key exchange runs as grouped point-to-point transfers
    const char *no_a2a = getenv("IRS_RCCL_NO_ALLTOALL");
    c->p2p_alltoall = !g_rccl.AllToAll || (no_a2a && no_a2a[0] == '1');
    irs_nccl_uid id;
    memcpy(id.internal, id128, IRS_COMM_ID_BYTES);
    int r = g_rccl.CommInitRank(&c->nccl, world, id, rank);
    if (r != 0) {
        snprintf(g_comm_err, sizeof(g_comm_err), "ncclCommInitRank(rank %d of %d): %s", rank, world, g_rccl.GetErrorString(r));
        delete c;
        return IRS_E_HIP;
    }
    // one tiny collective of each kind outside any capture: RCCL sets up its channels lazily on first use
    void *tmp = nullptr;
    if (hipMalloc(&tmp, (size_t)world * 512 + 512) == hipSuccess) {
        char *t = (char *)tmp;
        r = g_rccl.AllGather(t, t + 256, 4, IRS_NCCL_INT8, c->nccl, nullptr);
        if (!r) r = g_rccl.AllReduce(t, t, 1, IRS_NCCL_FLOAT32, IRS_NCCL_SUM, c->nccl, nullptr);
        if (!r && !c->p2p_alltoall) r = g_rccl.AllToAll(t + 256, t + 256 + (size_t)world * 128, 4, IRS_NCCL_INT8, c->nccl, nullptr);
        if (!r && c->p2p_alltoall) { // the fallback's first use sets up the point-to-point channels
            r = g_rccl.GroupStart();
            for (int p = 0; p < world && !r; ++p) {
                r = g_rccl.Send(t + 256 + (size_t)p * 4, 4, IRS_NCCL_INT8, p, c->nccl, nullptr);
                if (!r) r = g_rccl.Recv(t + 256 + (size_t)world * 128 + (size_t)p * 4, 4, IRS_NCCL_INT8, p, c->nccl, nullptr);
            }
            int r2 = g_rccl.GroupEnd();
            if (!r) r = r2;
        }
        hipError_t he = hipStreamSynchronize(nullptr);
        (void)hipFree(tmp);
        if (r != 0 || he != hipSuccess) {
            snprintf(g_comm_err, sizeof(g_comm_err), "RCCL warm-up collectives failed: %s", r ? g_rccl.GetErrorString(r) : hipGetErrorString(he));
            g_rccl.CommDestroy(c->nccl);
            delete c;
            return IRS_E_HIP;
        }
    }
    *out = c;
    return IRS_OK;
}

extern "C" int irs_comm_init_callbacks(irs_comm **out, int32_t rank, int32_t world, void *user, irs_allgather_fn allgather,
                                       irs_alltoall_fn alltoall, irs_allreduce_f32_fn allreduce) {
    if (!out || world < 1 || rank < 0 || rank >= world || !allgather || !alltoall || !allreduce) {
        snprintf(g_comm_err, sizeof(g_comm_err), "irs_comm_init_callbacks: bad arguments");
        return IRS_E_INVALID;
    }
    irs_comm *c = new (std::nothrow) irs_comm();
    if (!c) return IRS_E_INVALID;
    memset(c, 0, sizeof(*c));
    c->rank = rank;
    c->world = world;
    c->user = user;
    c->allgather = allgather;
    c->alltoall = alltoall;
    c->allreduce = allreduce;
    *out = c;
    return IRS_OK;
}

extern "C" void irs_comm_destroy(irs_comm *c) {
    if (!c) return;
    if (c->rccl && c->nccl && g_rccl.CommDestroy) g_rccl.CommDestroy(c->nccl);
    delete c;
}

extern "C" int irs_comm_is_rccl(const irs_comm *c) { return c && c->rccl ? 1 : 0; }
// 0: callbacks; 1: RCCL with ncclAllToAll; 2: RCCL with the grouped Send / Recv exchange
extern "C" int irs_comm_exchange_kind(const irs_comm *c) { return !c || !c->rccl ? 0 : (c->p2p_alltoall ? 2 : 1); }
extern "C" int irs_comm_rccl_version(void) { return g_rccl.lib ? g_rccl.version : 0; }

// ---- collectives (bytes; stream-ordered)
static int comm_allgather(irs_ctx *ctx, irs_comm *c, const void *send, void *recv, size_t bytes, hipStream_t s) {
    if (c->rccl) {
        int r = g_rccl.AllGather(send, recv, bytes, IRS_NCCL_INT8, c->nccl, s);
        if (r) IRS_FAIL(ctx, IRS_E_HIP, "ncclAllGather: %s", g_rccl.GetErrorString(r));
        return IRS_OK;
    }
    if (c->allgather(c->user, send, recv, bytes, (void *)s)) IRS_FAIL(ctx, IRS_E_HIP, "all-gather callback failed");
    return IRS_OK;
}

static int comm_alltoall(irs_ctx *ctx, irs_comm *c, const void *send, void *recv, size_t bytes, hipStream_t s) {
    if (c->rccl) {
        int r;
        if (!c->p2p_alltoall) r = g_rccl.AllToAll(send, recv, bytes, IRS_NCCL_INT8, c->nccl, s);
        else { // grouped point-to-point: the same exchange on any NCCL-compatible library
            r = g_rccl.GroupStart();
            for (int p = 0; p < c->world && !r; ++p) {
                r = g_rccl.Send((const char *)send + (size_t)p * bytes, bytes, IRS_NCCL_INT8, p, c->nccl, s);
                if (!r) r = g_rccl.Recv((char *)recv + (size_t)p * bytes, bytes, IRS_NCCL_INT8, p, c->nccl, s);
            }
            int r2 = g_rccl.GroupEnd();
            if (!r) r = r2;
        }
        if (r) IRS_FAIL(ctx, IRS_E_HIP, "ncclAllToAll: %s", g_rccl.GetErrorString(r));
        return IRS_OK;
    }
    if (c->alltoall(c->user, send, recv, bytes, (void *)s)) IRS_FAIL(ctx, IRS_E_HIP, "all-to-all callback failed");
    return IRS_OK;
}

static int comm_allreduce(irs_ctx *ctx, irs_comm *c, float *buf, size_t count, int op, hipStream_t s) {
    if (c->rccl) {
        int r = g_rccl.AllReduce(buf, buf, count, IRS_NCCL_FLOAT32, op == IRS_REDUCE_MAX ? IRS_NCCL_MAX : IRS_NCCL_SUM, c->nccl, s);
        if (r) IRS_FAIL(ctx, IRS_E_HIP, "ncclAllReduce: %s", g_rccl.GetErrorString(r));
        return IRS_OK;
    }
    if (c->allreduce(c->user, buf, count, op, (void *)s)) IRS_FAIL(ctx, IRS_E_HIP, "all-reduce callback failed");
    return IRS_OK;
}

static int comm_check(irs_ctx *ctx, const irs_comm *c, const char *fn) {
    if (!ctx) return IRS_E_INVALID;
    if (!c) IRS_FAIL(ctx, IRS_E_INVALID, "%s: null communicator", fn);
    if (c->world != ctx->shard.world || c->rank != ctx->shard.rank)
        IRS_FAIL(ctx, IRS_E_INVALID, "%s: communicator is rank %d of %d, the context's shard is rank %d of %d", fn, c->rank, c->world,
                 ctx->shard.rank, ctx->shard.world);
    return IRS_OK;
}

extern "C" int irs_allgather_rows(irs_ctx *ctx, irs_comm *comm, const float *rows_local, int32_t B, float *rows_all, void *stream) {
    int rc = comm_check(ctx, comm, "irs_allgather_rows");
    if (rc) return rc;
    if (!rows_local || !rows_all || B < 1) IRS_FAIL(ctx, IRS_E_INVALID, "irs_allgather_rows: bad arguments");
    return comm_allgather(ctx, comm, rows_local, rows_all, (size_t)B * ctx->dims.d * sizeof(float), (hipStream_t)stream);
}

extern "C" int irs_exchange_topk(irs_ctx *ctx, irs_comm *comm, const uint64_t *keys_send, uint64_t *keys_recv, int32_t B, int32_t k,
                                 void *stream) {
    int rc = comm_check(ctx, comm, "irs_exchange_topk");
    if (rc) return rc;
    if (!keys_send || !keys_recv || B < 1 || k < 1) IRS_FAIL(ctx, IRS_E_INVALID, "irs_exchange_topk: bad arguments");
    return comm_alltoall(ctx, comm, keys_send, keys_recv, (size_t)B * k * sizeof(uint64_t), (hipStream_t)stream);
}

extern "C" int irs_set_sharded_overlap(irs_ctx *ctx, int32_t on) {
    if (!ctx) return IRS_E_INVALID;
    ctx->sh_overlap = on ? 1 : 0;
    return IRS_OK;
}
extern "C" int irs_get_sharded_overlap(const irs_ctx *ctx) { return ctx ? ctx->sh_overlap : IRS_E_INVALID; }

extern "C" int irs_sharded_graph_state(const irs_ctx *ctx) {
    return ctx ? ((ctx->sh_graph ? 1 : 0) | (ctx->sh_nograph ? 2 : 0)) : 0;
}

// ---- small kernels of the sharded loops
// global (max, sum exp) of a row from the per-shard pairs: gm = all-reduced max (already in gmax); the local sum is
// rescaled to it before the sum all-reduce.  A shard whose maximum is -inf (no items) contributes 0.
__global__ void k_lse_rescale(const float *__restrict__ lmax, const float *__restrict__ gmax, float *__restrict__ lsum, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float m = lmax[i], g = gmax[i];
        lsum[i] = (m == -INFINITY) ? 0.f : lsum[i] * expf(m - g);
    }
}
// status bit of a merged list: fewer than k entries in the whole catalog
__global__ void k_short_list(const int64_t *__restrict__ ids0, int k, int n, int32_t *__restrict__ status) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && ids0[(size_t)i * k + (k - 1)] < 0) status[i] |= IRS_ROW_FEWER_THAN_K;
}

static int ready_sharded(irs_ctx *ctx, irs_comm *comm, const char *fn, int sweep) {
    int rc = comm_check(ctx, comm, fn);
    if (rc) return rc;
    if (!ctx->finalized) IRS_FAIL(ctx, IRS_E_STATE, "weights not finalized (irs_finalize_weights)");
    if (!ctx->ws) IRS_FAIL(ctx, IRS_E_STATE, "workspace not bound (irs_bind_workspace)");
    if (ctx->proj_stale && sweep == IRS_SWEEP_BF16)
        IRS_FAIL(ctx, IRS_E_STATE, "project.* may have changed since irs_finalize_weights: call it before filtering through the bf16 catalog");
    if (sweep != IRS_SWEEP_BF16 && sweep != IRS_SWEEP_F32) IRS_FAIL(ctx, IRS_E_INVALID, "%s: bad sweep", fn);
    return IRS_OK;
}

// capture `body` into a graph once per key and replay it `times` times; falls back to plain launches when the
// communicator cannot be captured (callbacks) or capture fails with the collectives inside
template <typename F>
static int run_steps(irs_ctx *ctx, irs_comm *comm, bool use_graph, hipGraphExec_t *exec, bool reuse, int times, hipStream_t s, F &&body) {
    int rc;
    if (use_graph && comm->rccl && ctx->prof_family == IRS_PROF_NONE && !ctx->sh_nograph) {
        if (!reuse || !*exec) {
            if (*exec) {
                (void)hipGraphExecDestroy(*exec);
                *exec = nullptr;
            }
            hipStream_t cs;
            IRS_CHECK_HIP(ctx, hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
            hipGraph_t graph = nullptr;
            hipError_t e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
            rc = IRS_OK;
            if (e == hipSuccess) {
                rc = body(cs);
                hipError_t e2 = hipStreamEndCapture(cs, &graph);
                if (rc == IRS_OK && e2 != hipSuccess) e = e2;
            }
            if (e == hipSuccess && rc == IRS_OK && graph) e = hipGraphInstantiate(exec, graph, nullptr, nullptr, 0);
            if (graph) (void)hipGraphDestroy(graph);
            (void)hipStreamDestroy(cs);
            if (rc != IRS_OK || e != hipSuccess) {
                // nothing of a captured step has executed: the state is untouched.  A collective library that cannot be
                // captured makes this context fall back to plain stream launches from now on (same results).
                *exec = nullptr;
                ctx->sh_nograph = 1;
                (void)hipGetLastError();
            }
        }
        if (*exec) {
            for (int i = 0; i < times; ++i) IRS_CHECK_HIP(ctx, hipGraphLaunch(*exec, s));
            return IRS_OK;
        }
    }
    for (int i = 0; i < times; ++i)
        if ((rc = body(s))) return rc;
    return IRS_OK;
}

// ------------------------------------------------------------------ greedy / sampled search, item-sharded
extern "C" int irs_generate_paths_sharded(irs_ctx *ctx, irs_comm *comm, int64_t *seq, const int64_t *user, int32_t *hep, int32_t B,
                                          int32_t max_path_len, int32_t k, int32_t sweep, int32_t sample, int32_t sample_k,
                                          uint64_t seed, int32_t use_graph, float *paths, int32_t *status, void *stream) {
    int rc = ready_sharded(ctx, comm, "irs_generate_paths_sharded", sweep);
    if (rc) return rc;
    const int world = comm->world;
    if (!seq || !hep || !paths || !status || B < 1 || max_path_len < 1) IRS_FAIL(ctx, IRS_E_INVALID, "irs_generate_paths_sharded: bad arguments");
    if (ctx->dims.mask_mode == IRS_MASK_IRN && !user) IRS_FAIL(ctx, IRS_E_INVALID, "irs_generate_paths_sharded: user is null");
    if (B > ctx->max_seqs || (int64_t)B * world > ctx->max_rows)
        IRS_FAIL(ctx, IRS_E_INVALID, "irs_generate_paths_sharded: B=%d needs max_seqs >= B and max_rows >= world * B = %d", B, B * world);
    if (k < 1 || k > ctx->dims.max_k || (int64_t)k * world > 2048) IRS_FAIL(ctx, IRS_E_INVALID, "irs_generate_paths_sharded: bad k");
    if (sample && (sample_k < 1 || sample_k > IRS_MAX_SAMPLE_K))
        IRS_FAIL(ctx, IRS_E_UNSUPPORTED, "irs_generate_paths_sharded: sample_k must be in [1, %d]", IRS_MAX_SAMPLE_K);
    hipStream_t s = (hipStream_t)stream;
    const int rows = B * world;
    IRS_CHECK_HIP(ctx, hipMemsetAsync(ctx->step_ctr, 0, 2 * sizeof(int32_t), s));
    IRS_CHECK_HIP(ctx, hipMemsetAsync(status, 0, sizeof(int32_t) * B, s));
    int step_no = 0; // (stream launches: steps behind the first may reuse the previous step's emission thresholds; a captured step never does)
    const bool graph_path = use_graph && comm->rccl && ctx->prof_family == IRS_PROF_NONE && !ctx->sh_nograph;
    auto body = [&](hipStream_t q) -> int {
        int r;
        const int carry = (!graph_path && step_no++ > 0) ? 1 : 0;
        if ((r = irs_launch_decode(ctx, seq, user, B, nullptr, hep, ctx->x_local, nullptr, q))) return r;
        if ((r = comm_allgather(ctx, comm, ctx->x_local, ctx->xrows, (size_t)B * ctx->dims.d * sizeof(float), q))) return r;
        if ((r = irs_launch_topk(ctx, ctx->xrows, rows, k, sweep, ctx->top_val, ctx->top_ids, ctx->row_status, q, nullptr, nullptr, nullptr, carry)))
            return r;
        if ((r = irs_launch_pack_topk(ctx, ctx->top_val, ctx->top_ids, (int64_t)rows * k, ctx->keys_send, q))) return r;
        if ((r = comm_alltoall(ctx, comm, ctx->keys_send, ctx->keys_recv, (size_t)B * k * sizeof(uint64_t), q))) return r;
        if ((r = irs_launch_merge_keys(ctx, ctx->keys_recv, world, B, k, ctx->top_val, ctx->top_ids, q))) return r;
        hipLaunchKernelGGL(k_short_list, dim3((B + 255) / 256), dim3(256), 0, q, ctx->top_ids, k, B, status);
        if ((r = irs_launch_path_step(ctx, seq, hep, B, ctx->top_val, ctx->top_ids, k, 0, ctx->step_ctr, paths, max_path_len, sample,
                                      sample_k, seed, status, q)))
            return r;
        return irs_launch_inc(ctx, ctx->step_ctr, q);
    };
    // ---- opt-in (irs_set_sharded_overlap): the step's users as TWO micro-batches, the collectives on a side stream chained by
    // events.  Half 1's decode runs while half 0's rows are gathered, half 0's sweep while half 1's rows are gathered, half 1's
    // sweep while half 0's keys are exchanged, half 0's merge + path step while half 1's keys are exchanged: per step one
    // all-gather and one all_to_all leave the critical path.  Rows are independent, so the results equal the one-batch loop's bit
    // for bit (greedy choice only: the sampled choice draws its random numbers by row index of the call).  Off by default: RCCL
    // with more than one rank has never run in the build loop, and two micro-batches cost decoder efficiency on small batches.
    const bool overlap = ctx->sh_overlap && !sample && B >= 2 && world >= 1;
    if (overlap && !ctx->sh_side) {
        IRS_CHECK_HIP(ctx, hipStreamCreateWithFlags(&ctx->sh_side, hipStreamNonBlocking));
        for (int i = 0; i < 8; ++i) IRS_CHECK_HIP(ctx, hipEventCreateWithFlags(&ctx->sh_ev[i], hipEventDisableTiming));
    }
    auto body2 = [&](hipStream_t q) -> int {
        int r;
        hipStream_t c = ctx->sh_side;
        const int d = ctx->dims.d, L = ctx->dims.max_len;
        const int Bh[2] = {B / 2, B - B / 2}, o0[2] = {0, B / 2};
        float *xr[2] = {ctx->xrows, ctx->xrows + (size_t)world * Bh[0] * d};
        uint64_t *ks[2] = {ctx->keys_send, ctx->keys_send + (size_t)world * Bh[0] * k};
        uint64_t *kr[2] = {ctx->keys_recv, ctx->keys_recv + (size_t)world * Bh[0] * k};
        for (int h = 0; h < 2; ++h) { // decode, then the row all-gather on the side stream
            if ((r = irs_launch_decode(ctx, seq + (size_t)o0[h] * L, user ? user + o0[h] : nullptr, Bh[h], nullptr, hep + o0[h],
                                       ctx->x_local + (size_t)o0[h] * d, nullptr, q)))
                return r;
            IRS_CHECK_HIP(ctx, hipEventRecord(ctx->sh_ev[h], q));
            IRS_CHECK_HIP(ctx, hipStreamWaitEvent(c, ctx->sh_ev[h], 0));
            if ((r = comm_allgather(ctx, comm, ctx->x_local + (size_t)o0[h] * d, xr[h], (size_t)Bh[h] * d * sizeof(float), c))) return r;
            IRS_CHECK_HIP(ctx, hipEventRecord(ctx->sh_ev[2 + h], c));
        }
        for (int h = 0; h < 2; ++h) { // sweep of this rank's item shard, keys packed; the key exchange on the side stream
            IRS_CHECK_HIP(ctx, hipStreamWaitEvent(q, ctx->sh_ev[2 + h], 0));
            if ((r = irs_launch_topk(ctx, xr[h], world * Bh[h], k, sweep, ctx->top_val, ctx->top_ids, ctx->row_status, q))) return r;
            if ((r = irs_launch_pack_topk(ctx, ctx->top_val, ctx->top_ids, (int64_t)world * Bh[h] * k, ks[h], q))) return r;
            IRS_CHECK_HIP(ctx, hipEventRecord(ctx->sh_ev[4 + h], q));
            IRS_CHECK_HIP(ctx, hipStreamWaitEvent(c, ctx->sh_ev[4 + h], 0));
            if ((r = comm_alltoall(ctx, comm, ks[h], kr[h], (size_t)Bh[h] * k * sizeof(uint64_t), c))) return r;
            IRS_CHECK_HIP(ctx, hipEventRecord(ctx->sh_ev[6 + h], c));
        }
        for (int h = 0; h < 2; ++h) { // merge of the world's lists of this rank's own rows, the path step
            IRS_CHECK_HIP(ctx, hipStreamWaitEvent(q, ctx->sh_ev[6 + h], 0));
            if ((r = irs_launch_merge_keys(ctx, kr[h], world, Bh[h], k, ctx->top_val, ctx->top_ids, q))) return r;
            hipLaunchKernelGGL(k_short_list, dim3((Bh[h] + 255) / 256), dim3(256), 0, q, ctx->top_ids, k, Bh[h], status + o0[h]);
            if ((r = irs_launch_path_step(ctx, seq + (size_t)o0[h] * L, hep + o0[h], Bh[h], ctx->top_val, ctx->top_ids, k, 0, ctx->step_ctr,
                                          paths + (size_t)o0[h] * max_path_len, max_path_len, sample, sample_k, seed, status + o0[h], q)))
                return r;
        }
        return irs_launch_inc(ctx, ctx->step_ctr, q);
    };
    if (overlap) { // (stream launches: the side stream's work is ordered against `s` by the events of every step)
        for (int i = 0; i < max_path_len; ++i)
            if ((rc = body2(s))) return rc;
        return IRS_OK;
    }
    const bool reuse = ctx->sh_graph && ctx->sh_kind == 1 && ctx->sh_comm == comm && ctx->sh_B == B && ctx->sh_W == 1 &&
                       ctx->sh_P == max_path_len && ctx->sh_k == k && ctx->sh_sweep == sweep && ctx->sh_sample == sample &&
                       ctx->sh_sample_k == sample_k && ctx->sh_seed == seed && ctx->sh_ptr[0] == seq && ctx->sh_ptr[1] == (void *)user &&
                       ctx->sh_ptr[2] == hep && ctx->sh_ptr[3] == paths && ctx->sh_ptr[4] == status;
    rc = run_steps(ctx, comm, use_graph != 0, &ctx->sh_graph, reuse, max_path_len, s, body);
    if (rc == IRS_OK && use_graph && comm->rccl) {
        ctx->sh_kind = 1, ctx->sh_comm = comm, ctx->sh_B = B, ctx->sh_W = 1, ctx->sh_P = max_path_len, ctx->sh_k = k, ctx->sh_sweep = sweep;
        ctx->sh_sample = sample, ctx->sh_sample_k = sample_k, ctx->sh_seed = seed;
        ctx->sh_ptr[0] = seq, ctx->sh_ptr[1] = (void *)user, ctx->sh_ptr[2] = hep, ctx->sh_ptr[3] = paths, ctx->sh_ptr[4] = status;
    }
    return rc;
}

// ------------------------------------------------------------------ beam search, item-sharded
extern "C" int irs_beam_search_sharded(irs_ctx *ctx, irs_comm *comm, const int64_t *seq0, const int64_t *user, const int32_t *hep0,
                                       int32_t B, int32_t W, int32_t P, int32_t k, int32_t sweep, int32_t split_decode,
                                       int32_t use_graph, float *paths, double *scores, int64_t *seq_final, int32_t *status,
                                       void *stream) {
    int rc = ready_sharded(ctx, comm, "irs_beam_search_sharded", sweep);
    if (rc) return rc;
    const int world = comm->world, R = B * W;
    if (!seq0 || !hep0 || !paths || !scores || !status || B < 1) IRS_FAIL(ctx, IRS_E_INVALID, "irs_beam_search_sharded: bad arguments");
    if (W < 1 || W > 32) IRS_FAIL(ctx, IRS_E_INVALID, "irs_beam_search_sharded: beam width must be in [1, 32]");
    if (P < 1 || P > IRS_MAX_PATH) IRS_FAIL(ctx, IRS_E_INVALID, "irs_beam_search_sharded: path length must be in [1, %d]", IRS_MAX_PATH);
    if (ctx->dims.mask_mode == IRS_MASK_IRN && !user) IRS_FAIL(ctx, IRS_E_INVALID, "irs_beam_search_sharded: user is null");
    if (k < 1 || k > ctx->dims.max_k || (int64_t)k * world > 2048) IRS_FAIL(ctx, IRS_E_INVALID, "irs_beam_search_sharded: bad k");
    const int rows_all = split_decode ? R : R * world; // rows every rank sweeps
    if (R > ctx->max_seqs || (int64_t)R * world > ctx->max_rows) // (split_decode: the gathered key lists are world x R x k)
        IRS_FAIL(ctx, IRS_E_INVALID, "irs_beam_search_sharded: B*W=%d needs max_seqs >= %d and max_rows >= %d", R, R, R * world);
    // split_decode: rank r decodes beam rows [r per, min((r + 1) per, R)), per = ceil(R / world) -- any world size (round 5: 32
    // beams over 3, 5 or 6 ranks; the last ranks' slices are short or empty).  The row all-gather moves `per` rows per rank, so
    // gathered position g holds beam row g for every g < R and the positions from R on are never read.
    const int per_split = (R + world - 1) / world;
    if (split_decode && (int64_t)per_split * world > ctx->max_rows)
        IRS_FAIL(ctx, IRS_E_INVALID, "irs_beam_search_sharded: split_decode gathers %d rows, max_rows is %d", per_split * world, ctx->max_rows);
    hipStream_t s = (hipStream_t)stream;
    IRS_CHECK_HIP(ctx, hipMemsetAsync(ctx->step_ctr, 0, 2 * sizeof(int32_t), s));
    IRS_CHECK_HIP(ctx, hipMemsetAsync(status, 0, sizeof(int32_t) * B, s));
    if ((rc = irs_launch_beam_init(ctx, seq0, user, hep0, B, W, P, ctx->bm_seq[0], ctx->bm_user, ctx->bm_hep[0], ctx->bm_cum[0],
                                   ctx->bm_paths[0], s)))
        return rc;
    const int d = ctx->dims.d, L = ctx->dims.max_len;
    auto step = [&](int in, hipStream_t q) -> int {
        const int out = in ^ 1;
        int r;
        const float *lmax = nullptr, *lsum = nullptr;
        if (split_decode) {
            const int per = per_split, r0 = comm->rank * per; // this rank's slice of the (replicated) beam windows
            const int mine = R - r0 < per ? (R - r0 > 0 ? R - r0 : 0) : per;
            if (mine > 0 && (r = irs_launch_decode(ctx, ctx->bm_seq[in] + (size_t)r0 * L, ctx->bm_user + r0, mine, nullptr, ctx->bm_hep[in] + r0,
                                                   ctx->x_local, nullptr, q)))
                return r;
            if ((r = comm_allgather(ctx, comm, ctx->x_local, ctx->xrows, (size_t)per * d * sizeof(float), q))) return r;
        } else {
            if ((r = irs_launch_decode(ctx, ctx->bm_seq[in], ctx->bm_user, R, nullptr, ctx->bm_hep[in], ctx->x_local, nullptr, q))) return r;
            if ((r = comm_allgather(ctx, comm, ctx->x_local, ctx->xrows, (size_t)R * d * sizeof(float), q))) return r;
        }
        if ((r = irs_launch_topk(ctx, ctx->xrows, rows_all, k, sweep, ctx->top_val, ctx->top_ids, ctx->row_status, q, nullptr,
                                 W > 1 ? ctx->lse_max : nullptr, W > 1 ? ctx->lse_sum : nullptr)))
            return r;
        if ((r = irs_launch_pack_topk(ctx, ctx->top_val, ctx->top_ids, (int64_t)rows_all * k, ctx->keys_send, q))) return r;
        if (split_decode) { // every rank needs every row's merged list: all-gather of the packed lists
            if ((r = comm_allgather(ctx, comm, ctx->keys_send, ctx->keys_recv, (size_t)R * k * sizeof(uint64_t), q))) return r;
        } else if ((r = comm_alltoall(ctx, comm, ctx->keys_send, ctx->keys_recv, (size_t)R * k * sizeof(uint64_t), q)))
            return r;
        if ((r = irs_launch_merge_keys(ctx, ctx->keys_recv, world, R, k, ctx->top_val, ctx->top_ids, q))) return r;
        if (W > 1) { // the rows' log-softmax normaliser over the WHOLE catalog: max of maxima, rescaled sums
            IRS_CHECK_HIP(ctx, hipMemcpyAsync(ctx->lse_gmax, ctx->lse_max, sizeof(float) * rows_all, hipMemcpyDeviceToDevice, q));
            if ((r = comm_allreduce(ctx, comm, ctx->lse_gmax, rows_all, IRS_REDUCE_MAX, q))) return r;
            hipLaunchKernelGGL(k_lse_rescale, dim3((rows_all + 255) / 256), dim3(256), 0, q, ctx->lse_max, ctx->lse_gmax, ctx->lse_sum, rows_all);
            if ((r = comm_allreduce(ctx, comm, ctx->lse_sum, rows_all, IRS_REDUCE_SUM, q))) return r;
            const int own0 = split_decode ? 0 : comm->rank * R; // this rank's rows inside the gathered order
            lmax = ctx->lse_gmax + own0;
            lsum = ctx->lse_sum + own0;
        }
        if ((r = irs_launch_beam_step(ctx, ctx->bm_seq[in], ctx->bm_hep[in], ctx->bm_cum[in], ctx->bm_paths[in], ctx->top_val,
                                      ctx->top_ids, lmax, lsum, B, W, k, 0, ctx->step_ctr, P, ctx->bm_seq[out], ctx->bm_hep[out],
                                      ctx->bm_cum[out], ctx->bm_paths[out], status, q)))
            return r;
        return irs_launch_inc(ctx, ctx->step_ctr, q);
    };
    int done = 0;
    if (use_graph && comm->rccl && P >= 2 && ctx->prof_family == IRS_PROF_NONE) {
        const bool reuse = ctx->sh_graph && ctx->sh_kind == 2 + (split_decode ? 1 : 0) && ctx->sh_comm == comm && ctx->sh_B == B &&
                           ctx->sh_W == W && ctx->sh_P == P && ctx->sh_k == k && ctx->sh_sweep == sweep && ctx->sh_ptr[4] == status;
        auto two = [&](hipStream_t q) -> int {
            int r = step(0, q);
            return r ? r : step(1, q);
        };
        if ((rc = run_steps(ctx, comm, true, &ctx->sh_graph, reuse, P / 2, s, two))) return rc;
        ctx->sh_kind = 2 + (split_decode ? 1 : 0), ctx->sh_comm = comm, ctx->sh_B = B, ctx->sh_W = W, ctx->sh_P = P, ctx->sh_k = k;
        ctx->sh_sweep = sweep, ctx->sh_ptr[4] = status;
        done = (P / 2) * 2;
    }
    for (; done < P; ++done)
        if ((rc = step(done & 1, s))) return rc;
    const int fin = P & 1;
    IRS_CHECK_HIP(ctx, hipMemcpyAsync(paths, ctx->bm_paths[fin], (size_t)R * P * sizeof(float), hipMemcpyDeviceToDevice, s));
    IRS_CHECK_HIP(ctx, hipMemcpyAsync(scores, ctx->bm_cum[fin], (size_t)R * sizeof(double), hipMemcpyDeviceToDevice, s));
    if (seq_final) IRS_CHECK_HIP(ctx, hipMemcpyAsync(seq_final, ctx->bm_seq[fin], (size_t)R * L * sizeof(int64_t), hipMemcpyDeviceToDevice, s));
    return IRS_OK;
}
