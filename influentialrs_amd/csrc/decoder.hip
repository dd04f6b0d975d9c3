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

// ------------------------------------------------------------------ linear
// Y[M,N] = act(X[M,K] . W[N,K]^T + bias[N]) (+ R[M,N]);  X, W, Y row-major.
// 256 threads = 4 waves (2x2), wave tile 64x64 = 2x2 MFMA 32x32x2f32 tiles.
#define LIN_BM 128
#define LIN_BN 128
#define LIN_BK 16
#define LIN_LD 17

template <bool VEC>
__device__ __forceinline__ void lin_load_tile(const float *__restrict__ P, int rows, int K, int r0, int k0, int tid,
                                              float4 (&v)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int idx = tid + i * 256;
        int r = idx >> 2, c = (idx & 3) * 4;
        int gr = r0 + r, gk = k0 + c;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gr < rows) {
            const float *p = P + (int64_t)gr * K + gk;
            if (VEC) {
                if (gk + 3 < K) t = *reinterpret_cast<const float4 *>(p);
                else {
                    if (gk < K) t.x = p[0];
                    if (gk + 1 < K) t.y = p[1];
                    if (gk + 2 < K) t.z = p[2];
                }
            } else {
                if (gk < K) t.x = p[0];
                if (gk + 1 < K) t.y = p[1];
                if (gk + 2 < K) t.z = p[2];
                if (gk + 3 < K) t.w = p[3];
            }
        }
        v[i] = t;
    }
}

__device__ __forceinline__ void lin_store_tile(float (*S)[LIN_LD], int tid, const float4 (&v)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int idx = tid + i * 256;
        int r = idx >> 2, c = (idx & 3) * 4;
        S[r][c + 0] = v[i].x;
        S[r][c + 1] = v[i].y;
        S[r][c + 2] = v[i].z;
        S[r][c + 3] = v[i].w;
    }
}

template <bool VEC, bool RELU>
__global__ void __launch_bounds__(256) k_linear(const float *__restrict__ X, const float *__restrict__ W,
                                                const float *__restrict__ bias, const float *__restrict__ R,
                                                float *__restrict__ Y, int M, int N, int K) {
    __shared__ float Xs[2][LIN_BM][LIN_LD];
    __shared__ float Ws[2][LIN_BN][LIN_LD];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int li = lane & 31, lk = lane >> 5;
    const int n0 = blockIdx.x * LIN_BN, m0 = blockIdx.y * LIN_BM;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float4 xv[2], wv[2];
    lin_load_tile<VEC>(X, M, K, m0, 0, tid, xv);
    lin_load_tile<VEC>(W, N, K, n0, 0, tid, wv);
    lin_store_tile(Xs[0], tid, xv);
    lin_store_tile(Ws[0], tid, wv);
    __syncthreads();
    const int nkt = (K + LIN_BK - 1) / LIN_BK;
    int cur = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) {
            lin_load_tile<VEC>(X, M, K, m0, (kt + 1) * LIN_BK, tid, xv);
            lin_load_tile<VEC>(W, N, K, n0, (kt + 1) * LIN_BK, tid, wv);
        }
#pragma unroll
        for (int ks = 0; ks < LIN_BK / 2; ++ks) {
            float a0 = Xs[cur][wr * 64 + li][2 * ks + lk];
            float a1 = Xs[cur][wr * 64 + 32 + li][2 * ks + lk];
            float b0 = Ws[cur][wc * 64 + li][2 * ks + lk];
            float b1 = Ws[cur][wc * 64 + 32 + li][2 * ks + lk];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (kt + 1 < nkt) {
            lin_store_tile(Xs[cur ^ 1], tid, xv);
            lin_store_tile(Ws[cur ^ 1], tid, wv);
        }
        __syncthreads();
        cur ^= 1;
    }
    // epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        int n = n0 + wc * 64 + tn * 32 + li;
        if (n >= N) continue;
        float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int m = m0 + wr * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (m < M) {
                    float v = acc[tm][tn][r] + bv;
                    if (RELU) v = fmaxf(v, 0.f);
                    if (R) v += R[(int64_t)m * N + n];
                    Y[(int64_t)m * N + n] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------ attention
// One workgroup per (head, sequence).  K_h, V_h of the whole sequence staged in
// LDS; one query row per thread (L <= 256); online softmax with the additive
// mask evaluated in registers:
//   IRN:    j == L-1 -> +1.0 (every row sees the target), j <= i -> + r_u, else -inf
//   causal: j <= i -> 0, else -inf
//   padded key (seq[j] == 0) -> -inf
// A fully masked row yields NaN like torch's softmax over all -inf.
template <int HD>
__global__ void __launch_bounds__(256) k_attn(const float *__restrict__ qkv, const int64_t *__restrict__ seq,
                                              const float *__restrict__ r_u, float *__restrict__ out, int L, int d,
                                              int hd, int mask_mode) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *Ks = reinterpret_cast<float *>(smem);
    float *Vs = Ks + (size_t)L * HD;
    unsigned char *pad = reinterpret_cast<unsigned char *>(Vs + (size_t)L * HD);
    const int h = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x;
    const int64_t base = (int64_t)b * L;
    const int ld = 3 * d;
    for (int idx = tid; idx < L * HD; idx += 256) {
        int j = idx / HD, c = idx % HD;
        float kv = 0.f, vv = 0.f;
        if (c < hd) {
            const float *row = qkv + (base + j) * ld + h * hd + c;
            kv = row[d];
            vv = row[2 * d];
        }
        Ks[idx] = kv;
        Vs[idx] = vv;
    }
    for (int j = tid; j < L; j += 256) pad[j] = (seq[base + j] == 0) ? 1 : 0;
    __syncthreads();

    const int i = tid;
    const bool active = i < L;
    const float scale = 1.0f / sqrtf((float)hd);
    float q[HD], o[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) {
        q[c] = (active && c < hd) ? qkv[(base + i) * ld + h * hd + c] * scale : 0.f;
        o[c] = 0.f;
    }
    const float add_allowed = (mask_mode == IRS_MASK_IRN) ? r_u[b] : 0.f;
    float m = -INFINITY, l = 0.f;
    const int wave_first = tid & ~63;
    int jend = wave_first + 63;
    if (jend > L - 1) jend = L - 1;
    const bool irn = (mask_mode == IRS_MASK_IRN);
    // keys 0..jend (wave-uniform bound); in IRN mode key L-1 is handled after the loop
    int jloop_end = irn ? (jend < L - 2 ? jend : L - 2) : jend;
    if (wave_first < L) {
        for (int j = 0; j <= jloop_end; ++j) {
            const float4 *k4 = reinterpret_cast<const float4 *>(Ks + (size_t)j * HD);
            float s = 0.f;
#pragma unroll
            for (int c4 = 0; c4 < HD / 4; ++c4) {
                float4 kk = k4[c4];
                s = __fmaf_rn(q[4 * c4 + 0], kk.x, s);
                s = __fmaf_rn(q[4 * c4 + 1], kk.y, s);
                s = __fmaf_rn(q[4 * c4 + 2], kk.z, s);
                s = __fmaf_rn(q[4 * c4 + 3], kk.w, s);
            }
            bool valid = active && (j <= i) && !pad[j];
            if (valid) {
                s += add_allowed;
                if (s > m) {
                    float corr = __expf(m - s);
                    l *= corr;
#pragma unroll
                    for (int c = 0; c < HD; ++c) o[c] *= corr;
                    m = s;
                }
                float p = __expf(s - m);
                l += p;
                const float4 *v4 = reinterpret_cast<const float4 *>(Vs + (size_t)j * HD);
#pragma unroll
                for (int c4 = 0; c4 < HD / 4; ++c4) {
                    float4 vv = v4[c4];
                    o[4 * c4 + 0] = __fmaf_rn(p, vv.x, o[4 * c4 + 0]);
                    o[4 * c4 + 1] = __fmaf_rn(p, vv.y, o[4 * c4 + 1]);
                    o[4 * c4 + 2] = __fmaf_rn(p, vv.z, o[4 * c4 + 2]);
                    o[4 * c4 + 3] = __fmaf_rn(p, vv.w, o[4 * c4 + 3]);
                }
            }
        }
        if (irn) {
            const int j = L - 1;
            const float4 *k4 = reinterpret_cast<const float4 *>(Ks + (size_t)j * HD);
            float s = 0.f;
#pragma unroll
            for (int c4 = 0; c4 < HD / 4; ++c4) {
                float4 kk = k4[c4];
                s = __fmaf_rn(q[4 * c4 + 0], kk.x, s);
                s = __fmaf_rn(q[4 * c4 + 1], kk.y, s);
                s = __fmaf_rn(q[4 * c4 + 2], kk.z, s);
                s = __fmaf_rn(q[4 * c4 + 3], kk.w, s);
            }
            bool valid = active && !pad[j];
            if (valid) {
                s += 1.0f;
                if (s > m) {
                    float corr = __expf(m - s);
                    l *= corr;
#pragma unroll
                    for (int c = 0; c < HD; ++c) o[c] *= corr;
                    m = s;
                }
                float p = __expf(s - m);
                l += p;
                const float4 *v4 = reinterpret_cast<const float4 *>(Vs + (size_t)j * HD);
#pragma unroll
                for (int c4 = 0; c4 < HD / 4; ++c4) {
                    float4 vv = v4[c4];
                    o[4 * c4 + 0] = __fmaf_rn(p, vv.x, o[4 * c4 + 0]);
                    o[4 * c4 + 1] = __fmaf_rn(p, vv.y, o[4 * c4 + 1]);
                    o[4 * c4 + 2] = __fmaf_rn(p, vv.z, o[4 * c4 + 2]);
                    o[4 * c4 + 3] = __fmaf_rn(p, vv.w, o[4 * c4 + 3]);
                }
            }
        }
    }
    if (active) {
        float inv = 1.0f / l; // l == 0 (fully masked) -> inf * 0 = NaN, as torch
        float *orow = out + (base + i) * d + h * hd;
#pragma unroll
        for (int c = 0; c < HD; ++c)
            if (c < hd) orow[c] = o[c] * inv;
    }
}

// ------------------------------------------------------------------ layer norm
// y = LN(z; g1, b1); if (c) y = LN(y + c; g2, b2).  One wave per row, d <= 512.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__global__ void __launch_bounds__(256) k_ln(const float *__restrict__ z, const float *__restrict__ g1,
                                            const float *__restrict__ b1, const float *__restrict__ c,
                                            const float *__restrict__ g2, const float *__restrict__ b2,
                                            float *__restrict__ y, int rows, int d) {
    int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (row >= rows) return;
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
static int launch_linear(irs_ctx *ctx, const float *X, const float *W, const float *bias, const float *R, float *Y,
                         int M, int N, int K, bool relu, hipStream_t s) {
    dim3 grid((N + LIN_BN - 1) / LIN_BN, (M + LIN_BM - 1) / LIN_BM);
    bool vec = (K % 4 == 0) && ((((uintptr_t)X) & 15) == 0) && ((((uintptr_t)W) & 15) == 0);
    irs_prof_begin(ctx, IRS_PROF_LINEAR, s);
    if (vec) {
        if (relu) hipLaunchKernelGGL((k_linear<true, true>), grid, dim3(256), 0, s, X, W, bias, R, Y, M, N, K);
        else hipLaunchKernelGGL((k_linear<true, false>), grid, dim3(256), 0, s, X, W, bias, R, Y, M, N, K);
    } else {
        if (relu) hipLaunchKernelGGL((k_linear<false, true>), grid, dim3(256), 0, s, X, W, bias, R, Y, M, N, K);
        else hipLaunchKernelGGL((k_linear<false, false>), grid, dim3(256), 0, s, X, W, bias, R, Y, M, N, K);
    }
    irs_prof_end(ctx, IRS_PROF_LINEAR, s, 2.0 * M * (double)N * K,
                 4.0 * ((double)M * K + (double)N * K + (double)M * N * (R ? 2 : 1)));
    IRS_CHECK_HIP(ctx, hipGetLastError());
    return IRS_OK;
}

static int launch_attn(irs_ctx *ctx, const float *qkv, const int64_t *seq, const float *r_u, float *out, int B,
                       hipStream_t s) {
    const int L = ctx->dims.max_len, d = ctx->dims.d, H = ctx->dims.n_heads, hd = d / H;
    int HD = hd <= 8 ? 8 : hd <= 16 ? 16 : hd <= 32 ? 32 : 64;
    size_t lds = (size_t)2 * L * HD * sizeof(float) + ((L + 15) & ~15);
    dim3 grid(H, B);
    irs_prof_begin(ctx, IRS_PROF_ATTN, s);
    switch (HD) {
    case 8: hipLaunchKernelGGL((k_attn<8>), grid, dim3(256), lds, s, qkv, seq, r_u, out, L, d, hd, ctx->dims.mask_mode); break;
    case 16: hipLaunchKernelGGL((k_attn<16>), grid, dim3(256), lds, s, qkv, seq, r_u, out, L, d, hd, ctx->dims.mask_mode); break;
    case 32: hipLaunchKernelGGL((k_attn<32>), grid, dim3(256), lds, s, qkv, seq, r_u, out, L, d, hd, ctx->dims.mask_mode); break;
    default: hipLaunchKernelGGL((k_attn<64>), grid, dim3(256), lds, s, qkv, seq, r_u, out, L, d, hd, ctx->dims.mask_mode); break;
    }
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

int irs_launch_decode(irs_ctx *ctx, const int64_t *seq, const int64_t *user, int B, float *x_out, const int32_t *pos,
                      float *xrows, float *r_u_out, hipStream_t s) {
    const int L = ctx->dims.max_len, d = ctx->dims.d, F = ctx->dims.ffn_dim;
    const int rows = B * L;
    int rc;
    if ((rc = irs_launch_pif(ctx, user, B, ctx->act_ru, s)) != IRS_OK) return rc;
    if (r_u_out) IRS_CHECK_HIP(ctx, hipMemcpyAsync(r_u_out, ctx->act_ru, sizeof(float) * B, hipMemcpyDeviceToDevice, s));
    float *x = ctx->act_x, *y = ctx->act_y;
    hipLaunchKernelGGL(k_embed, dim3((rows + 3) / 4), dim3(256), 0, s, seq, ctx->item_emb, ctx->pe, x, rows, L, d,
                       sqrtf((float)d), ctx->dims.n_item);
    IRS_CHECK_HIP(ctx, hipGetLastError());
    for (int l = 0; l < ctx->dims.n_layers; ++l) {
        const irs_layer_w &w = ctx->layer[l];
        // qkv = x W_in^T + b_in
        if ((rc = launch_linear(ctx, x, w.sa_in_w, w.sa_in_b, nullptr, ctx->act_qkv, rows, 3 * d, d, false, s))) return rc;
        if ((rc = launch_attn(ctx, ctx->act_qkv, seq, ctx->act_ru, ctx->act_ao, B, s))) return rc;
        // y = x + ao W_o^T + b_o ; x = LN2(LN1(y) + c_l)
        if ((rc = launch_linear(ctx, ctx->act_ao, w.sa_out_w, w.sa_out_b, x, y, rows, d, d, false, s))) return rc;
        hipLaunchKernelGGL(k_ln, dim3((rows + 3) / 4), dim3(256), 0, s, y, w.n1_w, w.n1_b, ctx->c_l + (size_t)l * d,
                           w.n2_w, w.n2_b, x, rows, d);
        // h = relu(x W1^T + b1); y = x + h W2^T + b2; x = LN3(y)
        if ((rc = launch_linear(ctx, x, w.l1_w, w.l1_b, nullptr, ctx->act_h, rows, F, d, true, s))) return rc;
        if ((rc = launch_linear(ctx, ctx->act_h, w.l2_w, w.l2_b, x, y, rows, d, F, false, s))) return rc;
        hipLaunchKernelGGL(k_ln, dim3((rows + 3) / 4), dim3(256), 0, s, y, w.n3_w, w.n3_b, (const float *)nullptr,
                           (const float *)nullptr, (const float *)nullptr, x, rows, d);
        IRS_CHECK_HIP(ctx, hipGetLastError());
    }
    if (x_out) IRS_CHECK_HIP(ctx, hipMemcpyAsync(x_out, x, sizeof(float) * (size_t)rows * d, hipMemcpyDeviceToDevice, s));
    if (pos && xrows) {
        hipLaunchKernelGGL(k_gather_rows, dim3(B), dim3(64), 0, s, x, pos, xrows, B, L, d);
        IRS_CHECK_HIP(ctx, hipGetLastError());
    }
    return IRS_OK;
}
