// HBM-bound row kernels: LayerNorm fwd/bwd, embedding (+PE) fwd/bwd, casts, column sums and
// the fused label-smoothed cross entropy of run_batch.  fp32 statistics throughout.
#include "common.h"
#include "kernels.h"

namespace {

// ---------------------------------------------------------------- LayerNorm (eps 1e-5, torch semantics)
// one wave per row; E <= 64*16
constexpr int LN_MAXPT = 16;
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ y32,
                                                     bf16* __restrict__ y16, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int rows, int E) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (long)row * E;
    float v[LN_MAXPT];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXPT; ++i) {
        const int c = lane + i * 64;
        v[i] = c < E ? xr[c] : 0.f;
        s += v[i];
    }
    const float mu = wave_sum(s) / E;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXPT; ++i) {
        const int c = lane + i * 64;
        const float d = c < E ? v[i] - mu : 0.f;
        q += d * d;
    }
    const float rs = rsqrtf(wave_sum(q) / E + 1e-5f);
#pragma unroll
    for (int i = 0; i < LN_MAXPT; ++i) {
        const int c = lane + i * 64;
        if (c < E) {
            const float o = (v[i] - mu) * rs * gamma[c] + beta[c];
            if (y32) y32[(long)row * E + c] = o;
            if (y16) y16[(long)row * E + c] = (bf16)o;
        }
    }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

// backward: each block handles LN_ROWS rows (one wave per row, looping), accumulating dgamma/dbeta
// partials per thread-column, written to slab[block][2][E]; reduced by ln_bwd_reduce.
constexpr int LN_ROWS = 8;
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, float* __restrict__ dx32,
                                                     bf16* __restrict__ dx16, float drop_p, uint32_t seed, uint32_t site,
                                                     float* __restrict__ slab, int rows, int E) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float dg[LN_MAXPT], db[LN_MAXPT];
#pragma unroll
    for (int i = 0; i < LN_MAXPT; ++i) { dg[i] = 0.f; db[i] = 0.f; }
    const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    for (int rr = wave; rr < LN_ROWS; rr += 4) {
        const int row = blockIdx.x * LN_ROWS + rr;
        if (row >= rows) break;
        const float mu = mean[row], rs = rstd[row];
        float g[LN_MAXPT], xh[LN_MAXPT];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXPT; ++i) {
            const int c = lane + i * 64;
            if (c < E) {
                const float d = dy[(long)row * E + c];
                xh[i] = (x[(long)row * E + c] - mu) * rs;
                g[i] = d * gamma[c];
                dg[i] += d * xh[i];
                db[i] += d;
                s1 += g[i];
                s2 += g[i] * xh[i];
            } else { g[i] = 0.f; xh[i] = 0.f; }
        }
        s1 = wave_sum(s1) / E;
        s2 = wave_sum(s2) / E;
#pragma unroll
        for (int i = 0; i < LN_MAXPT; ++i) {
            const int c = lane + i * 64;
            if (c < E) {
                const float o = (g[i] - s1 - xh[i] * s2) * rs;
                const long idx = (long)row * E + c;
                if (dx32) dx32[idx] = o;
                if (dx16) {
                    float od = o;
                    if (drop_p > 0.f) od *= dropout_scale(seed, site, (uint32_t)idx, drop_p, inv_keep);
                    dx16[idx] = (bf16)od;
                }
            }
        }
    }
    __shared__ float red[4][2][64 * LN_MAXPT];
#pragma unroll
    for (int i = 0; i < LN_MAXPT; ++i) {
        const int c = lane + i * 64;
        if (c < E) { red[wave][0][c] = dg[i]; red[wave][1][c] = db[i]; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < E; c += 256) {
        slab[((long)blockIdx.x * 2 + 0) * E + c] = red[0][0][c] + red[1][0][c] + red[2][0][c] + red[3][0][c];
        slab[((long)blockIdx.x * 2 + 1) * E + c] = red[0][1][c] + red[1][1][c] + red[2][1][c] + red[3][1][c];
    }
}
// 64 columns x 4 block-lanes per workgroup; fixed-order -> deterministic
__global__ __launch_bounds__(256) void ln_bwd_reduce(const float* __restrict__ slab, int nblocks, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, int E) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
    __shared__ float red[4][64];
    float s = 0.f;
    if (c < 2 * E) {
        const int which = c / E, col = c % E;
        for (int b = part; b < nblocks; b += 4) s += slab[((long)b * 2 + which) * E + col];
    }
    red[part][threadIdx.x & 63] = s;
    __syncthreads();
    if (part == 0 && c < 2 * E) {
        const float t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        (c / E == 0 ? dgamma : dbeta)[c % E] = t;
    }
}

// ---------------------------------------------------------------- embedding + positional encoding
__global__ void embed_fwd_kernel(const int* __restrict__ tok, const float* __restrict__ table, const float* __restrict__ pe,
                                 float* __restrict__ y32, bf16* __restrict__ y16, int B, int L, int E,
                                 float drop_p, uint32_t seed, uint32_t site) {
    const long n = (long)B * L * E;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int e = (int)(i % E);
    const long row = i / E;                       // b*L + l
    const int l = (int)(row % L);
    float v = table[(long)tok[row] * E + e] + pe[(long)l * E + e];
    if (drop_p > 0.f) v *= dropout_scale(seed, site, (uint32_t)i, drop_p, 1.f / (1.f - drop_p));
    y32[i] = v;
    y16[i] = (bf16)v;
}
// one block per vocabulary row: deterministic sum over the (few hundred) token rows
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int* __restrict__ tok, const float* __restrict__ dy,
                                                        float* __restrict__ dtable, int rows, int E, int accumulate,
                                                        float drop_p, uint32_t seed, uint32_t site) {
    const int v = blockIdx.x;
    const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    // rows holding token v, gathered in order (deterministic sum); most vocabulary rows have none
    __shared__ int hit[1024]; __shared__ int nhit;
    float s[4] = {0.f, 0.f, 0.f, 0.f};                                  // E <= 1024 -> <= 4 columns per thread
    for (int r0 = 0; r0 < rows; r0 += 1024) {
        __syncthreads();
        if (threadIdx.x == 0) {
            int n = 0;
            const int r1 = r0 + 1024 < rows ? r0 + 1024 : rows;
            for (int r = r0; r < r1; ++r) if (tok[r] == v) hit[n++] = r;
            nhit = n;
        }
        __syncthreads();
        for (int h = 0; h < nhit; ++h) {
            const int r = hit[h];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = threadIdx.x + i * 256;
                if (e < E) {
                    float g = dy[(long)r * E + e];
                    if (drop_p > 0.f) g *= dropout_scale(seed, site, (uint32_t)((long)r * E + e), drop_p, inv_keep);
                    s[i] += g;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = threadIdx.x + i * 256;
        if (e < E) { if (accumulate) dtable[(long)v * E + e] += s[i]; else dtable[(long)v * E + e] = s[i]; }
    }
}

__global__ void cast_dropout_kernel(const float* __restrict__ x, bf16* __restrict__ y, long n, float drop_p,
                                    uint32_t seed, uint32_t site) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (i + j < n) {
            float v = x[i + j];
            if (drop_p > 0.f) v *= dropout_scale(seed, site, (uint32_t)(i + j), drop_p, inv_keep);
            y[i + j] = (bf16)v;
        }
}

// ---------------------------------------------------------------- column sums (bias gradients)
constexpr int CS_ROWS = 256;      // rows per block
__global__ __launch_bounds__(256) void colsum_kernel(const bf16* __restrict__ x, long ld, float* __restrict__ slab,
                                                     int rows, int cols) {
    // thread -> 8 columns; 256 threads cover up to (256/ (cols/8)) row lanes
    const int c8n = cols / 8;
    const int rl = threadIdx.x / c8n, c8 = threadIdx.x % c8n;
    const int nrl = 256 / c8n;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (rl < nrl) {
        const int r0 = blockIdx.x * CS_ROWS;
        for (int r = r0 + rl; r < r0 + CS_ROWS && r < rows; r += nrl) {
            const bf16x8 v = ld8(x + (long)r * ld + c8 * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
        }
    }
    __shared__ float red[256][9];
#pragma unroll
    for (int j = 0; j < 8; ++j) red[threadIdx.x][j] = acc[j];
    __syncthreads();
    for (int c = threadIdx.x; c < cols; c += 256) {
        float s = 0.f;
        for (int k = 0; k < nrl; ++k) s += red[k * c8n + c / 8][c % 8];
        slab[(long)blockIdx.x * cols + c] = s;
    }
}
__global__ void colsum_reduce(const float* __restrict__ slab, int nblocks, float* __restrict__ out, int cols, int out_cols) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= out_cols) return;
    float s = 0.f;
    for (int b = 0; b < nblocks; ++b) s += slab[(long)b * cols + c];
    out[c] = s;
}

// ---------------------------------------------------------------- label-smoothed CE
// reference: src/transformer_torch_trainer.py:64-84 -- q = onehot*(1-eps) + (1-onehot)*eps/C (note /C),
// loss_i = -sum_c q_c logp_c, masked mean over gold != -1; eps == 0 -> plain CE(ignore_index=-1).
// d loss / d logit = (sum_c q_c) * softmax - q, times 1/n_total.  One wave per row.
__global__ __launch_bounds__(256) void ls_ce_kernel(const float* __restrict__ logits, long ld, const int* __restrict__ gold,
                                                    int rows, int C, float eps, float inv_ntotal,
                                                    bf16* __restrict__ dlogits, float* __restrict__ row_loss,
                                                    int* __restrict__ row_correct) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* z = logits + (long)row * ld;
    const int g = gold[row];
    float mx = -3.4e38f; int amax = 0;
    for (int c = lane; c < C; c += 64) { const float v = z[c]; if (v > mx) { mx = v; amax = c; } }
    // wave arg-max, ties -> lowest index (torch max(1) returns the first maximal index on CPU)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(mx, o, 64); const int oa = __shfl_xor(amax, o, 64);
        if (om > mx || (om == mx && oa < amax)) { mx = om; amax = oa; }
    }
    float se = 0.f, sz = 0.f;
    for (int c = lane; c < C; c += 64) { se += __expf(z[c] - mx); sz += z[c]; }
    se = wave_sum(se); sz = wave_sum(sz);
    const float lse = mx + __logf(se);
    const bool valid = g >= 0;
    const float off = eps > 0.f ? eps / C : 0.f, on = eps > 0.f ? 1.f - eps : 1.f;
    const float qsum = on + (C - 1) * off;
    if (lane == 0) {
        float loss = 0.f;
        if (valid) {
            const float sum_logp = sz - C * lse;                         // sum_c logp_c
            const float lg = z[g] - lse;
            loss = -((on - off) * lg + off * sum_logp);
        }
        row_loss[row] = loss;
        row_correct[row] = (valid && amax == g) ? 1 : 0;
    }
    bf16* d = dlogits + (long)row * ld;
    for (int c = lane; c < ld; c += 64) {
        float v = 0.f;
        if (valid && c < C) {
            const float p = __expf(z[c] - lse);
            v = (qsum * p - (c == g ? on : off)) * inv_ntotal;
        }
        d[c] = (bf16)v;
    }
}
__global__ __launch_bounds__(256) void ls_ce_reduce(const float* __restrict__ row_loss, const int* __restrict__ row_correct,
                                                    int rows, float inv_ntotal, float* __restrict__ stats) {
    __shared__ float sl[256]; __shared__ int sc[256];
    float l = 0.f; int c = 0;
    for (int r = threadIdx.x; r < rows; r += 256) { l += row_loss[r]; c += row_correct[r]; }
    sl[threadIdx.x] = l; sc[threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { sl[threadIdx.x] += sl[threadIdx.x + o]; sc[threadIdx.x] += sc[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { stats[0] = sl[0] * inv_ntotal; stats[1] = (float)sc[0]; stats[2] = rintf(1.f / inv_ntotal); }
}

}  // namespace

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? 0 : (mk_set_error(__func__, "launch failed"), -1))

int mk_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y32, bf16* y16, float* mean,
                       float* rstd, int rows, int E, hipStream_t s) {
    if (E > 64 * LN_MAXPT) { mk_set_error("mk_layernorm_fwd", "d_model > 1024 unsupported"); return -1; }
    hipLaunchKernelGGL(ln_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, gamma, beta, y32, y16, mean, rstd, rows, E);
    return LAUNCH_OK();
}
long mk_layernorm_bwd_slab_floats(int rows, int E) { return (long)((rows + LN_ROWS - 1) / LN_ROWS) * 2 * E; }
int mk_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                       float* dx32, bf16* dx16, float drop_p, uint32_t seed, uint32_t site, float* dgamma, float* dbeta,
                       float* slab, int rows, int E, hipStream_t s) {
    if (E > 64 * LN_MAXPT) { mk_set_error("mk_layernorm_bwd", "d_model > 1024 unsupported"); return -1; }
    const int nb = (rows + LN_ROWS - 1) / LN_ROWS;
    hipLaunchKernelGGL(ln_bwd_kernel, dim3(nb), dim3(256), 0, s, dy, x, gamma, mean, rstd, dx32, dx16, drop_p, seed, site, slab, rows, E);
    hipLaunchKernelGGL(ln_bwd_reduce, dim3((2 * E + 63) / 64), dim3(256), 0, s, slab, nb, dgamma, dbeta, E);
    return LAUNCH_OK();
}
int mk_embed_fwd(const int* tok, const float* table, const float* pe, float* y32, bf16* y16, int B, int L, int E,
                   float drop_p, uint32_t seed, uint32_t site, hipStream_t s) {
    const long n = (long)B * L * E;
    hipLaunchKernelGGL(embed_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, tok, table, pe, y32, y16, B, L, E, drop_p, seed, site);
    return LAUNCH_OK();
}
int mk_embed_bwd(const int* tok, const float* dy, float* dtable, int rows, int V, int E, int accumulate, float drop_p,
                   uint32_t seed, uint32_t site, hipStream_t s) {
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(V), dim3(256), 0, s, tok, dy, dtable, rows, E, accumulate, drop_p, seed, site);
    return LAUNCH_OK();
}
int mk_cast_dropout(const float* x, bf16* y, long n, float drop_p, uint32_t seed, uint32_t site, hipStream_t s) {
    hipLaunchKernelGGL(cast_dropout_kernel, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, s, x, y, n, drop_p, seed, site);
    return LAUNCH_OK();
}
long mk_colsum_slab_floats(int rows, int cols) { return (long)((rows + CS_ROWS - 1) / CS_ROWS) * cols; }
int mk_colsum(const bf16* x, long ld, float* out, float* slab, int rows, int cols, int out_cols, hipStream_t s) {
    if ((cols & 7) || cols > 2048 || (ld & 7)) { mk_set_error("mk_colsum", "cols must be a multiple of 8, <= 2048"); return -1; }
    const int nb = (rows + CS_ROWS - 1) / CS_ROWS;
    hipLaunchKernelGGL(colsum_kernel, dim3(nb), dim3(256), 0, s, x, ld, slab, rows, cols);
    hipLaunchKernelGGL(colsum_reduce, dim3((cols + 255) / 256), dim3(256), 0, s, slab, nb, out, cols, out_cols);
    return LAUNCH_OK();
}
int mk_ls_ce(const float* logits, long ld, const int* gold, int rows, int C, float eps, float inv_ntotal, bf16* dlogits,
               float* row_loss, int* row_correct, float* stats, hipStream_t s) {
    hipLaunchKernelGGL(ls_ce_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, logits, ld, gold, rows, C, eps, inv_ntotal, dlogits, row_loss, row_correct);
    hipLaunchKernelGGL(ls_ce_reduce, dim3(1), dim3(256), 0, s, row_loss, row_correct, rows, inv_ntotal, stats);
    return LAUNCH_OK();
}
