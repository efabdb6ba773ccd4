// HBM-bound row kernels: LayerNorm fwd/bwd, embedding (+PE) fwd/bwd, casts, column sums and
// the fused label-smoothed cross entropy of run_batch.  fp32 statistics throughout.
#include "common.h"
#include "kernels.h"
#include "folds.h"

namespace {

// ---------------------------------------------------------------- LayerNorm (eps 1e-5, torch semantics)
// one wave per row, E = 64 * PT exactly (compile-time): every load of a row is issued unconditionally and up front --
// a bounds check around the loads makes hipcc serialise them on s_waitcnt vmcnt(0).
// These kernels are bound by the NUMBER of vector-memory wave-instructions (one texture addresser per CU, ~40-90 cycles each whatever
// their width): with 4-byte lanes a 512-wide row cost 8 instructions per stream (backward: 32 per row -> 9-11 us for 4000 rows, 3 us of
// bytes); a lane now owns 4 consecutive columns per 256-column group (16-byte loads / stores, 8-byte bf16 stores): 7 per row.
template <int VW> struct LnVec;
template <> struct LnVec<4> {
    static __device__ __forceinline__ void ld(const float* p, float (&o)[4]) { const f32x4 t = *reinterpret_cast<const f32x4*>(p); o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; o[3] = t[3]; }
    static __device__ __forceinline__ void st(float* p, const float (&o)[4]) { *reinterpret_cast<f32x4*>(p) = f32x4{o[0], o[1], o[2], o[3]}; }
    static __device__ __forceinline__ void st16(bf16* p, const float (&o)[4]) { bf16x4 t; t[0] = (bf16)o[0]; t[1] = (bf16)o[1]; t[2] = (bf16)o[2]; t[3] = (bf16)o[3]; *reinterpret_cast<bf16x4*>(p) = t; }
};
template <> struct LnVec<1> {
    static __device__ __forceinline__ void ld(const float* p, float (&o)[1]) { o[0] = p[0]; }
    static __device__ __forceinline__ void st(float* p, const float (&o)[1]) { p[0] = o[0]; }
    static __device__ __forceinline__ void st16(bf16* p, const float (&o)[1]) { p[0] = (bf16)o[0]; }
};
// dropout keep-scales of a lane's VW consecutive elements (one hash word per element pair: common.h)
template <int VW>
__device__ __forceinline__ void ln_keep_scales(uint32_t seed, uint32_t site, uint32_t idx0, float p, float inv_keep, float (&ks)[VW]) {
    if constexpr (VW == 4) {
        ks[0] = ks[1] = ks[2] = ks[3] = 1.f;
        if (p > 0.f) dropout_scale4(seed, site, idx0, p, inv_keep, ks);
    } else {
#pragma unroll
        for (int e = 0; e < VW; ++e) ks[e] = p > 0.f ? dropout_scale(seed, site, idx0 + e, p, inv_keep) : 1.f;
    }
}
// lane's columns: group i (of NV) holds columns (i * 64 + lane) * VW .. + VW - 1
// SUM (kernels.h LnSumArgs): the row is not in memory -- it is what the epilogue of a k-split GEMM would have written: the sum of sm.n fp32
// partial products (+ bias) (x dropout) + residual, formed here in the epilogue's order of operations and stored to sm.sum_out (the backward
// pass normalises it again).  The few-row GEMMs with a long reduction run as n x more workgroups with a short k chain each, and no combine pass.
template <int PT, bool SUM = false>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ y32,
                                                     bf16* __restrict__ y16, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int rows, const LnSumArgs sm) {
    constexpr int E = 64 * PT, VW = PT % 4 == 0 ? 4 : 1, NV = PT / VW;
    using V = LnVec<VW>;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (long)row * E;
    float v[NV][VW], gm[NV][VW], bt[NV][VW];
    float s = 0.f;
    if constexpr (SUM) {
        const uint32_t seed = sm.seed_ptr ? *sm.seed_ptr : sm.seed;
        const float inv_keep = sm.drop_p > 0.f ? 1.f / (1.f - sm.drop_p) : 1.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * VW;
            const long idx = (long)row * E + c;
            float acc[VW], t[VW], rsd[VW], bs[VW];
            V::ld(sm.part + idx, acc);
            for (int z = 1; z < sm.n; ++z) {
                V::ld(sm.part + z * sm.stride + idx, t);
#pragma unroll
                for (int e = 0; e < VW; ++e) acc[e] += t[e];
            }
            V::ld(sm.residual + idx, rsd);
#pragma unroll
            for (int e = 0; e < VW; ++e) bs[e] = 0.f;
            if (sm.bias) {                                    // (parameters sit unpadded in the flat buffer: a bias need not be 16-byte aligned)
#pragma unroll
                for (int e = 0; e < VW; ++e) bs[e] = sm.bias[c + e];
            }
            float ks[VW];
            ln_keep_scales<VW>(seed, sm.site, (uint32_t)idx, sm.drop_p, inv_keep, ks);
#pragma unroll
            for (int e = 0; e < VW; ++e) v[i][e] = (acc[e] + bs[e]) * ks[e] + rsd[e];
            if (sm.sum_out) V::st(sm.sum_out + idx, v[i]);
            V::ld(gamma + c, gm[i]); V::ld(beta + c, bt[i]);
        }
    } else {
#pragma unroll
    for (int i = 0; i < NV; ++i) { const int c = (i * 64 + lane) * VW; V::ld(xr + c, v[i]); V::ld(gamma + c, gm[i]); V::ld(beta + c, bt[i]); }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < VW; ++e) s += v[i][e];
    const float mu = wave_sum(s) / E;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < VW; ++e) { const float d = v[i][e] - mu; q += d * d; }
    const float rs = rsqrtf(wave_sum(q) / E + 1e-5f);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float o[VW];
#pragma unroll
        for (int e = 0; e < VW; ++e) o[e] = (v[i][e] - mu) * rs * gm[i][e] + bt[i][e];
        const long idx = (long)row * E + (i * 64 + lane) * VW;
        if (y32) V::st(y32 + idx, o);
        if (y16) V::st16(y16 + idx, o);
    }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

// backward: a workgroup handles 4 * R rows (a wave R consecutive rows, all of their loads issued up front), accumulating dgamma/dbeta
// partials per thread-column, written to slab[block][2][E]; reduced by ln_bwd_reduce.  R = 4 for the encoder's rows (250 workgroups for
// 4000 rows), 1 for the decoder's few hundred (148 workgroups instead of 37).
__host__ __device__ inline int ln_rows_per_wave(int rows) { return rows >= 2048 ? 4 : 1; }
// SUM: dy is the sum of sm.n partial products of a k-split dgrad GEMM + its residual gradient (see ln_fwd_kernel)
template <int PT, int R, bool SUM = false>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, float* dx32 /* SUM: may be sm.residual (in place) */,
                                                     bf16* __restrict__ dx16, float drop_p, uint32_t seed, uint32_t site,
                                                     float* __restrict__ slab, int rows, const uint32_t* __restrict__ seed_ptr, const LnSumArgs sm) {
    constexpr int E = 64 * PT, VW = PT % 4 == 0 ? 4 : 1, NV = PT / VW;
    using V = LnVec<VW>;
    if (seed_ptr) seed = *seed_ptr;                           // replayed (graph-captured) step: the seed of THIS step lives on the device
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float dg[NV][VW], db[NV][VW], gm[NV][VW];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        V::ld(gamma + (i * 64 + lane) * VW, gm[i]);
#pragma unroll
        for (int e = 0; e < VW; ++e) { dg[i][e] = 0.f; db[i][e] = 0.f; }
    }
    const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    const int row0 = (blockIdx.x * 4 + wave) * R;
    float d[R][NV][VW], xv[R][NV][VW], mu[R], rs[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = row0 + r < rows ? row0 + r : rows - 1;               // (clamped: loads stay unconditional)
        mu[r] = mean[row]; rs[r] = rstd[row];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const long idx = (long)row * E + (i * 64 + lane) * VW;
            if constexpr (SUM) {
                float t[VW];
                V::ld(sm.part + idx, d[r][i]);
                for (int z = 1; z < sm.n; ++z) {
                    V::ld(sm.part + z * sm.stride + idx, t);
#pragma unroll
                    for (int e = 0; e < VW; ++e) d[r][i][e] += t[e];
                }
                V::ld(sm.residual + idx, t);
#pragma unroll
                for (int e = 0; e < VW; ++e) d[r][i][e] += t[e];
            } else {
                V::ld(dy + idx, d[r][i]);
            }
            V::ld(x + idx, xv[r][i]);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = row0 + r;
        if (row >= rows) break;
        float g[NV][VW], xh[NV][VW];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < VW; ++e) {
                xh[i][e] = (xv[r][i][e] - mu[r]) * rs[r];
                g[i][e] = d[r][i][e] * gm[i][e];
                dg[i][e] += d[r][i][e] * xh[i][e];
                db[i][e] += d[r][i][e];
                s1 += g[i][e];
                s2 += g[i][e] * xh[i][e];
            }
        s1 = wave_sum(s1) / E;
        s2 = wave_sum(s2) / E;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float o[VW];
#pragma unroll
            for (int e = 0; e < VW; ++e) o[e] = (g[i][e] - s1 - xh[i][e] * s2) * rs[r];
            const long idx = (long)row * E + (i * 64 + lane) * VW;
            if (dx32) V::st(dx32 + idx, o);
            if (dx16) {
                if (drop_p > 0.f) {
                    float ks[VW];
                    ln_keep_scales<VW>(seed, site, (uint32_t)idx, drop_p, inv_keep, ks);
#pragma unroll
                    for (int e = 0; e < VW; ++e) o[e] *= ks[e];
                }
                V::st16(dx16 + idx, o);
            }
        }
    }
    __shared__ __attribute__((aligned(16))) float red[4][2][E];
#pragma unroll
    for (int i = 0; i < NV; ++i) { V::st(&red[wave][0][(i * 64 + lane) * VW], dg[i]); V::st(&red[wave][1][(i * 64 + lane) * VW], db[i]); }
    __syncthreads();
    for (int c = threadIdx.x; c < E; c += 256) {
        slab[((long)blockIdx.x * 2 + 0) * E + c] = red[0][0][c] + red[1][0][c] + red[2][0][c] + red[3][0][c];
        slab[((long)blockIdx.x * 2 + 1) * E + c] = red[0][1][c] + red[1][1][c] + red[2][1][c] + red[3][1][c];
    }
}
// 32 columns x 8 block-lanes per workgroup, 4 independent partial sums per thread; fixed order -> deterministic
__global__ __launch_bounds__(256) void ln_bwd_reduce(const float* __restrict__ slab, int nblocks, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, int E) {
    ln_bwd_reduce_body(slab, nblocks, dgamma, dbeta, E, blockIdx.x);
}
// all LayerNorms of a backward pass at once: blockIdx.y = which LayerNorm (descriptors in the kernel-argument segment)
__global__ __launch_bounds__(256) void ln_bwd_reduce_grouped(LnReduceGroup grp, int E) {
    const LnReduceDesc& d = grp.p[blockIdx.y];
    ln_bwd_reduce_body(d.slab, d.nblocks, d.dgamma, d.dbeta, E, blockIdx.x);
}

// ---------------------------------------------------------------- embedding + positional encoding
__global__ void embed_fwd_kernel(const int* __restrict__ tok, const float* __restrict__ table, const float* __restrict__ pe,
                                 float* __restrict__ y32, bf16* __restrict__ y16, int B, int L, int E,
                                 float drop_p, uint32_t seed, uint32_t site, const uint32_t* __restrict__ seed_ptr) {
    if (seed_ptr) seed = *seed_ptr;
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
// dtable[v][:] (+)= sum of dy rows whose token is v.  Workgroup = (vocabulary row v, 64-column slice).  The rows holding v come sorted from
// the host (masr_run_batch counts the step's tokens while it stages them: order[start[v] .. start[v + 1]) = their positions, ascending)
// and are summed by 4 waves (wave w takes hits w, w+4, ...; fixed order -> deterministic).  The padding token (</s>) has hundreds of
// hits, hence the 2-D split.  (Until round 4 every workgroup scanned all tokens and compacted its hits with ballots: 29 us for 640
// tokens against 25 now; 16 row loads in flight instead of 8 change nothing.)
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int* __restrict__ order, const int* __restrict__ start, const float* __restrict__ dy,
                                                        float* __restrict__ dtable, int E, int accumulate,
                                                        float drop_p, uint32_t seed, uint32_t site, const uint32_t* __restrict__ seed_ptr) {
    embed_bwd_body(order, start, dy, dtable, E, accumulate, drop_p, seed, site, seed_ptr, blockIdx.x, blockIdx.y);
}

__global__ void cast_dropout_kernel(const float* __restrict__ x, bf16* __restrict__ y, long n, float drop_p,
                                    uint32_t seed, uint32_t site, const uint32_t* __restrict__ seed_ptr) {
    if (seed_ptr) seed = *seed_ptr;
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    float ks[4] = {1.f, 1.f, 1.f, 1.f};
    if (drop_p > 0.f) dropout_scale4(seed, site, (uint32_t)i, drop_p, inv_keep, ks);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (i + j < n) y[i + j] = (bf16)(x[i + j] * ks[j]);
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

// ---------------------------------------------------------------- greedy decode helpers (MyTransformer.recog, :143-176)
// decoder input of step `L`: tok[b][0] = sos, tok[b][l] = previous step's output token at position l-1
__global__ void recog_build_tok_kernel(int* __restrict__ tok, const int* __restrict__ out, int B, int L, int sos) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * L) return;
    const int b = i / L, l = i % L;
    tok[i] = l == 0 ? sos : out[(l - 1) * B + b];
}
// out[l][b] = argmax_c logits[b*L + l][c]  (first maximal index, as torch.argmax); one wave per row
__global__ __launch_bounds__(256) void recog_argmax_kernel(const float* __restrict__ logits, long ld, int* __restrict__ out, int B, int L, int C) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B * L) return;
    const float* z = logits + (long)row * ld;
    float mx = -3.4e38f; int am = 0;
    for (int c = lane; c < C; c += 64) { const float v = z[c]; if (v > mx) { mx = v; am = c; } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(mx, o, 64); const int oa = __shfl_xor(am, o, 64);
        if (om > mx || (om == mx && oa < am)) { mx = om; am = oa; }
    }
    if (lane == 0) out[(row % L) * B + row / L] = am;
}

// ---------------------------------------------------------------- label-smoothed CE
// reference: src/transformer_torch_trainer.py:64-84 -- q = onehot*(1-eps) + (1-onehot)*eps/C (note /C),
// loss_i = -sum_c q_c logp_c, masked mean over gold != -1; eps == 0 -> plain CE(ignore_index=-1).
// d loss / d logit = (sum_c q_c) * softmax - q, times 1/n_total.  One wave per row.
__global__ __launch_bounds__(256) void ls_ce_kernel(const float* __restrict__ logits, long ld, const int* __restrict__ gold,
                                                    int rows, int C, float eps, float inv_ntotal,
                                                    bf16* __restrict__ dlogits, float* __restrict__ row_loss,
                                                    int* __restrict__ row_correct, const float* __restrict__ inv_ptr) {
    if (inv_ptr) inv_ntotal = *inv_ptr;                       // replayed step: 1 / n_total of THIS batch lives on the device
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
                                                    int rows, float inv_ntotal, float* __restrict__ stats, const float* __restrict__ inv_ptr) {
    if (inv_ptr) inv_ntotal = *inv_ptr;
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

template <int PT>
static void ln_fwd_launch(const float* x, const float* gamma, const float* beta, float* y32, bf16* y16, float* mean, float* rstd, int rows, hipStream_t s,
                          const LnSumArgs* sm = nullptr) {
    if (sm) hipLaunchKernelGGL((ln_fwd_kernel<PT, true>), dim3((rows + 3) / 4), dim3(256), 0, s, x, gamma, beta, y32, y16, mean, rstd, rows, *sm);
    else hipLaunchKernelGGL((ln_fwd_kernel<PT, false>), dim3((rows + 3) / 4), dim3(256), 0, s, x, gamma, beta, y32, y16, mean, rstd, rows, LnSumArgs{});
}
template <int PT>
static void ln_bwd_launch(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, float* dx32, bf16* dx16,
                          float drop_p, uint32_t seed, uint32_t site, float* slab, int rows, int nb, hipStream_t s, const uint32_t* seed_ptr,
                          const LnSumArgs* sm = nullptr) {
    if (sm)            // (few rows only: one row per wave)
        hipLaunchKernelGGL((ln_bwd_kernel<PT, 1, true>), dim3(nb), dim3(256), 0, s, dy, x, gamma, mean, rstd, dx32, dx16, drop_p, seed, site, slab, rows, seed_ptr, *sm);
    else if (ln_rows_per_wave(rows) == 4)
        hipLaunchKernelGGL((ln_bwd_kernel<PT, 4>), dim3(nb), dim3(256), 0, s, dy, x, gamma, mean, rstd, dx32, dx16, drop_p, seed, site, slab, rows, seed_ptr, LnSumArgs{});
    else
        hipLaunchKernelGGL((ln_bwd_kernel<PT, 1>), dim3(nb), dim3(256), 0, s, dy, x, gamma, mean, rstd, dx32, dx16, drop_p, seed, site, slab, rows, seed_ptr, LnSumArgs{});
}
#define LN_DISPATCH(E, CALL)                                                                      \
    switch ((E) / 64) {                                                                           \
        case 1: CALL(1); break; case 2: CALL(2); break; case 4: CALL(4); break; case 6: CALL(6); break;   \
        case 8: CALL(8); break; case 12: CALL(12); break; case 16: CALL(16); break;              \
        default: mk_set_error("layernorm", "d_model must be 64 x {1,2,4,6,8,12,16}"); return -1; \
    }
int mk_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y32, bf16* y16, float* mean,
                       float* rstd, int rows, int E, hipStream_t s) {
    if (E % 64) { mk_set_error("mk_layernorm_fwd", "d_model must be a multiple of 64"); return -1; }
#define CALL(P) ln_fwd_launch<P>(x, gamma, beta, y32, y16, mean, rstd, rows, s)
    LN_DISPATCH(E, CALL)
#undef CALL
    return LAUNCH_OK();
}
static bool ln_sum_ok(const LnSumArgs& sm, int rows, int E, const char* who) {
    if (!sm.part || !sm.residual || sm.n < 1 || sm.n > 8 || rows >= 2048 || (sm.stride & 3) || ((uintptr_t)sm.part & 15) || ((uintptr_t)sm.residual & 15) ||
        (sm.sum_out && ((uintptr_t)sm.sum_out & 15)) || E % 64) {
        mk_set_error(who, "k-split partial sums: 1..8 aligned partials + a residual, fewer than 2048 rows"); return false;
    }
    return true;
}
int mk_layernorm_fwd_sum(const LnSumArgs& sm, const float* gamma, const float* beta, float* y32, bf16* y16, float* mean, float* rstd, int rows, int E,
                         hipStream_t s) {
    if (!ln_sum_ok(sm, rows, E, "mk_layernorm_fwd_sum")) return -1;
    const float* x = nullptr;
#define CALL(P) ln_fwd_launch<P>(x, gamma, beta, y32, y16, mean, rstd, rows, s, &sm)
    LN_DISPATCH(E, CALL)
#undef CALL
    return LAUNCH_OK();
}
static int ln_bwd_blocks(int rows) { const int per = 4 * ln_rows_per_wave(rows); return (rows + per - 1) / per; }
int mk_layernorm_bwd_blocks(int rows) { return ln_bwd_blocks(rows); }
// capacity for ANY row count up to `rows` (the block count is not monotone in the rows: few rows take 4 per workgroup, many 16)
long mk_layernorm_bwd_slab_floats(int rows, int E) {
    const int small = ln_bwd_blocks(rows < 2047 ? rows : 2047), here = ln_bwd_blocks(rows);
    return (long)(small > here ? small : here) * 2 * E;
}
int mk_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                       float* dx32, bf16* dx16, float drop_p, uint32_t seed, uint32_t site, float* dgamma, float* dbeta,
                       float* slab, int rows, int E, hipStream_t s, const uint32_t* seed_ptr) {
    if (E % 64) { mk_set_error("mk_layernorm_bwd", "d_model must be a multiple of 64"); return -1; }
    const int nb = ln_bwd_blocks(rows);
#define CALL(P) ln_bwd_launch<P>(dy, x, gamma, mean, rstd, dx32, dx16, drop_p, seed, site, slab, rows, nb, s, seed_ptr)
    LN_DISPATCH(E, CALL)
#undef CALL
    if (dgamma) hipLaunchKernelGGL(ln_bwd_reduce, dim3((2 * E + 31) / 32), dim3(256), 0, s, slab, nb, dgamma, dbeta, E);
    return LAUNCH_OK();
}
// the same with dy = sum of the partial products of a k-split dgrad GEMM + its residual gradient; partials of dgamma / dbeta to `slab` only
int mk_layernorm_bwd_sum(const LnSumArgs& sm, const float* x, const float* gamma, const float* mean, const float* rstd, float* dx32, bf16* dx16,
                         float drop_p, uint32_t seed, uint32_t site, float* slab, int rows, int E, hipStream_t s, const uint32_t* seed_ptr) {
    if (!ln_sum_ok(sm, rows, E, "mk_layernorm_bwd_sum")) return -1;
    const int nb = (rows + 3) / 4;
    const float* dy = nullptr;
#define CALL(P) ln_bwd_launch<P>(dy, x, gamma, mean, rstd, dx32, dx16, drop_p, seed, site, slab, rows, nb, s, seed_ptr, &sm)
    LN_DISPATCH(E, CALL)
#undef CALL
    return LAUNCH_OK();
}
int mk_layernorm_bwd_reduce_grouped(const LnReduceGroup& grp, int E, hipStream_t s) {
    if (grp.n <= 0) return 0;
    hipLaunchKernelGGL(ln_bwd_reduce_grouped, dim3((2 * E + 31) / 32, grp.n), dim3(256), 0, s, grp, E);
    return LAUNCH_OK();
}
int mk_embed_fwd(const int* tok, const float* table, const float* pe, float* y32, bf16* y16, int B, int L, int E,
                   float drop_p, uint32_t seed, uint32_t site, hipStream_t s, const uint32_t* seed_ptr) {
    const long n = (long)B * L * E;
    hipLaunchKernelGGL(embed_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, tok, table, pe, y32, y16, B, L, E, drop_p, seed, site, seed_ptr);
    return LAUNCH_OK();
}
int mk_embed_bwd(const int* order, const int* start, const float* dy, float* dtable, int V, int E, int accumulate, float drop_p,
                   uint32_t seed, uint32_t site, hipStream_t s, const uint32_t* seed_ptr) {
    if (E % 64) { mk_set_error("mk_embed_bwd", "d_model must be a multiple of 64"); return -1; }
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(V, E / 64), dim3(256), 0, s, order, start, dy, dtable, E, accumulate, drop_p, seed, site, seed_ptr);
    return LAUNCH_OK();
}
int mk_recog_build_tok(int* tok, const int* out, int B, int L, int sos, hipStream_t s) {
    hipLaunchKernelGGL(recog_build_tok_kernel, dim3((B * L + 255) / 256), dim3(256), 0, s, tok, out, B, L, sos);
    return LAUNCH_OK();
}
int mk_recog_argmax(const float* logits, long ld, int* out, int B, int L, int C, hipStream_t s) {
    hipLaunchKernelGGL(recog_argmax_kernel, dim3((B * L + 3) / 4), dim3(256), 0, s, logits, ld, out, B, L, C);
    return LAUNCH_OK();
}
__global__ void dropout_mask_kernel(float* __restrict__ out, long n, float p, uint32_t seed, uint32_t site) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = dropout_scale(seed, site, (uint32_t)i, p, 1.0f / (1.0f - p));
}
int mk_dropout_mask(float* out, long n, float p, uint32_t seed, uint32_t site, hipStream_t s) {
    hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, out, n, p, seed, site);
    return LAUNCH_OK();
}
int mk_cast_dropout(const float* x, bf16* y, long n, float drop_p, uint32_t seed, uint32_t site, hipStream_t s, const uint32_t* seed_ptr) {
    hipLaunchKernelGGL(cast_dropout_kernel, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, s, x, y, n, drop_p, seed, site, seed_ptr);
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
               float* row_loss, int* row_correct, float* stats, hipStream_t s, const float* inv_ptr) {
    hipLaunchKernelGGL(ls_ce_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, logits, ld, gold, rows, C, eps, inv_ntotal, dlogits, row_loss, row_correct, inv_ptr);
    hipLaunchKernelGGL(ls_ce_reduce, dim3(1), dim3(256), 0, s, row_loss, row_correct, rows, inv_ntotal, stats, inv_ptr);
    return LAUNCH_OK();
}
