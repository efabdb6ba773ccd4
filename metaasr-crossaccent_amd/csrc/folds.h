// Bodies of the small "fold" passes that close a backward pass (slab reduces of the conv / conv1 / LayerNorm weight gradients, the
// un-permutation of vgg2enc's weight gradient, the embedding backward).  Each is a device function that takes its block coordinates as
// arguments: their own kernels call them with blockIdx (BLSTM engine, tests), and fold.hip's merged launch runs ALL of a transformer
// step's folds as ONE grid -- eight launches of 5-25 us each, mostly launch latency, were the tail of every step.
#pragma once
#include "common.h"

namespace {

// dw[co][ci][3][3] = sum over the per-workgroup partial slabs; 64 outputs x 4 slab-lanes per workgroup, four independent partial sums
// per thread (fixed order -> deterministic); the trailing COUT entries are the bias gradient.  A slab holds the weight part in the REGISTER order
// of conv3x3_wgrad2_kernel (conv.hip write_slab): per (64-ci slab cs, 64-co half ch) block of 36 864 floats, element
// ((tap * 4 + fm) * 256 + wave * 64 + lane) * 4 + r is output channel ch * 64 + fm * 16 + (lane >> 4) * 4 + r, input channel cs * 64 + wave * 16 + (lane & 15)
__device__ __forceinline__ void conv3x3_wgrad_reduce_body(const float* __restrict__ slab, int nsplit, float* __restrict__ dw,
                                                          float* __restrict__ db, int CIN, int COUT, int bx) {
    const int KTOT = 9 * CIN, NW = COUT * KTOT;
    const int cl = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int i = bx * 64 + cl;                       // slab element (register order), then NW + co for the bias
    __shared__ float red[4][64];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < NW + COUT) {
        const bool is_b = i >= NW;
        const float* p = is_b ? slab + (long)nsplit * NW + (i - NW) : slab + i;
        const long stride = is_b ? COUT : NW;
        int k = part;
        for (; k + 12 < nsplit; k += 16) {
            s0 += p[(long)k * stride]; s1 += p[(long)(k + 4) * stride]; s2 += p[(long)(k + 8) * stride]; s3 += p[(long)(k + 12) * stride];
        }
        for (; k < nsplit; k += 4) s0 += p[(long)k * stride];
    }
    red[part][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (part == 0 && i < NW + COUT) {
        const float t = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
        if (i >= NW) { if (db) db[i - NW] = t; }
        else {
            const int yb = i / 36864, rem = i % 36864, tf = rem >> 10, th = (rem & 1023) >> 2, r = rem & 3, lane = th & 63;
            const int cs = yb % (CIN / 64), ch = yb / (CIN / 64), tap = tf >> 2, fm = tf & 3;
            const int co = ch * 64 + fm * 16 + (lane >> 4) * 4 + r, ci = cs * 64 + (th >> 6) * 16 + (lane & 15);
            dw[((long)co * CIN + ci) * 9 + tap] = t;
        }
    }
}

__device__ __forceinline__ void conv1_wgrad_reduce_body(const float* __restrict__ slab, int nblocks, float* __restrict__ dw,
                                                        float* __restrict__ db, float* __restrict__ part_out, int bx, int by, int gy) {
    __shared__ float part[16][17];
    const int o = threadIdx.x & 15, l = threadIdx.x >> 4, i = bx * 16 + o;
    const int rows = (nblocks + gy - 1) / gy;
    const int b0 = by * rows, b1 = b0 + rows < nblocks ? b0 + rows : nblocks;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = b0 + l;
    for (; b + 48 < b1; b += 64) {                          // four independent chains keep 4 loads in flight per thread
        s0 += slab[(long)b * 640 + i]; s1 += slab[(long)(b + 16) * 640 + i];
        s2 += slab[(long)(b + 32) * 640 + i]; s3 += slab[(long)(b + 48) * 640 + i];
    }
    for (; b < b1; b += 16) s0 += slab[(long)b * 640 + i];
    part[l][o] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (threadIdx.x < 16) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += part[k][threadIdx.x];
        if (part_out) part_out[by * 640 + i] = s;
        else {
            const int co = i / 10, k = i % 10;
            if (k < 9) dw[co * 9 + k] = s; else db[co] = s;
        }
    }
}

// the two passes above (C1_RSPLIT = 16 row chunks, then their 16 partials) in ONE block per 16 columns, for slabs of at most 256 rows (the
// fused conv1 weight gradient writes one row per persistent workgroup): with <= 16 rows per chunk the first pass's thread l of chunk c holds
// exactly one row, so a chunk's partial is the sequential sum of its rows and the result the sequential sum of the chunks -- thread l sums
// chunk l here, thread 0..15 the sixteen partials: the same additions in the same order, bit for bit.
__device__ __forceinline__ void conv1_wgrad_reduce256_body(const float* __restrict__ slab, int nblocks, float* __restrict__ dw, float* __restrict__ db, int bx) {
    __shared__ float part[16][17];
    const int o = threadIdx.x & 15, l = threadIdx.x >> 4, i = bx * 16 + o;
    const int rows = (nblocks + 15) / 16;
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { const int b = l * rows + r; v[r] = (r < rows && b < nblocks) ? slab[(long)b * 640 + i] : 0.f; }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += v[r];
    part[l][o] = s;
    __syncthreads();
    if (threadIdx.x < 16) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += part[k][threadIdx.x];
        const int co = i / 10, k = i % 10;
        if (k < 9) dw[co * 9 + k] = t; else db[co] = t;
    }
}

__device__ __forceinline__ void ln_bwd_reduce_body(const float* __restrict__ slab, int nblocks, float* __restrict__ dgamma,
                                                   float* __restrict__ dbeta, int E, int bx) {
    const int cl = threadIdx.x & 31, part = threadIdx.x >> 5;
    const int c = bx * 32 + cl;
    __shared__ float red[8][32];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < 2 * E) {
        const float* p = slab + (long)(c / E) * E + c % E;
        int b = part;
        for (; b + 24 < nblocks; b += 32) {
            s0 += p[(long)b * 2 * E]; s1 += p[(long)(b + 8) * 2 * E]; s2 += p[(long)(b + 16) * 2 * E]; s3 += p[(long)(b + 24) * 2 * E];
        }
        for (; b < nblocks; b += 8) s0 += p[(long)b * 2 * E];
    }
    red[part][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (part == 0 && c < 2 * E) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][cl];
        (c / E == 0 ? dgamma : dbeta)[c % E] = t;
    }
}

__device__ __forceinline__ void vgg2enc_unpermute_body(const float* __restrict__ g, float* __restrict__ dw, int E, int C, int Dp, long bx) {
    const int F = C * Dp;
    const long n = (long)E * F;
    const long i = bx * 256 + threadIdx.x;
    if (i >= n) return;
    const int f = (int)(i % F), e = (int)(i / F);
    const int c = f / Dp, d = f % Dp;
    dw[i] = g[(long)e * F + d * C + c];
}

__device__ __forceinline__ void embed_bwd_body(const int* __restrict__ order, const int* __restrict__ start, const float* __restrict__ dy,
                                               float* __restrict__ dtable, int E, int accumulate,
                                               float drop_p, uint32_t seed, uint32_t site, const uint32_t* __restrict__ seed_ptr, int bx, int by) {
    if (seed_ptr) seed = *seed_ptr;
    const int v = bx, col = by * 64 + (threadIdx.x & 63);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    __shared__ float part[4][64];
    __shared__ int hit[1024];
    const int s0 = start[v], ntot = start[v + 1] - s0;
    if (ntot == 0) { if (!accumulate && wave == 0) dtable[(long)v * E + col] = 0.f; return; }
    float s = 0.f;
    for (int c0 = 0; c0 < ntot; c0 += 1024) {                          // (the hit list through LDS: the row loads must not wait for it one by one)
        const int n = ntot - c0 < 1024 ? ntot - c0 : 1024;
        if (c0) __syncthreads();
        for (int i = threadIdx.x; i < n; i += 256) hit[i] = order[s0 + c0 + i];
        __syncthreads();
        for (int h0 = wave; h0 < n; h0 += 32) {                         // 8 independent row loads in flight per wave
            float gv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int h = h0 + 4 * u;
                const int r = hit[h < n ? h : n - 1];
                float g = dy[(long)r * E + col];
                if (drop_p > 0.f) g *= dropout_scale(seed, site, (uint32_t)((long)r * E + col), drop_p, inv_keep);
                gv[u] = h < n ? g : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += gv[u];
        }
    }
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0) {
        const float t = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        if (accumulate) dtable[(long)v * E + col] += t; else dtable[(long)v * E + col] = t;
    }
}

}  // namespace
