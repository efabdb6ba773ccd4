// Fused multi-head attention forward / backward (flash-style, online softmax) for gfx950.
//
// Restates torch nn.MultiheadAttention as used by nn.TransformerEncoder/DecoderLayer in
// src/model/transformer_pytorch/mono_transformer_torch.py:74-98,200-203:
//   scores = q k^T / sqrt(hd); key-padding mask (encoder self, decoder cross) or causal mask
//   (decoder self; no target key-padding mask -- quirk Q7); softmax; dropout on the
//   probabilities; . v
// q/k/v are read straight out of the packed in_proj GEMM output (row b*T+t, head h at
// column h*hd), so no head split/merge copies exist.  Score matrices never reach HBM -- nor LDS:
//
// One workgroup = 4 waves = 64 query rows (fwd, dQ) or 64 key rows (dK/dV); K/V (or Q/dO) tiles of 64 rows live in LDS in their
// natural [row][dim] layout and serve both MFMA operand forms: 16-byte row reads (contraction over dim) and ds_read_b64_tr_b16
// transposing reads (contraction over the row index).  Every product is computed TRANSPOSED with respect to the textbook form, the
// tile from LDS as the A operand and the wave's own rows (registers) as B: the 16 x 16 accumulator of S^T = K Q^T then holds, in lane
// (i, g), keys 4g..4g+3 of query i -- and that IS the B-operand layout of the next product (O^T = V^T P^T, contraction over keys in
// the order 4g+e | 16+4g+e that frag_tr's transposing reads deliver), so the probabilities go from one MFMA to the next through
// registers.  (They used to travel through a per-wave LDS scratch: 16 two-byte writes, 4 reads and a barrier per key block, and
// softmax statistics folded over 16 lanes instead of in the lane.)
#include "common.h"
#include "kernels.h"

namespace {

constexpr int BLK = 64;             // rows per tile
constexpr float NEG = -1e30f;
constexpr float LOG2E = 1.44269504088896341f;

template <int HD> struct Cfg {
    static constexpr int HDP = HD < 32 ? 32 : HD;   // contraction length for q.k (zero padded)
    static constexpr int KS = HDP / 32;
    static constexpr int DT = HD / 16;
    static constexpr int LD = HDP + 16;             // tile row stride: (LD*2 B) == 8 dwords mod 64 for hd 64
};

typedef __attribute__((address_space(3))) bf16x4 lds_b4;

// fragment with contraction over the tile's ROW index: rows krow0..krow0+31, columns c0..c0+15 (lane i = column c0 + i).
// k order inside the fragment is permuted: element e of lane group g is row 4g+e (e < 4) | 16+4g+(e-4).
template <int LD>
__device__ __forceinline__ bf16x8 frag_tr(const bf16* tile, int krow0, int c0, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const bf16* a0 = tile + (krow0 + 4 * g + q) * LD + c0 + 4 * p;
    const bf16* a1 = a0 + 16 * LD;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)a0);
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)a1);
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}
// the same k order from two accumulator tiles (rows 4g+r of the 16-row tiles 2m and 2m+1): the B operand of the follow-up product
__device__ __forceinline__ bf16x8 frag_acc(const f32x4& t0, const f32x4& t1) {
    bf16x8 f;
    f[0] = (bf16)t0[0]; f[1] = (bf16)t0[1]; f[2] = (bf16)t0[2]; f[3] = (bf16)t0[3];
    f[4] = (bf16)t1[0]; f[5] = (bf16)t1[1]; f[6] = (bf16)t1[2]; f[7] = (bf16)t1[3];
    return f;
}
// row-read fragment (contraction over dim): rows r0..r0+15, dims ks*32 + 8g..
template <int LD>
__device__ __forceinline__ bf16x8 frag_row(const bf16* tile, int r0, int ks, int lane) {
    return ld8(tile + (r0 + (lane & 15)) * LD + ks * 32 + (lane >> 4) * 8);
}
// fragment of 16 rows straight from global memory (rows beyond nrows / dims beyond HD are zero)
template <int HD>
__device__ __forceinline__ bf16x8 frag_global(const bf16* base, long ld, int row, int nrows, int ks, int lane) {
    const int r = row + (lane & 15), dcol = ks * 32 + (lane >> 4) * 8;
    const bool ok = r < nrows && dcol < HD;
    const bf16x8 v = ld8(base + (long)(r < nrows ? r : nrows - 1) * ld + (dcol < HD ? dcol : 0));
    return ok ? v : zero8();
}

// A [64][HD] tile (rows row0.. of a [nrows] matrix) on its way into LDS: fetch() issues the loads (unconditionally, row
// clamped: a branch around a global load costs an s_waitcnt vmcnt(0)), commit() stores them, zero-filling missing rows.
// The loops below fetch tile i+2 right after committing tile i+1, so the memory round trip runs under a tile's MFMAs
// (fetch-then-use inside one iteration exposed it once per key/query block: 4 times for the 250-frame encoder).
template <int HD>
struct TileRegs {
    static constexpr int CPR = HD / 8;                 // chunks per row
    static constexpr int N = (BLK * CPR + 255) / 256;  // chunks per thread
    bf16x8 v[N];
    int row0 = 0;
    __device__ __forceinline__ void fetch(const bf16* src, long ld, int r0, int nrows, int tid) {
        row0 = r0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int c = tid + i * 256;
            const int r = (c / CPR) % BLK, dc = (c % CPR) * 8;
            const int row = r0 + r < nrows ? r0 + r : nrows - 1;
            v[i] = ld8(src + (long)row * ld + dc);
        }
    }
    __device__ __forceinline__ void commit(bf16* dst, int nrows, int tid) const {
        constexpr int LD = Cfg<HD>::LD;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int c = tid + i * 256;
            const int r = c / CPR, dc = (c % CPR) * 8;
            if (c < BLK * CPR) st8(dst + r * LD + dc, row0 + r < nrows ? v[i] : zero8());
        }
    }
};
// only head dims < 32 have padding columns (dims HD..31) that must read as zero
template <int HD>
__device__ __forceinline__ void zero_tile(bf16* dst, int tid) {
    constexpr int LD = Cfg<HD>::LD;
    if constexpr (Cfg<HD>::HDP != HD) {
        for (int c = tid; c < BLK * LD / 8; c += 256) st8(dst + c * 8, zero8());
    }
}

// Write a wave's TRANSPOSED 16 x HD result (lane (i, g): row i, columns dt*16 + 4g .. + 3 in acc[dt]) -- 8 bytes per lane and
// column tile, a row's 32-byte runs side by side
template <int HD>
__device__ __forceinline__ void store_rows_t(const f32x4 (&acc)[HD / 16], float mul, bf16* dst, long ld, int row0, int nrows, int lane) {
    const int row = row0 + (lane & 15);
    if (row >= nrows) return;
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16)(acc[dt][r] * mul);
        *reinterpret_cast<bf16x4*>(dst + (long)row * ld + dt * 16 + (lane >> 4) * 4) = o;
    }
}
__device__ __forceinline__ float fold_groups_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float fold_groups_sum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }

// ---------------------------------------------------------------------------- forward
template <int HD>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a) {
    using C = Cfg<HD>;
    constexpr int LD = C::LD, KS = C::KS, DT = C::DT;
    __shared__ __attribute__((aligned(16))) bf16 sK[2][BLK * LD];      // double-buffered: ONE barrier per key block
    __shared__ __attribute__((aligned(16))) bf16 sV[2][BLK * LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * BLK;
    const int Tq = a.Tq, Tk = a.Tk;
    const int klen = a.klens ? a.klens[b] : Tk;
    const float scale = rsqrtf((float)HD);
    const float inv_keep = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    const uint32_t dseed = (a.drop_p > 0.f && a.seed_ptr) ? *a.seed_ptr : a.seed;

    const bf16* qb = a.q + (long)b * Tq * a.ldq + h * HD;
    const bf16* kb_ = a.k + (long)b * Tk * a.ldk + h * HD;
    const bf16* vb = a.v + (long)b * Tk * a.ldv + h * HD;

    zero_tile<HD>(sK[0], tid); zero_tile<HD>(sK[1], tid);
    zero_tile<HD>(sV[0], tid); zero_tile<HD>(sV[1], tid);

    const int qrow0 = q0 + wave * 16, qi = qrow0 + (lane & 15), g4 = (lane >> 4) * 4;
    bf16x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = frag_global<HD>(qb, a.ldq, qrow0, Tq, ks, lane);

    float m = NEG, l = 0.f;                                 // of query qi (the same in the four lane groups); m in units of q.k
    const float sc2 = scale * LOG2E;
    const uint32_t dbase = (uint32_t)((((long)b * a.H + h) * Tq + qi) * Tk);      // dropout index of (qi, key 0)
    f32x4 o[DT];                                            // O^T: dims dt*16 + 4g + r of query qi
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    int nkb = (Tk + BLK - 1) / BLK;
    if (a.causal) { const int lim = (q0 + BLK - 1) / BLK + 1; if (lim < nkb) nkb = lim; }

    TileRegs<HD> tk, tv;
    tk.fetch(kb_, a.ldk, 0, Tk, tid);
    tv.fetch(vb, a.ldv, 0, Tk, tid);
    __syncthreads();                                        // (the zero fill)
    tk.commit(sK[0], Tk, tid);
    tv.commit(sV[0], Tk, tid);
    if (nkb > 1) { tk.fetch(kb_, a.ldk, BLK, Tk, tid); tv.fetch(vb, a.ldv, BLK, Tk, tid); }
    __syncthreads();
    for (int kb = 0; kb < nkb; ++kb) {
        const bf16* cK = sK[kb & 1]; const bf16* cV = sV[kb & 1];
        f32x4 s[4];                                         // S^T: keys jt*16 + 4g + r of query qi
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            s[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) s[jt] = mma16(frag_row<LD>(cK, jt * 16, ks, lane), qf[ks], s[jt]);
        }
        // softmax in base 2 on the RAW scores (m = running maximum of q.k, sc2 = scale * log2 e): p = exp2((s - m) sc2) -- the difference
        // FIRST: s sc2 - fl(m sc2) is off by half an ulp of m sc2, which is a factor of 2^16 at scores of 1e9 (a diverging run must
        // still produce the one-hot softmax torch does).  The backward forms fl(s scale) - lse with lse = fl(m scale) + log l: the SAME rounded
        // product on both sides (__fmul_rn: never contracted into an FMA), exact at the maximum.
        // A block without a masked element (wave-uniform test) skips the per-element index arithmetic and compares.
        const bool full = kb * BLK + BLK <= klen && (!a.causal || kb * BLK + BLK - 1 <= qrow0);
        float mx = NEG;
        if (!full) {
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kj = kb * BLK + jt * 16 + g4 + r;
                    if (kj >= klen || (a.causal && kj > qi)) s[jt][r] = NEG;
                }
        }
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[jt][r]);
        const float mn = fmaxf(m, fold_groups_max(mx));
        const float alpha = __builtin_amdgcn_exp2f((m - mn) * sc2);
        m = mn;
        float rs = 0.f;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            float k4[4] = {1.f, 1.f, 1.f, 1.f};                   // keep-scales of this lane's four consecutive keys (one hash word per pair)
            if (a.drop_p > 0.f) dropout_scale4(dseed, a.site, dbase + (uint32_t)(kb * BLK + jt * 16 + g4), a.drop_p, inv_keep, k4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float p = __builtin_amdgcn_exp2f((s[jt][r] - m) * sc2);
                if (!full && s[jt][r] <= NEG) p = 0.f;
                rs += p;
                s[jt][r] = p * k4[r];
            }
        }
        l = l * alpha + fold_groups_sum(rs);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[dt][r] *= alpha;
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2) {
            const bf16x8 pf = frag_acc(s[2 * m2], s[2 * m2 + 1]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) o[dt] = mma16(frag_tr<LD>(cV, m2 * 32, dt * 16, lane), pf, o[dt]);
        }
        if (kb + 1 < nkb) {
            tk.commit(sK[(kb + 1) & 1], Tk, tid);           // (that buffer was last read in iteration kb-1: behind its barrier)
            tv.commit(sV[(kb + 1) & 1], Tk, tid);
            if (kb + 2 < nkb) { tk.fetch(kb_, a.ldk, (kb + 2) * BLK, Tk, tid); tv.fetch(vb, a.ldv, (kb + 2) * BLK, Tk, tid); }
            __syncthreads();
        }
    }
    if (qi < Tq && lane < 16) a.lse[((long)b * a.H + h) * Tq + qi] = __fmul_rn(m, scale) + __logf(l);
    store_rows_t<HD>(o, 1.f / l, a.o + (long)b * Tq * a.ldo + h * HD, a.ldo, qrow0, Tq, lane);
}

// ---------------------------------------------------------------------------- backward
// delta_i = rowsum(dO_i * O_i) is not a separate pass: the dQ workgroups take it from the dO / O fragments they hold anyway,
// the dK/dV workgroups compute it for each query block while that block's tiles are in flight (two 16-byte loads per thread).
// LDS of the two bodies of attn_bwd_kernel is ONE buffer (a workgroup runs one body or the other): 2 x 2 tiles (+ lse / delta)
template <int HD> struct BwdSmem {
    static constexpr int LD = Cfg<HD>::LD;
    static constexpr size_t TILE = sizeof(bf16) * BLK * LD;
    static constexpr size_t BYTES = 4 * TILE + sizeof(float) * 4 * BLK;
};
template <int HD>
__device__ __forceinline__ void attn_bwd_dq_body(const AttnArgs& a, const int bx, char* smem) {
    using C = Cfg<HD>;
    constexpr int LD = C::LD, KS = C::KS, DT = C::DT;
    bf16* sKb[2] = {reinterpret_cast<bf16*>(smem), reinterpret_cast<bf16*>(smem) + 2 * BLK * LD};
    bf16* sVb[2] = {sKb[0] + BLK * LD, sKb[1] + BLK * LD};

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, h = blockIdx.y, q0 = bx * BLK;
    const int Tq = a.Tq, Tk = a.Tk;
    const int klen = a.klens ? a.klens[b] : Tk;
    const float scale = rsqrtf((float)HD);
    const float inv_keep = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    const uint32_t dseed = (a.drop_p > 0.f && a.seed_ptr) ? *a.seed_ptr : a.seed;
    const bf16* qb = a.q + (long)b * Tq * a.ldq + h * HD;
    const bf16* kb_ = a.k + (long)b * Tk * a.ldk + h * HD;
    const bf16* vb = a.v + (long)b * Tk * a.ldv + h * HD;
    const bf16* dob = a.dout + (long)b * Tq * a.lddo + h * HD;
    const bf16* ob = a.o + (long)b * Tq * a.ldo + h * HD;

    zero_tile<HD>(sKb[0], tid); zero_tile<HD>(sKb[1], tid);
    zero_tile<HD>(sVb[0], tid); zero_tile<HD>(sVb[1], tid);
    const int qrow0 = q0 + wave * 16, qi = qrow0 + (lane & 15), g4 = (lane >> 4) * 4;
    bf16x8 qf[KS], dof[KS];
    float dl = 0.f;                                         // rowsum(dO * O) of query qi: this lane's share, then folded over the groups
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        qf[ks] = frag_global<HD>(qb, a.ldq, qrow0, Tq, ks, lane);
        dof[ks] = frag_global<HD>(dob, a.lddo, qrow0, Tq, ks, lane);
        const bf16x8 of = frag_global<HD>(ob, a.ldo, qrow0, Tq, ks, lane);
#pragma unroll
        for (int j = 0; j < 8; ++j) dl = fmaf((float)of[j], (float)dof[ks][j], dl);
    }
    dl = fold_groups_sum(dl);
    const float lse = qi < Tq ? a.lse[((long)b * a.H + h) * Tq + qi] : 0.f;
    const uint32_t dbase = (uint32_t)((((long)b * a.H + h) * Tq + qi) * Tk);      // dropout index of (qi, key 0)
    f32x4 dq[DT];                                           // dQ^T: dims dt*16 + 4g + r of query qi
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    int nkb = (Tk + BLK - 1) / BLK;
    if (a.causal) { const int lim = (q0 + BLK - 1) / BLK + 1; if (lim < nkb) nkb = lim; }
    TileRegs<HD> tk, tv;
    tk.fetch(kb_, a.ldk, 0, Tk, tid);
    tv.fetch(vb, a.ldv, 0, Tk, tid);
    __syncthreads();
    tk.commit(sKb[0], Tk, tid);
    tv.commit(sVb[0], Tk, tid);
    if (nkb > 1) { tk.fetch(kb_, a.ldk, BLK, Tk, tid); tv.fetch(vb, a.ldv, BLK, Tk, tid); }
    __syncthreads();
    for (int kb = 0; kb < nkb; ++kb) {
        const bf16* cK = sKb[kb & 1]; const bf16* cV = sVb[kb & 1];
        const bool full = kb * BLK + BLK <= klen && (!a.causal || kb * BLK + BLK - 1 <= qrow0) && qrow0 + 16 <= Tq;   // nothing masked (wave-uniform)
        // two halves of 32 keys: dS^T (keys jt*16 + 4g + r of query qi) of a half lives in 8 registers between its two products
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2) {
            f32x4 ds[2];
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) {
                const int jt = 2 * m2 + j2;
                f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    s = mma16(frag_row<LD>(cK, jt * 16, ks, lane), qf[ks], s);
                    dp = mma16(frag_row<LD>(cV, jt * 16, ks, lane), dof[ks], dp);
                }
                float k4[4] = {1.f, 1.f, 1.f, 1.f};
                if (a.drop_p > 0.f) dropout_scale4(dseed, a.site, dbase + (uint32_t)(kb * BLK + jt * 16 + g4), a.drop_p, inv_keep, k4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float p = __builtin_amdgcn_exp2f((__fmul_rn(s[r], scale) - lse) * LOG2E);
                    if (!full) {
                        const int kj = kb * BLK + jt * 16 + g4 + r;
                        if (kj >= klen || (a.causal && kj > qi) || qi >= Tq) p = 0.f;
                    }
                    ds[j2][r] = p * (dp[r] * k4[r] - dl) * scale;
                }
            }
            const bf16x8 df = frag_acc(ds[0], ds[1]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) dq[dt] = mma16(frag_tr<LD>(cK, m2 * 32, dt * 16, lane), df, dq[dt]);
        }
        if (kb + 1 < nkb) {
            tk.commit(sKb[(kb + 1) & 1], Tk, tid);
            tv.commit(sVb[(kb + 1) & 1], Tk, tid);
            if (kb + 2 < nkb) { tk.fetch(kb_, a.ldk, (kb + 2) * BLK, Tk, tid); tv.fetch(vb, a.ldv, (kb + 2) * BLK, Tk, tid); }
            __syncthreads();
        }
    }
    store_rows_t<HD>(dq, 1.f, a.dq + (long)b * Tq * a.lddq + h * HD, a.lddq, qrow0, Tq, lane);
}

template <int HD>
__device__ __forceinline__ void attn_bwd_dkv_body(const AttnArgs& a, const int bx, char* smem) {
    using C = Cfg<HD>;
    constexpr int LD = C::LD, KS = C::KS, DT = C::DT;
    bf16* sQb[2] = {reinterpret_cast<bf16*>(smem), reinterpret_cast<bf16*>(smem) + 2 * BLK * LD};
    bf16* sDOb[2] = {sQb[0] + BLK * LD, sQb[1] + BLK * LD};
    float* sStat = reinterpret_cast<float*>(sQb[0] + 4 * BLK * LD);      // [2][lse | delta][BLK]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, h = blockIdx.y, k0 = bx * BLK;
    const int Tq = a.Tq, Tk = a.Tk;
    const int klen = a.klens ? a.klens[b] : Tk;
    const float scale = rsqrtf((float)HD);
    const float inv_keep = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    const uint32_t dseed = (a.drop_p > 0.f && a.seed_ptr) ? *a.seed_ptr : a.seed;
    const bf16* qb = a.q + (long)b * Tq * a.ldq + h * HD;
    const bf16* kb_ = a.k + (long)b * Tk * a.ldk + h * HD;
    const bf16* vb = a.v + (long)b * Tk * a.ldv + h * HD;
    const bf16* dob = a.dout + (long)b * Tq * a.lddo + h * HD;
    const bf16* ob = a.o + (long)b * Tq * a.ldo + h * HD;

    zero_tile<HD>(sQb[0], tid); zero_tile<HD>(sQb[1], tid);
    zero_tile<HD>(sDOb[0], tid); zero_tile<HD>(sDOb[1], tid);
    const int krow0 = k0 + wave * 16, kj = krow0 + (lane & 15), g4 = (lane >> 4) * 4;
    bf16x8 kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        kf[ks] = frag_global<HD>(kb_, a.ldk, krow0, Tk, ks, lane);
        vf[ks] = frag_global<HD>(vb, a.ldv, krow0, Tk, ks, lane);
    }
    const uint32_t dbase = (uint32_t)(((long)b * a.H + h) * Tq * Tk + kj);        // dropout index of (query 0, kj)
    f32x4 dk[DT], dv[DT];                                   // dK^T / dV^T: dims dt*16 + 4g + r of key kj
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    const int nqb = (Tq + BLK - 1) / BLK;
    const int qb0 = a.causal ? k0 / BLK : 0;
    TileRegs<HD> tq, tdo;
    float nlse = 0.f, ndl = 0.f;       // lse (threads 0..63) / delta (threads 4r..4r+3: row r) of the next query block
    auto fetch_q = [&](int qbi) {
        tq.fetch(qb, a.ldq, qbi * BLK, Tq, tid);
        tdo.fetch(dob, a.lddo, qbi * BLK, Tq, tid);
        if (tid < BLK) {
            const int qi = qbi * BLK + tid;
            const long idx = ((long)b * a.H + h) * Tq + (qi < Tq ? qi : Tq - 1);
            const float l0 = a.lse[idx];
            nlse = qi < Tq ? l0 : 0.f;
        }
        {   // delta of row tid/4: the row's HD/8 16-byte chunks are dealt to its 4 threads, then folded over those lanes
            constexpr int CPR = HD / 8;
            const int qi = qbi * BLK + (tid >> 2), row = qi < Tq ? qi : Tq - 1;
            float d0 = 0.f;
#pragma unroll
            for (int cc = 0; cc < (CPR + 3) / 4; ++cc) {
                const int ch = (tid & 3) + 4 * cc, chc = ch < CPR ? ch : 0;
                const bf16x8 x = ld8(ob + (long)row * a.ldo + chc * 8), y = ld8(dob + (long)row * a.lddo + chc * 8);
                float t = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) t = fmaf((float)x[j], (float)y[j], t);
                d0 += ch < CPR ? t : 0.f;
            }
            d0 += __shfl_xor(d0, 1, 64); d0 += __shfl_xor(d0, 2, 64);
            ndl = qi < Tq ? d0 : 0.f;
        }
    };
    auto commit_q = [&](int buf) {
        tq.commit(sQb[buf], Tq, tid);
        tdo.commit(sDOb[buf], Tq, tid);
        if (tid < BLK) sStat[buf * 2 * BLK + tid] = nlse;
        if ((tid & 3) == 0) sStat[buf * 2 * BLK + BLK + (tid >> 2)] = ndl;
    };
    if (qb0 < nqb) {
        fetch_q(qb0);
        __syncthreads();
        commit_q(0);
        if (qb0 + 1 < nqb) fetch_q(qb0 + 1);
        __syncthreads();
    }
    for (int qbi = qb0; qbi < nqb; ++qbi) {
        const int cur = (qbi - qb0) & 1;
        const bf16* cQ = sQb[cur]; const bf16* cDO = sDOb[cur];
        const float* cLse = sStat + cur * 2 * BLK; const float* cDl = cLse + BLK;
        const bool full = krow0 + 16 <= klen && qbi * BLK + BLK <= Tq && (!a.causal || krow0 + 15 <= qbi * BLK);      // nothing masked (wave-uniform)
        // two halves of 32 queries: P / dS (queries jt*16 + 4g + r of key kj) of a half live in 16 registers between the products
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2) {
            f32x4 pt[2], dst[2];
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) {
                const int jt = 2 * m2 + j2;
                f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f}, dpt = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    st = mma16(frag_row<LD>(cQ, jt * 16, ks, lane), kf[ks], st);
                    dpt = mma16(frag_row<LD>(cDO, jt * 16, ks, lane), vf[ks], dpt);
                }
                const f32x4 ls4 = *reinterpret_cast<const f32x4*>(cLse + jt * 16 + g4);
                const f32x4 dl4 = *reinterpret_cast<const f32x4*>(cDl + jt * 16 + g4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qi = qbi * BLK + jt * 16 + g4 + r;
                    float p = __builtin_amdgcn_exp2f((__fmul_rn(st[r], scale) - ls4[r]) * LOG2E);
                    if (!full && (kj >= klen || (a.causal && kj > qi) || qi >= Tq)) p = 0.f;
                    float ms = 1.f;
                    if (a.drop_p > 0.f) ms = dropout_scale(dseed, a.site, dbase + (uint32_t)qi * (uint32_t)Tk, a.drop_p, inv_keep);
                    pt[j2][r] = p * ms;
                    dst[j2][r] = p * (dpt[r] * ms - dl4[r]) * scale;
                }
            }
            const bf16x8 pf = frag_acc(pt[0], pt[1]);
            const bf16x8 df = frag_acc(dst[0], dst[1]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                dv[dt] = mma16(frag_tr<LD>(cDO, m2 * 32, dt * 16, lane), pf, dv[dt]);
                dk[dt] = mma16(frag_tr<LD>(cQ, m2 * 32, dt * 16, lane), df, dk[dt]);
            }
        }
        if (qbi + 1 < nqb) {
            commit_q(cur ^ 1);
            if (qbi + 2 < nqb) fetch_q(qbi + 2);
            __syncthreads();
        }
    }
    store_rows_t<HD>(dk, 1.f, a.dk + (long)b * Tk * a.lddk + h * HD, a.lddk, krow0, Tk, lane);
    store_rows_t<HD>(dv, 1.f, a.dv + (long)b * Tk * a.lddv + h * HD, a.lddv, krow0, Tk, lane);
}

// dQ and dK/dV of one attention as ONE grid: blocks [0, nqb) own 64 query rows each, the rest 64 key rows each.  The two
// halves are independent given delta, so they overlap instead of queueing as two launch-latency-bound kernels.
template <int HD>
__global__ __launch_bounds__(256, 3) void attn_bwd_kernel(AttnArgs a, int nqb) {
    __shared__ __attribute__((aligned(16))) char smem[BwdSmem<HD>::BYTES];
    if ((int)blockIdx.x < nqb) attn_bwd_dq_body<HD>(a, blockIdx.x, smem);
    else attn_bwd_dkv_body<HD>(a, blockIdx.x - nqb, smem);
}

// ============================================================================ head dim 64: tiles through an LDS-DMA ring
// The kernels above keep ONE tile pair in flight (registers): with four 64-row blocks per sequence their loop runs at one memory
// round trip per block (tools/bench_attn.py: 1.3-1.8 us per block for a lone workgroup, 0.3 us of it arithmetic).  For the path's
// head dim (64: a tile row is 128 bytes = 8 chunks) the tile pairs travel by global_load_lds into a ring of NST stages, NST - 1 of
// them in flight, no registers held: unpadded rows, 16-byte chunk c of row r at position c ^ (r & 7) (as gemm_glds_kernel), a counted
// s_waitcnt vmcnt + ONE raw barrier per block.  Rows behind the end of a matrix are clamped, not zero-filled: every use of them is
// masked to an exact zero and the data is finite.  rowsum(dO * O) of the dK/dV workgroups is computed ahead for 8 query blocks at a
// time (one exposed round trip per 512 queries) instead of block by block inside the loop.
constexpr int RTILE = BLK * 64;                              // elements of a ring tile
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;
__device__ __forceinline__ bf16x8 rfrag_row(const bf16* tile, int r0, int ks, int lane) {
    const int row = r0 + (lane & 15), ch = ks * 4 + (lane >> 4);
    return ld8(tile + row * 64 + ((ch ^ (row & 7)) * 8));
}
__device__ __forceinline__ bf16x8 rfrag_tr(const bf16* tile, int krow0, int c0, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int row = krow0 + 4 * g + q, col = c0 + 4 * p;
    const bf16* a0 = tile + row * 64 + (((col >> 3) ^ (row & 7)) * 8) + (col & 7);
    const bf16* a1 = a0 + 16 * 64;                           // (row + 16: the same swizzle)
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)a0);
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)a1);
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}
// rows r0.. of a [nrows][ld] matrix (64 dims from column 0 of `src`) -> ring tile; 512 / NT DMA instructions per thread
template <int NT>
__device__ __forceinline__ void ring_issue(bf16* tile, const bf16* src, long ld, int r0, int nrows, int tid) {
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (the LDS base of a DMA instruction is wave-uniform: M0)
#pragma unroll
    for (int i = 0; i < 512 / NT; ++i) {
        const int c = tid + i * NT, r = c >> 3;
        const int row = r0 + r < nrows ? r0 + r : nrows - 1;
        const bf16* g = src + (long)row * ld + (((c & 7) ^ (r & 7)) * 8);
        // (behind this builtin the compiler drains the ring -- s_waitcnt vmcnt(0) -- in front of the first ds_read_b64_tr_b16 of a block,
        // common.h glds16; with the asm form instead the encoder / decoder attention slots measured the same, 105 / 200 us: four key
        // blocks per sequence, nothing left in flight to lose.  The encoder-row weight-gradient GEMM is where it mattered.)
        __builtin_amdgcn_global_load_lds((gptr_t*)g, (lptr_t*)(tile + (wave * 64 + i * NT) * 8), 16, 0, 0);
    }
}
// s_waitcnt vmcnt(LT n): all but the n youngest stages (LT = 2 * 512 / NT DMA instructions per thread each) have landed
template <int LT>
__device__ __forceinline__ void ring_wait(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LT) : "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LT) : "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LT) : "memory"); break;
    }
}

// NW = waves per workgroup, RT = 16-row tiles per wave: a workgroup owns 16 NW RT rows.  Every workgroup streams the WHOLE other operand through its ring, so at
// RT = 1 the 250-frame encoder pulled each K/V tile four times from L2 (33 MB per launch: 6.7 us at the ~5 TB/s the L2 -> LDS path
// gives all CUs together, tools/bench_attn.py with the arithmetic ablated) and a wave's 8 + 8 MFMAs per block sat in a serial chain of
// LDS reads, folds and exp2; RT = 2 halves the streamed bytes, shares every LDS fragment between two row tiles and gives the chain
// twice the independent work -- but 256 one-wave-per-SIMD workgroups run their chains without anyone to overlap with (forward 12.7 -> 13.2 us).
// EIGHT waves of one row tile each stream the same bytes per row and keep two chains per SIMD; four waves serve short sequences (the decoder's 37 tokens).
constexpr int FWD_NST = 4, BWD_NST = 3, STAT_BLKS = 8;
template <int NW, int RT>
__global__ __launch_bounds__(64 * NW) void attn_fwd_ring_kernel(AttnArgs a) {
    constexpr int HD = 64, KS = 2, DT = 4, NST = FWD_NST, BQ = 16 * NW * RT, NT = 64 * NW, LT = 2 * 512 / NT;
    __shared__ __attribute__((aligned(1024))) bf16 ring[NST][2][RTILE];              // [stage][K | V]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * BQ;
    const int Tq = a.Tq, Tk = a.Tk;
    const int klen = a.klens ? a.klens[b] : Tk;
    const float scale = rsqrtf((float)HD);
    const float inv_keep = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    const uint32_t dseed = (a.drop_p > 0.f && a.seed_ptr) ? *a.seed_ptr : a.seed;
    const bf16* qb = a.q + (long)b * Tq * a.ldq + h * HD;
    const bf16* kb_ = a.k + (long)b * Tk * a.ldk + h * HD;
    const bf16* vb = a.v + (long)b * Tk * a.ldv + h * HD;

    const int qrow0 = q0 + wave * 16 * RT, g4 = (lane >> 4) * 4;
    bf16x8 qf[RT][KS];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[rt][ks] = frag_global<HD>(qb, a.ldq, qrow0 + rt * 16, Tq, ks, lane);

    int nkb = (Tk + BLK - 1) / BLK;
    if (a.causal) { const int lim = (q0 + BQ - 1) / BLK + 1; if (lim < nkb) nkb = lim; }
    auto issue = [&](int kb) { bf16* st = ring[kb % NST][0]; ring_issue<NT>(st, kb_, a.ldk, kb * BLK, Tk, tid); ring_issue<NT>(st + RTILE, vb, a.ldv, kb * BLK, Tk, tid); };
    for (int t = 0; t < NST - 1 && t < nkb; ++t) issue(t);

    const float sc2 = scale * LOG2E;
    float m[RT], l[RT];                                     // of query qi (the same in the four lane groups); m in units of q.k
    uint32_t dbase[RT];
    f32x4 o[RT][DT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        m[rt] = NEG; l[rt] = 0.f;
        dbase[rt] = (uint32_t)((((long)b * a.H + h) * Tq + qrow0 + rt * 16 + (lane & 15)) * Tk);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[rt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    for (int kb = 0; kb < nkb; ++kb) {
        ring_wait<LT>(nkb - 1 - kb < NST - 2 ? nkb - 1 - kb : NST - 2);
        __builtin_amdgcn_s_barrier();                       // block kb landed for every wave; stage (kb-1) % NST is no longer read
        if (kb + NST - 1 < nkb) issue(kb + NST - 1);
        const bf16* cK = ring[kb % NST][0]; const bf16* cV = cK + RTILE;
        f32x4 s[RT][4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            bf16x8 kfr[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) kfr[ks] = rfrag_row(cK, jt * 16, ks, lane);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                s[rt][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) s[rt][jt] = mma16(kfr[ks], qf[rt][ks], s[rt][jt]);
            }
        }
        bf16x8 pf[RT][2];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int qr = qrow0 + rt * 16, qi = qr + (lane & 15);
            const bool full = kb * BLK + BLK <= klen && (!a.causal || kb * BLK + BLK - 1 <= qr);
            float mx = NEG;
            if (!full) {
#pragma unroll
                for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int kj = kb * BLK + jt * 16 + g4 + r;
                        if (kj >= klen || (a.causal && kj > qi)) s[rt][jt][r] = NEG;
                    }
            }
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[rt][jt][r]);
            const float mn = fmaxf(m[rt], fold_groups_max(mx));
            const float alpha = __builtin_amdgcn_exp2f((m[rt] - mn) * sc2);
            m[rt] = mn;
                float rs = 0.f;
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                float k4[4] = {1.f, 1.f, 1.f, 1.f};               // keep-scales of this lane's four consecutive keys (one hash word per pair)
                if (a.drop_p > 0.f) dropout_scale4(dseed, a.site, dbase[rt] + (uint32_t)(kb * BLK + jt * 16 + g4), a.drop_p, inv_keep, k4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float p = __builtin_amdgcn_exp2f((s[rt][jt][r] - mn) * sc2);
                    if (!full && s[rt][jt][r] <= NEG) p = 0.f;
                    rs += p;
                    s[rt][jt][r] = p * k4[r];
                }
            }
            l[rt] = l[rt] * alpha + fold_groups_sum(rs);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[rt][dt][r] *= alpha;
            pf[rt][0] = frag_acc(s[rt][0], s[rt][1]);
            pf[rt][1] = frag_acc(s[rt][2], s[rt][3]);
        }
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const bf16x8 vfr = rfrag_tr(cV, m2 * 32, dt * 16, lane);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) o[rt][dt] = mma16(vfr, pf[rt][m2], o[rt][dt]);
            }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int qi = qrow0 + rt * 16 + (lane & 15);
        if (qi < Tq && lane < 16) a.lse[((long)b * a.H + h) * Tq + qi] = __fmul_rn(m[rt], scale) + __logf(l[rt]);
        store_rows_t<HD>(o[rt], 1.f / l[rt], a.o + (long)b * Tq * a.ldo + h * HD, a.ldo, qrow0 + rt * 16, Tq, lane);
    }
}

template <int NW, int RT>
__device__ __forceinline__ void attn_bwd_ring_dq(const AttnArgs& a, const int bx, bf16* ringp) {
    constexpr int HD = 64, KS = 2, DT = 4, NST = BWD_NST, BQ = 16 * NW * RT, NT = 64 * NW, LT = 2 * 512 / NT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, h = blockIdx.y, q0 = bx * BQ;
    const int Tq = a.Tq, Tk = a.Tk;
    const int klen = a.klens ? a.klens[b] : Tk;
    const float scale = rsqrtf((float)HD);
    const float inv_keep = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    const uint32_t dseed = (a.drop_p > 0.f && a.seed_ptr) ? *a.seed_ptr : a.seed;
    const bf16* qb = a.q + (long)b * Tq * a.ldq + h * HD;
    const bf16* kb_ = a.k + (long)b * Tk * a.ldk + h * HD;
    const bf16* vb = a.v + (long)b * Tk * a.ldv + h * HD;
    const bf16* dob = a.dout + (long)b * Tq * a.lddo + h * HD;
    const bf16* ob = a.o + (long)b * Tq * a.ldo + h * HD;

    const int qrow0 = q0 + wave * 16 * RT, g4 = (lane >> 4) * 4;
    bf16x8 qf[RT][KS], dof[RT][KS];
    float dl[RT], lse[RT];
    uint32_t dbase[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        dl[rt] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[rt][ks] = frag_global<HD>(qb, a.ldq, qrow0 + rt * 16, Tq, ks, lane);
            dof[rt][ks] = frag_global<HD>(dob, a.lddo, qrow0 + rt * 16, Tq, ks, lane);
            const bf16x8 of = frag_global<HD>(ob, a.ldo, qrow0 + rt * 16, Tq, ks, lane);
#pragma unroll
            for (int j = 0; j < 8; ++j) dl[rt] = fmaf((float)of[j], (float)dof[rt][ks][j], dl[rt]);
        }
        const int qi = qrow0 + rt * 16 + (lane & 15);
        lse[rt] = qi < Tq ? a.lse[((long)b * a.H + h) * Tq + qi] : 0.f;
        dbase[rt] = (uint32_t)((((long)b * a.H + h) * Tq + qi) * Tk);
    }
    int nkb = (Tk + BLK - 1) / BLK;
    if (a.causal) { const int lim = (q0 + BQ - 1) / BLK + 1; if (lim < nkb) nkb = lim; }
    auto issue = [&](int kb) { bf16* st = ringp + (kb % NST) * 2 * RTILE; ring_issue<NT>(st, kb_, a.ldk, kb * BLK, Tk, tid); ring_issue<NT>(st + RTILE, vb, a.ldv, kb * BLK, Tk, tid); };
    for (int t = 0; t < NST - 1 && t < nkb; ++t) issue(t);
    f32x4 dq[RT][DT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        dl[rt] = fold_groups_sum(dl[rt]);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) dq[rt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    for (int kb = 0; kb < nkb; ++kb) {
        ring_wait<LT>(nkb - 1 - kb < NST - 2 ? nkb - 1 - kb : NST - 2);
        __builtin_amdgcn_s_barrier();
        if (kb + NST - 1 < nkb) issue(kb + NST - 1);
        const bf16* cK = ringp + (kb % NST) * 2 * RTILE; const bf16* cV = cK + RTILE;
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2) {
            f32x4 ds[RT][2];
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) {
                const int jt = 2 * m2 + j2;
                bf16x8 kfr[KS], vfr[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) { kfr[ks] = rfrag_row(cK, jt * 16, ks, lane); vfr[ks] = rfrag_row(cV, jt * 16, ks, lane); }
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const int qr = qrow0 + rt * 16, qi = qr + (lane & 15);
                    const bool full = kb * BLK + BLK <= klen && (!a.causal || kb * BLK + BLK - 1 <= qr) && qr + 16 <= Tq;
                    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        s = mma16(kfr[ks], qf[rt][ks], s);
                        dp = mma16(vfr[ks], dof[rt][ks], dp);
                    }
                    float k4[4] = {1.f, 1.f, 1.f, 1.f};
                    if (a.drop_p > 0.f) dropout_scale4(dseed, a.site, dbase[rt] + (uint32_t)(kb * BLK + jt * 16 + g4), a.drop_p, inv_keep, k4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float p = __builtin_amdgcn_exp2f((__fmul_rn(s[r], scale) - lse[rt]) * LOG2E);
                        if (!full) {
                            const int kj = kb * BLK + jt * 16 + g4 + r;
                            if (kj >= klen || (a.causal && kj > qi) || qi >= Tq) p = 0.f;
                        }
                        ds[rt][j2][r] = p * (dp[r] * k4[r] - dl[rt]) * scale;
                    }
                }
            }
            bf16x8 df[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) df[rt] = frag_acc(ds[rt][0], ds[rt][1]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const bf16x8 ktr = rfrag_tr(cK, m2 * 32, dt * 16, lane);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) dq[rt][dt] = mma16(ktr, df[rt], dq[rt][dt]);
            }
        }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) store_rows_t<HD>(dq[rt], 1.f, a.dq + (long)b * Tq * a.lddq + h * HD, a.lddq, qrow0 + rt * 16, Tq, lane);
}

template <int NW, int RT>
__device__ __forceinline__ void attn_bwd_ring_dkv(const AttnArgs& a, const int bx, bf16* ringp, float* sStat) {
    constexpr int HD = 64, KS = 2, DT = 4, NST = BWD_NST, BK = 16 * NW * RT, NT = 64 * NW, LT = 2 * 512 / NT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, h = blockIdx.y, k0 = bx * BK;
    const int Tq = a.Tq, Tk = a.Tk;
    const int klen = a.klens ? a.klens[b] : Tk;
    const float scale = rsqrtf((float)HD);
    const float inv_keep = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    const uint32_t dseed = (a.drop_p > 0.f && a.seed_ptr) ? *a.seed_ptr : a.seed;
    const bf16* qb = a.q + (long)b * Tq * a.ldq + h * HD;
    const bf16* kb_ = a.k + (long)b * Tk * a.ldk + h * HD;
    const bf16* vb = a.v + (long)b * Tk * a.ldv + h * HD;
    const bf16* dob = a.dout + (long)b * Tq * a.lddo + h * HD;
    const bf16* ob = a.o + (long)b * Tq * a.ldo + h * HD;

    const int krow0 = k0 + wave * 16 * RT, g4 = (lane >> 4) * 4;
    bf16x8 kf[RT][KS], vf[RT][KS];
    uint32_t dbase[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kf[rt][ks] = frag_global<HD>(kb_, a.ldk, krow0 + rt * 16, Tk, ks, lane);
            vf[rt][ks] = frag_global<HD>(vb, a.ldv, krow0 + rt * 16, Tk, ks, lane);
        }
        dbase[rt] = (uint32_t)(((long)b * a.H + h) * Tq * Tk + krow0 + rt * 16 + (lane & 15));
    }
    f32x4 dk[RT][DT], dv[RT][DT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) { dk[rt][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[rt][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    const int nqb = (Tq + BLK - 1) / BLK;
    const int qb0 = a.causal ? k0 / BLK : 0;
    auto issue = [&](int qbi) {
        bf16* st = ringp + ((qbi - qb0) % NST) * 2 * RTILE;
        ring_issue<NT>(st, qb, a.ldq, qbi * BLK, Tq, tid); ring_issue<NT>(st + RTILE, dob, a.lddo, qbi * BLK, Tq, tid);
    };
    // lse and rowsum(dO * O) of STAT_BLKS query blocks at a time: row tid/4 of each block, its 8 chunks dealt to 4 threads
    auto stats = [&](int qbs) {
        const int nb = nqb - qbs < STAT_BLKS ? nqb - qbs : STAT_BLKS;
        if (NW > 4 && tid >= 256) return;                    // (a block's 64 rows x 4 threads; wave-uniform)
        for (int j0 = 0; j0 < nb; j0 += 4) {                 // 4 blocks = 16 16-byte loads per thread in flight
            float d0[4], l0[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int qi = (qbs + j0 + j) * BLK + (tid >> 2), row = qi < Tq ? qi : Tq - 1;
                float t = 0.f;
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    const int ch = (tid & 3) + 4 * cc;
                    const bf16x8 x = ld8(ob + (long)row * a.ldo + ch * 8), y = ld8(dob + (long)row * a.lddo + ch * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) t = fmaf((float)x[e], (float)y[e], t);
                }
                d0[j] = qi < Tq ? t : 0.f;
                l0[j] = qi < Tq ? a.lse[((long)b * a.H + h) * Tq + row] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float d = d0[j];
                d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64);
                if ((tid & 3) == 0 && j0 + j < nb) {
                    sStat[(j0 + j) * BLK + (tid >> 2)] = l0[j];
                    sStat[STAT_BLKS * BLK + (j0 + j) * BLK + (tid >> 2)] = d;
                }
            }
        }
    };
    if (qb0 < nqb) {
        stats(qb0);                                          // (ordinary loads first: they retire before the DMAs in the vmcnt order)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // (the statistics are in LDS before this wave reaches the loop's first raw barrier)
        for (int t = 0; t < NST - 1 && qb0 + t < nqb; ++t) issue(qb0 + t);
    }
    for (int qbi = qb0; qbi < nqb; ++qbi) {
        const int it = qbi - qb0, sb = it % STAT_BLKS;
        if (it > 0 && sb == 0) {
            // next group of statistics: drain the ring's DMAs around it (the loads of stats() share the vmcnt counter); all waves are past
            // their reads of the previous group after the barrier
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            stats(qbi);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (qbi + NST - 1 < nqb) issue(qbi + NST - 1);
        } else {
            ring_wait<LT>(nqb - 1 - qbi < NST - 2 ? nqb - 1 - qbi : NST - 2);
            __builtin_amdgcn_s_barrier();
            if (qbi + NST - 1 < nqb) issue(qbi + NST - 1);
        }
        const bf16* cQ = ringp + (it % NST) * 2 * RTILE; const bf16* cDO = cQ + RTILE;
        const float* cLse = sStat + sb * BLK; const float* cDl = cLse + STAT_BLKS * BLK;
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2) {
            f32x4 pt[RT][2], dst[RT][2];
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) {
                const int jt = 2 * m2 + j2;
                bf16x8 qfr[KS], dofr[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) { qfr[ks] = rfrag_row(cQ, jt * 16, ks, lane); dofr[ks] = rfrag_row(cDO, jt * 16, ks, lane); }
                const f32x4 ls4 = *reinterpret_cast<const f32x4*>(cLse + jt * 16 + g4);
                const f32x4 dl4 = *reinterpret_cast<const f32x4*>(cDl + jt * 16 + g4);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const int kr = krow0 + rt * 16, kj = kr + (lane & 15);
                    const bool full = kr + 16 <= klen && qbi * BLK + BLK <= Tq && (!a.causal || kr + 15 <= qbi * BLK);
                    f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f}, dpt = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        st = mma16(qfr[ks], kf[rt][ks], st);
                        dpt = mma16(dofr[ks], vf[rt][ks], dpt);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int qi = qbi * BLK + jt * 16 + g4 + r;
                        float p = __builtin_amdgcn_exp2f((__fmul_rn(st[r], scale) - ls4[r]) * LOG2E);
                        if (!full && (kj >= klen || (a.causal && kj > qi) || qi >= Tq)) p = 0.f;
                        float ms = 1.f;
                        if (a.drop_p > 0.f) ms = dropout_scale(dseed, a.site, dbase[rt] + (uint32_t)qi * (uint32_t)Tk, a.drop_p, inv_keep);
                        pt[rt][j2][r] = p * ms;
                        dst[rt][j2][r] = p * (dpt[r] * ms - dl4[r]) * scale;
                    }
                }
            }
            bf16x8 pf[RT], df[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) { pf[rt] = frag_acc(pt[rt][0], pt[rt][1]); df[rt] = frag_acc(dst[rt][0], dst[rt][1]); }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const bf16x8 dotr = rfrag_tr(cDO, m2 * 32, dt * 16, lane), qtr = rfrag_tr(cQ, m2 * 32, dt * 16, lane);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    dv[rt][dt] = mma16(dotr, pf[rt], dv[rt][dt]);
                    dk[rt][dt] = mma16(qtr, df[rt], dk[rt][dt]);
                }
            }
        }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        store_rows_t<HD>(dk[rt], 1.f, a.dk + (long)b * Tk * a.lddk + h * HD, a.lddk, krow0 + rt * 16, Tk, lane);
        store_rows_t<HD>(dv[rt], 1.f, a.dv + (long)b * Tk * a.lddv + h * HD, a.lddv, krow0 + rt * 16, Tk, lane);
    }
}
template <int NW, int RQ, int RK>
__global__ __launch_bounds__(64 * NW, (NW > 4 || RQ > 1 || RK > 1) ? 2 : 3) void attn_bwd_ring_kernel(AttnArgs a, int nqb) {
    __shared__ __attribute__((aligned(1024))) bf16 ring[BWD_NST * 2 * RTILE];
    __shared__ __attribute__((aligned(16))) float stat[2 * STAT_BLKS * BLK];
    if ((int)blockIdx.x < nqb) attn_bwd_ring_dq<NW, RQ>(a, blockIdx.x, ring);
    else attn_bwd_ring_dkv<NW, RK>(a, blockIdx.x - nqb, ring, stat);
}

template <int HD>
int launch_fwd(const AttnArgs& a, hipStream_t s) {
    constexpr int long_min = 65;                             // sequences from this length on: 128-row workgroups (8 waves x 1 row tile)
    if (HD == 64) {
        if (a.Tq < long_min) hipLaunchKernelGGL((attn_fwd_ring_kernel<4, 1>), dim3((a.Tq + 63) / 64, a.H, a.B), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((attn_fwd_ring_kernel<8, 1>), dim3((a.Tq + 127) / 128, a.H, a.B), dim3(512), 0, s, a);
    } else {
        hipLaunchKernelGGL(attn_fwd_kernel<HD>, dim3((a.Tq + BLK - 1) / BLK, a.H, a.B), dim3(256), 0, s, a);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
template <int HD>
int launch_bwd(const AttnArgs& a, hipStream_t s) {
    constexpr int long_min = 65;
    if (HD == 64) {
        // the dQ and the dK/dV workgroups share one grid, hence one workgroup size.  Measured (16 x 8 heads, fwd + bwd, us): 250 x 250:
        // 4 waves x 1 tile 52.9, 8 x 1 45.5, 4 x 2 44.3; 37 x 250: 24.7, 29.0 (a 128-row dQ workgroup holds 37 rows), 23.5 with
        // two tiles on the key side only
        const bool lq = a.Tq >= long_min, lk = a.Tk >= long_min;
        {
            const int rq = lq ? 2 : 1, rk = lk ? 2 : 1;
            const int nqb = (a.Tq + 64 * rq - 1) / (64 * rq), nkb = (a.Tk + 64 * rk - 1) / (64 * rk);
            const dim3 grid(nqb + nkb, a.H, a.B);
            if (rq == 2 && rk == 2) hipLaunchKernelGGL((attn_bwd_ring_kernel<4, 2, 2>), grid, dim3(256), 0, s, a, nqb);
            else if (rq == 1 && rk == 2) hipLaunchKernelGGL((attn_bwd_ring_kernel<4, 1, 2>), grid, dim3(256), 0, s, a, nqb);
            else if (rq == 2) hipLaunchKernelGGL((attn_bwd_ring_kernel<4, 2, 1>), grid, dim3(256), 0, s, a, nqb);
            else hipLaunchKernelGGL((attn_bwd_ring_kernel<4, 1, 1>), grid, dim3(256), 0, s, a, nqb);
        }
    } else {
        const int nqb = (a.Tq + BLK - 1) / BLK, nkb = (a.Tk + BLK - 1) / BLK;
        hipLaunchKernelGGL(attn_bwd_kernel<HD>, dim3(nqb + nkb, a.H, a.B), dim3(256), 0, s, a, nqb);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

bool check(const AttnArgs& a) {
    if ((a.ldq & 7) || (a.ldk & 7) || (a.ldv & 7)) { mk_set_error("mk_attn", "row strides must be multiples of 8"); return false; }
    if (a.hd != 16 && a.hd != 32 && a.hd != 64) { mk_set_error("mk_attn", "head dim must be 16, 32 or 64"); return false; }
    return true;
}

}  // namespace

int mk_attn_fwd(const AttnArgs& a, hipStream_t s) {
    if (!check(a)) return -1;
    if (a.hd == 64) return launch_fwd<64>(a, s);
    if (a.hd == 32) return launch_fwd<32>(a, s);
    return launch_fwd<16>(a, s);
}
int mk_attn_bwd(const AttnArgs& a, hipStream_t s) {
    if (!check(a)) return -1;
    if ((a.lddo & 7)) { mk_set_error("mk_attn_bwd", "lddo must be a multiple of 8"); return -1; }
    if (a.hd == 64) return launch_bwd<64>(a, s);
    if (a.hd == 32) return launch_bwd<32>(a, s);
    return launch_bwd<16>(a, s);
}
