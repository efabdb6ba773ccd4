// Fused multi-head attention forward / backward (flash-style, online softmax) for gfx950.
//
// Restates torch nn.MultiheadAttention as used by nn.TransformerEncoder/DecoderLayer in
// src/model/transformer_pytorch/mono_transformer_torch.py:74-98,200-203:
//   scores = q k^T / sqrt(hd); key-padding mask (encoder self, decoder cross) or causal mask
//   (decoder self; no target key-padding mask -- quirk Q7); softmax; dropout on the
//   probabilities; . v
// q/k/v are read straight out of the packed in_proj GEMM output (row b*T+t, head h at
// column h*hd), so no head split/merge copies exist.  Score matrices never reach HBM.
//
// One workgroup = 4 waves = 64 query rows (fwd, dQ) or 64 key rows (dK/dV); K/V (or Q/dO)
// tiles of 64 rows live in LDS in their natural [row][dim] layout and serve both MFMA operand
// forms: 16-byte row reads (contraction over dim) and ds_read_b64_tr_b16 transposing reads
// (contraction over the row index).
#include "common.h"
#include "kernels.h"

namespace {

constexpr int BLK = 64;             // rows per tile
constexpr int LDP = 72;             // P scratch row stride (elements)
constexpr float NEG = -1e30f;

template <int HD> struct Cfg {
    static constexpr int HDP = HD < 32 ? 32 : HD;   // contraction length for q.k (zero padded)
    static constexpr int KS = HDP / 32;
    static constexpr int DT = HD / 16;
    static constexpr int LD = HDP + 16;             // tile row stride: (LD*2 B) == 8 dwords mod 64 for hd 64
};

typedef __attribute__((address_space(3))) bf16x4 lds_b4;

// B-operand fragment, contraction over the tile's ROW index: rows krow0..krow0+31, columns c0..c0+15.
// k order inside the fragment is permuted (4g+e | 16+4g+e); the A operand uses frag_a_perm to match.
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
// A-operand fragment from the P scratch ([16 rows][64 cols]) with the same k permutation
__device__ __forceinline__ bf16x8 frag_a_perm(const bf16* sp, int k0, int lane) {
    const int g = lane >> 4, row = lane & 15;
    const bf16x4 lo = *reinterpret_cast<const bf16x4*>(sp + row * LDP + k0 + 4 * g);
    const bf16x4 hi = *reinterpret_cast<const bf16x4*>(sp + row * LDP + k0 + 16 + 4 * g);
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
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
// The loops below fetch tile i+1 right after committing tile i, so the memory round trip runs under tile i's MFMAs
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

// Write a wave's 16 x HD accumulator tile (lane: rows 4g+r, column dt*16 + (lane&15)) as whole row segments: through the
// wave's P scratch, then 16 bytes per lane.  The element-wise form issued 4*DT two-byte stores per lane -- for the three
// outputs of the backward pass 192 vector-memory instructions per workgroup, more than all its loads.
template <int HD>
__device__ __forceinline__ void store_rows16(bf16* sp, const f32x4 (&acc)[HD / 16], const float (&mul)[4], bf16* dst, long ld, int row0,
                                             int nrows, int lane) {
    constexpr int DT = HD / 16, CPR = HD / 8;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) sp[((lane >> 4) * 4 + r) * LDP + dt * 16 + (lane & 15)] = (bf16)(acc[dt][r] * mul[r]);
    // (same wave: LDS operations complete in order, no barrier needed)
#pragma unroll
    for (int u = 0; u < (16 * CPR + 63) / 64; ++u) {
        const int c = lane + 64 * u, row = c / CPR, col = (c % CPR) * 8;
        if (c < 16 * CPR && row0 + row < nrows) st8(dst + (long)(row0 + row) * ld + col, ld8(sp + row * LDP + col));
    }
}

// ---------------------------------------------------------------------------- forward
template <int HD>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a) {
    using C = Cfg<HD>;
    constexpr int LD = C::LD, KS = C::KS, DT = C::DT;
    __shared__ __attribute__((aligned(16))) bf16 sK[BLK * LD];
    __shared__ __attribute__((aligned(16))) bf16 sV[BLK * LD];
    __shared__ __attribute__((aligned(16))) bf16 sP[4][16 * LDP];

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

    zero_tile<HD>(sK, tid);
    zero_tile<HD>(sV, tid);

    const int qrow0 = q0 + wave * 16;
    bf16x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = frag_global<HD>(qb, a.ldq, qrow0, Tq, ks, lane);

    float m[4], l[4];
    f32x4 o[DT];
#pragma unroll
    for (int r = 0; r < 4; ++r) { m[r] = NEG; l[r] = 0.f; }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    int nkb = (Tk + BLK - 1) / BLK;
    if (a.causal) { const int lim = (q0 + BLK - 1) / BLK + 1; if (lim < nkb) nkb = lim; }

    TileRegs<HD> tk, tv;
    tk.fetch(kb_, a.ldk, 0, Tk, tid);
    tv.fetch(vb, a.ldv, 0, Tk, tid);
    for (int kb = 0; kb < nkb; ++kb) {
        __syncthreads();
        tk.commit(sK, Tk, tid);
        tv.commit(sV, Tk, tid);
        __syncthreads();
        if (kb + 1 < nkb) { tk.fetch(kb_, a.ldk, (kb + 1) * BLK, Tk, tid); tv.fetch(vb, a.ldv, (kb + 1) * BLK, Tk, tid); }

        f32x4 s[4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            s[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) s[jt] = mma16(qf[ks], frag_row<LD>(sK, jt * 16, ks, lane), s[jt]);
        }
        float mx[4] = {NEG, NEG, NEG, NEG};
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            const int kj = kb * BLK + jt * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qi = qrow0 + (lane >> 4) * 4 + r;
                float v = s[jt][r] * scale;
                if (kj >= klen || (a.causal && kj > qi)) v = NEG;
                s[jt][r] = v;
                mx[r] = fmaxf(mx[r], v);
            }
        }
        float alpha[4], rs[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = mx[r];
            v = fmaxf(v, __shfl_xor(v, 1, 64)); v = fmaxf(v, __shfl_xor(v, 2, 64));
            v = fmaxf(v, __shfl_xor(v, 4, 64)); v = fmaxf(v, __shfl_xor(v, 8, 64));
            const float mn = fmaxf(m[r], v);
            alpha[r] = __expf(m[r] - mn);
            m[r] = mn;
            rs[r] = 0.f;
        }
        bf16* sp = sP[wave];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            const int kj = kb * BLK + jt * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = (lane >> 4) * 4 + r;
                float p = __expf(s[jt][r] - m[r]);
                if (s[jt][r] <= NEG) p = 0.f;
                rs[r] += p;
                if (a.drop_p > 0.f) {
                    const uint32_t idx = (uint32_t)((((long)b * a.H + h) * Tq + (qrow0 + row)) * Tk + kj);
                    p *= dropout_scale(dseed, a.site, idx, a.drop_p, inv_keep);
                }
                sp[row * LDP + jt * 16 + (lane & 15)] = (bf16)p;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = rs[r];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            l[r] = l[r] * alpha[r] + v;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) o[dt][r] *= alpha[r];
        }
        __syncthreads();
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
            const bf16x8 pf = frag_a_perm(sp, ks2 * 32, lane);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) o[dt] = mma16(pf, frag_tr<LD>(sV, ks2 * 32, dt * 16, lane), o[dt]);
        }
    }
    float inv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qi = qrow0 + (lane >> 4) * 4 + r;
        inv[r] = 1.f / l[r];
        if (qi < Tq && (lane & 15) == 0) a.lse[((long)b * a.H + h) * Tq + qi] = m[r] + __logf(l[r]);
    }
    store_rows16<HD>(sP[wave], o, inv, a.o + (long)b * Tq * a.ldo + h * HD, a.ldo, qrow0, Tq, lane);
}

// ---------------------------------------------------------------------------- backward
// delta_i = rowsum(dO_i * O_i) is not a separate pass: the dQ workgroups take it from the dO / O fragments they hold anyway,
// the dK/dV workgroups compute it for each query block while that block's tiles are in flight (two 16-byte loads per thread).
// LDS of the two bodies of attn_bwd_kernel is ONE buffer (a workgroup runs one body or the other): as separate __shared__ arrays the
// kernel was charged the SUM (69 KB -> two workgroups per CU); the union is 39 KB -> four per CU
template <int HD> struct BwdSmem {
    static constexpr int LD = Cfg<HD>::LD;
    static constexpr size_t TILE = sizeof(bf16) * BLK * LD, SCR = sizeof(bf16) * 4 * 16 * LDP;
    static constexpr size_t DQ = 2 * TILE + SCR, DKV = 2 * TILE + 2 * SCR + sizeof(float) * 2 * BLK;
    static constexpr size_t BYTES = DQ > DKV ? DQ : DKV;
};
template <int HD>
__device__ __forceinline__ void attn_bwd_dq_body(const AttnArgs& a, const int bx, char* smem) {
    using C = Cfg<HD>;
    constexpr int LD = C::LD, KS = C::KS, DT = C::DT;
    bf16* sK = reinterpret_cast<bf16*>(smem);
    bf16* sV = sK + BLK * LD;
    bf16 (*sP)[16 * LDP] = reinterpret_cast<bf16 (*)[16 * LDP]>(sV + BLK * LD);

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

    zero_tile<HD>(sK, tid);
    zero_tile<HD>(sV, tid);
    const int qrow0 = q0 + wave * 16;
    bf16x8 qf[KS], dof[KS];
    float dpart = 0.f;                                     // this lane's share of rowsum(dO * O) for query row (lane & 15)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        qf[ks] = frag_global<HD>(qb, a.ldq, qrow0, Tq, ks, lane);
        dof[ks] = frag_global<HD>(dob, a.lddo, qrow0, Tq, ks, lane);
        const bf16x8 of = frag_global<HD>(ob, a.ldo, qrow0, Tq, ks, lane);
#pragma unroll
        for (int j = 0; j < 8; ++j) dpart = fmaf((float)of[j], (float)dof[ks][j], dpart);
    }
    dpart += __shfl_xor(dpart, 16, 64); dpart += __shfl_xor(dpart, 32, 64);     // the 4 lanes of a row hold its 4 dim chunks
    float lse[4], dl[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qi = qrow0 + (lane >> 4) * 4 + r;
        const long idx = ((long)b * a.H + h) * Tq + qi;
        lse[r] = qi < Tq ? a.lse[idx] : 0.f;
        dl[r] = __shfl(dpart, (lane >> 4) * 4 + r, 64);      // accumulator row (lane>>4)*4 + r  <-  lane with that row index
    }
    f32x4 dq[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    int nkb = (Tk + BLK - 1) / BLK;
    if (a.causal) { const int lim = (q0 + BLK - 1) / BLK + 1; if (lim < nkb) nkb = lim; }
    TileRegs<HD> tk, tv;
    tk.fetch(kb_, a.ldk, 0, Tk, tid);
    tv.fetch(vb, a.ldv, 0, Tk, tid);
    for (int kb = 0; kb < nkb; ++kb) {
        __syncthreads();
        tk.commit(sK, Tk, tid);
        tv.commit(sV, Tk, tid);
        __syncthreads();
        if (kb + 1 < nkb) { tk.fetch(kb_, a.ldk, (kb + 1) * BLK, Tk, tid); tv.fetch(vb, a.ldv, (kb + 1) * BLK, Tk, tid); }
        bf16* sp = sP[wave];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                s = mma16(qf[ks], frag_row<LD>(sK, jt * 16, ks, lane), s);
                dp = mma16(dof[ks], frag_row<LD>(sV, jt * 16, ks, lane), dp);
            }
            const int kj = kb * BLK + jt * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = (lane >> 4) * 4 + r, qi = qrow0 + row;
                float p = __expf(s[r] * scale - lse[r]);
                if (kj >= klen || (a.causal && kj > qi) || qi >= Tq) p = 0.f;
                float dpv = dp[r];
                if (a.drop_p > 0.f) {
                    const uint32_t idx = (uint32_t)((((long)b * a.H + h) * Tq + qi) * Tk + kj);
                    dpv *= dropout_scale(dseed, a.site, idx, a.drop_p, inv_keep);
                }
                sp[row * LDP + jt * 16 + (lane & 15)] = (bf16)(p * (dpv - dl[r]) * scale);
            }
        }
        __syncthreads();
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
            const bf16x8 df = frag_a_perm(sp, ks2 * 32, lane);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) dq[dt] = mma16(df, frag_tr<LD>(sK, ks2 * 32, dt * 16, lane), dq[dt]);
        }
    }
    {
        const float one[4] = {1.f, 1.f, 1.f, 1.f};
        store_rows16<HD>(sP[wave], dq, one, a.dq + (long)b * Tq * a.lddq + h * HD, a.lddq, qrow0, Tq, lane);
    }
}

template <int HD>
__device__ __forceinline__ void attn_bwd_dkv_body(const AttnArgs& a, const int bx, char* smem) {
    using C = Cfg<HD>;
    constexpr int LD = C::LD, KS = C::KS, DT = C::DT;
    bf16* sQ = reinterpret_cast<bf16*>(smem);
    bf16* sDO = sQ + BLK * LD;
    bf16 (*sP)[16 * LDP] = reinterpret_cast<bf16 (*)[16 * LDP]>(sDO + BLK * LD);
    bf16 (*sDS)[16 * LDP] = sP + 4;
    float* sLse = reinterpret_cast<float*>(sDS + 4);
    float* sDl = sLse + BLK;

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

    zero_tile<HD>(sQ, tid);
    zero_tile<HD>(sDO, tid);
    const int krow0 = k0 + wave * 16;
    bf16x8 kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        kf[ks] = frag_global<HD>(kb_, a.ldk, krow0, Tk, ks, lane);
        vf[ks] = frag_global<HD>(vb, a.ldv, krow0, Tk, ks, lane);
    }
    f32x4 dk[DT], dv[DT];
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
    if (qb0 < nqb) fetch_q(qb0);
    for (int qbi = qb0; qbi < nqb; ++qbi) {
        __syncthreads();
        tq.commit(sQ, Tq, tid);
        tdo.commit(sDO, Tq, tid);
        if (tid < BLK) sLse[tid] = nlse;
        if ((tid & 3) == 0) sDl[tid >> 2] = ndl;
        __syncthreads();
        if (qbi + 1 < nqb) fetch_q(qbi + 1);
        bf16* sp = sP[wave];
        bf16* sd = sDS[wave];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f}, dpt = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                st = mma16(kf[ks], frag_row<LD>(sQ, jt * 16, ks, lane), st);
                dpt = mma16(vf[ks], frag_row<LD>(sDO, jt * 16, ks, lane), dpt);
            }
            const int ql = jt * 16 + (lane & 15), qi = qbi * BLK + ql;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = (lane >> 4) * 4 + r, kj = krow0 + row;
                float p = __expf(st[r] * scale - sLse[ql]);
                if (kj >= klen || (a.causal && kj > qi) || qi >= Tq) p = 0.f;
                float ms = 1.f;
                if (a.drop_p > 0.f) {
                    const uint32_t idx = (uint32_t)((((long)b * a.H + h) * Tq + qi) * Tk + kj);
                    ms = dropout_scale(dseed, a.site, idx, a.drop_p, inv_keep);
                }
                sp[row * LDP + ql] = (bf16)(p * ms);
                sd[row * LDP + ql] = (bf16)(p * (dpt[r] * ms - sDl[ql]) * scale);
            }
        }
        __syncthreads();
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
            const bf16x8 pf = frag_a_perm(sp, ks2 * 32, lane);
            const bf16x8 df = frag_a_perm(sd, ks2 * 32, lane);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                dv[dt] = mma16(pf, frag_tr<LD>(sDO, ks2 * 32, dt * 16, lane), dv[dt]);
                dk[dt] = mma16(df, frag_tr<LD>(sQ, ks2 * 32, dt * 16, lane), dk[dt]);
            }
        }
    }
    {
        const float one[4] = {1.f, 1.f, 1.f, 1.f};
        store_rows16<HD>(sP[wave], dk, one, a.dk + (long)b * Tk * a.lddk + h * HD, a.lddk, krow0, Tk, lane);
        store_rows16<HD>(sDS[wave], dv, one, a.dv + (long)b * Tk * a.lddv + h * HD, a.lddv, krow0, Tk, lane);
    }
}

// dQ and dK/dV of one attention as ONE grid: blocks [0, nqb) own 64 query rows each, the rest 64 key rows each.  The two
// halves are independent given delta, so they overlap instead of queueing as two launch-latency-bound kernels.
template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_kernel(AttnArgs a, int nqb) {
    __shared__ __attribute__((aligned(16))) char smem[BwdSmem<HD>::BYTES];
    if ((int)blockIdx.x < nqb) attn_bwd_dq_body<HD>(a, blockIdx.x, smem);
    else attn_bwd_dkv_body<HD>(a, blockIdx.x - nqb, smem);
}

template <int HD>
int launch_fwd(const AttnArgs& a, hipStream_t s) {
    dim3 grid((a.Tq + BLK - 1) / BLK, a.H, a.B);
    hipLaunchKernelGGL(attn_fwd_kernel<HD>, grid, dim3(256), 0, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
template <int HD>
int launch_bwd(const AttnArgs& a, hipStream_t s) {
    const int nqb = (a.Tq + BLK - 1) / BLK, nkb = (a.Tk + BLK - 1) / BLK;
    hipLaunchKernelGGL(attn_bwd_kernel<HD>, dim3(nqb + nkb, a.H, a.B), dim3(256), 0, s, a, nqb);
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
