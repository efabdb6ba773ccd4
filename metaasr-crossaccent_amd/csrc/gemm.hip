// bf16 MFMA GEMM with fused epilogues (gfx950).
//
//   C[m][n] = epi( alpha * sum_k A(m,k) * B(n,k) )
//
// The three products of a Linear layer (reference: torch nn.Linear forward/backward as
// dispatched from src/model/transformer_pytorch/mono_transformer_torch.py:62,66,120,206 and
// the nn.Transformer* layers :74-98) map onto two operand forms:
//   forward   Y  = X  W^T    A = X [M][K],  B = W  [N][K]   both k-contiguous      (NT form)
//   dgrad     dX = dY W      A = dY[M][N],  B = Wt [K][N]   (transposed bf16 shadow, NT form)
//   wgrad     dW = dY^T X    A = dY viewed [k=m][i=n], B = X viewed [k=m][j]       (reduction-major form)
// Reduction-major tiles are staged in their natural layout ([k][row], 16-byte vector writes)
// and the MFMA fragments are fetched with gfx950's transposing LDS read (ds_read_b64_tr_b16),
// so no scatter writes are needed.  Both operands then use the same k permutation inside a
// fragment (k = 4g+e | 16+4g+e), which leaves the dot product unchanged and makes the
// transposed reads bank-conflict free (row stride == 8 dwords mod 64).
//
// Tile: BM x BN x 32, 256 threads = 4 waves in a 2x2 grid, register-staged double buffer.
#include "common.h"
#include "kernels.h"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int BK = 64;    // k-tile: halves the barrier / LDS round trips per FLOP of the 32-deep version
constexpr int LDK = 80;   // NT form: padded LDS row (elements); 160-byte stride is conflict-free for ds_read_b128 lane groups (144 B is 2-way)

template <int ROWS>
struct TileNT { bf16 d[ROWS * LDK]; };
template <int ROWS>
struct TileRM { bf16 d[BK * (ROWS + 16)]; };   // [k][row], row stride (ROWS+16)*2 B == 8 dwords mod 64

// Loads are UNCONDITIONAL (addresses clamped into the matrix) and the zero-fill of out-of-range chunks is applied
// when the registers are written to LDS: a branch around a global load makes hipcc fall back to s_waitcnt vmcnt(0)
// at every join, which serialises the whole prefetch ring.
// ---- NT stager: element (r,k) at src[r*ld + k]
template <int ROWS>
struct StagerNT {
    static constexpr int CHUNKS = ROWS * BK / 8;
    static constexpr int PT = CHUNKS / 256;
    static_assert(CHUNKS % 256 == 0, "tile must be a multiple of 256 chunks");
    bf16x8 regs[PT];
    unsigned ok;
    __device__ __forceinline__ void load(const bf16* __restrict__ src, long ld, int r0, int k0, int nrows, int nk, int tid) {
        ok = 0;
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int c = tid + i * 256;
            const int r = r0 + c / (BK / 8), k = k0 + (c % (BK / 8)) * 8;
            if (r < nrows && k < nk) ok |= 1u << i;
            const int rc = r < nrows ? r : nrows - 1, kc = k < nk ? k : nk - 8;
            regs[i] = ld8(src + (long)rc * ld + kc);
        }
    }
    __device__ __forceinline__ void store(bf16* __restrict__ dst, int tid) {
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int c = tid + i * 256;
            st8(dst + (c / (BK / 8)) * LDK + (c % (BK / 8)) * 8, (ok >> i) & 1 ? regs[i] : zero8());
        }
    }
};
// ---- reduction-major stager: element (r,k) at src[k*ld + r]; LDS keeps [k][r]
template <int ROWS>
struct StagerRM {
    static constexpr int LDT = ROWS + 16;
    static constexpr int RC = ROWS / 8;
    static constexpr int CHUNKS = BK * RC;
    static constexpr int PT = CHUNKS / 256;
    static_assert(CHUNKS % 256 == 0, "tile must be a multiple of 256 chunks");
    bf16x8 regs[PT];
    unsigned ok;
    __device__ __forceinline__ void load(const bf16* __restrict__ src, long ld, int r0, int k0, int nrows, int nk, int tid) {
        ok = 0;
        const int rmax = (nrows + 7) / 8 * 8 - 8;              // last 8-row chunk (the padded row exists, see mk_gemm)
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int c = tid + i * 256;
            const int k = k0 + c / RC, r = r0 + (c % RC) * 8;
            if (k < nk && r < nrows) ok |= 1u << i;
            const int kc = k < nk ? k : nk - 1, rc = r < nrows ? r : rmax;
            regs[i] = ld8(src + (long)kc * ld + rc);
        }
    }
    __device__ __forceinline__ void store(bf16* __restrict__ dst, int tid) {
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int c = tid + i * 256;
            st8(dst + (c / RC) * LDT + (c % RC) * 8, (ok >> i) & 1 ? regs[i] : zero8());
        }
    }
    // running column sums of the staged tile (each thread always owns the same 8 rows: 256 % RC == 0)
    __device__ __forceinline__ void accumulate(float (&acc)[8]) {
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const float m = (ok >> i) & 1 ? 1.f : 0.f;             // branch-free (a divergent branch here costs exec-mask juggling per chunk)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(m, (float)regs[i][j], acc[j]);
        }
    }
};

// fragment of 16 rows starting at r0 from a reduction-major tile (k permuted, see header)
template <int LDT>
__device__ __forceinline__ bf16x8 frag_rm(const bf16* tile, int r0, int lane, int k0 = 0) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    typedef __attribute__((address_space(3))) bf16x4 lds_b4;
    const bf16* a0 = tile + (k0 + 4 * g + q) * LDT + r0 + 4 * p;
    const bf16* a1 = a0 + 16 * LDT;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)a0);
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)a1);
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

// EPI >= 0: the set of epilogue features is a compile-time constant (dead paths vanish: the fully generic epilogue is
// ~2000 instructions, which alone cost ~8 us per launch on the small decoder GEMMs); EPI < 0: decided at run time.
// row m of the fp32 output (see GemmArgs::cseg_rows)
__device__ __forceinline__ long crow(const GemmArgs& g, int m) {
    return g.cseg_rows ? (long)(m / g.cseg_rows) * g.cseg_stride + (long)(m % g.cseg_rows) * g.ldc : (long)m * g.ldc;
}
enum { E_BIAS = 1, E_PE = 2, E_RELU = 4, E_MASK = 8, E_DROP = 16, E_RES = 32, E_ACC = 64, E_C32 = 128, E_C16 = 256 };
#define HAS(flag, runtime) (EPI >= 0 ? bool(EPI & (flag)) : bool(runtime))

// ---- shared epilogue (see the comment inside): consumes the accumulators of one BM x BN tile
// NT = threads that take part (512: the loader waves of gemm_ring_kernel store strips too; they hold no accumulators: has_acc false)
template <int BM, int BN, int EPI, int NT = 256>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, f32x4 (&acc)[BM / 32][BN / 32], char* smem, int m0, int n0, long out_delta,
                                              bool has_acc = true) {
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int FM = WM / 16, FN = WN / 16;
    constexpr int LDC = BN + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // epilogue: alpha -> bias -> pe -> relu -> mask -> dropout -> residual -> (accumulate) -> store.
    // The accumulators go through an fp32 LDS tile so that every global access of the epilogue is a 16-byte (fp32) or
    // 8-byte (bf16) row segment, and each optional input is fetched with ONE batched load per 4 outputs (per-element
    // branches around loads serialise on s_waitcnt vmcnt(0) and used to dominate the small decoder GEMMs).
    float* ct = reinterpret_cast<float*>(smem);
    if (has_acc) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    ct[(wm * WM + i * 16 + (lane >> 4) * 4 + r) * LDC + wn * WN + j * 16 + (lane & 15)] = acc[i][j][r] * g.alpha;
    }
    __syncthreads();
    // A thread owns strips of SW columns.  The epilogue is bound by the NUMBER of vector-memory wave-instructions (one
    // texture addresser per CU, ~70 cycles each whatever their width), so epilogues with 16-bit streams (bf16 output, ReLU
    // mask) use 8-column strips = 16 bytes per lane; pure fp32 epilogues keep 4 columns (same instruction count, fewer LDS
    // bank conflicts on the staged tile).  All optional inputs of a thread's strips are fetched BEFORE the first store:
    // a load after a store may alias it as far as the compiler knows, so a load-compute-store loop pays one full memory
    // round trip per strip (8 to 16 per tile; the masked dgrad GEMMs ran 3x longer than their unmasked twins).  Strips are
    // therefore handled in batches of four: loads of the batch, then its arithmetic and stores.
    constexpr int SW = (EPI >= 0 && (EPI & (E_C16 | E_MASK))) ? 8 : 4;
    constexpr int CS = BN / SW, NCH = BM * CS / NT;
    static_assert(NCH >= 1, "tile too small for this many threads");
    const int cs = (tid % CS) * SW, n = n0 + cs;
    if (n >= g.N) return;
    const bool f_bias = HAS(E_BIAS, g.bias), f_pe = HAS(E_PE, g.pe), f_relu = HAS(E_RELU, g.relu), f_mask = HAS(E_MASK, g.mask);
    const bool f_drop = HAS(E_DROP, g.drop_p > 0.f) && g.drop_p > 0.f, f_res = HAS(E_RES, g.residual);
    const bool f_acc = HAS(E_ACC, g.accumulate), f_c32 = HAS(E_C32, g.C32), f_c16 = HAS(E_C16, g.C16);
    const int nv = g.N - n < SW ? g.N - n : SW;                 // valid columns of this thread's strip
    const float inv_keep = f_drop ? 1.0f / (1.0f - g.drop_p) : 1.0f;
    const uint32_t seed = (f_drop && g.seed_ptr) ? *g.seed_ptr : g.seed;
    float bv[SW];
#pragma unroll
    for (int e = 0; e < SW; ++e) bv[e] = (f_bias && e < nv) ? g.bias[n + e] : 0.f;
    // vector paths need a full strip and aligned rows (n is a multiple of SW by construction)
    constexpr int A16 = SW - 1;                                  // bf16 strip = SW * 2 bytes: leading dimension multiple of SW
    const bool v_c32 = nv == SW && f_c32 && !((g.ldc | out_delta | g.cseg_stride) & 3) && !((uintptr_t)g.C32 & 15);
    const bool v_res = nv == SW && f_res && !(g.ldres & 3) && !((uintptr_t)g.residual & 15);
    const bool v_c16 = nv == SW && f_c16 && !(g.ldc16 & A16) && !((uintptr_t)g.C16 & (2 * SW - 1));
    const bool v_msk = nv == SW && f_mask && !(g.ldmask & A16) && !((uintptr_t)g.mask & (2 * SW - 1));
    const bool v_pe = nv == SW && f_pe && !(g.N & 3) && !((uintptr_t)g.pe & 15);
    auto ldf = [](const float* p, float (&o)[SW]) {
#pragma unroll
        for (int h = 0; h < SW / 4; ++h) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(p + 4 * h);
            o[4 * h] = t[0]; o[4 * h + 1] = t[1]; o[4 * h + 2] = t[2]; o[4 * h + 3] = t[3];
        }
    };
    // strips are handled NB at a time (the batch's inputs must fit the registers the accumulators just vacated)
    constexpr int NB = NCH < 4 ? NCH : 4;
#pragma unroll
    for (int i0 = 0; i0 < NCH; i0 += NB) {
    // ---- phase 1: every optional input of the batch
    float rs[NB][SW], old[NB][SW], pev[NB][SW], mk[NB][SW];
#pragma unroll
    for (int ii = 0; ii < NB; ++ii) {
        const int i = ii, row = (tid + (i0 + ii) * NT) / CS, m = m0 + row;
#pragma unroll
        for (int e = 0; e < SW; ++e) { rs[i][e] = 0.f; old[i][e] = 0.f; pev[i][e] = 0.f; mk[i][e] = 1.f; }
        if (m >= g.M) continue;
        if (f_res) {
            const float* p = g.residual + (long)m * g.ldres + n;
            if (v_res) ldf(p, rs[i]);
            else { for (int e = 0; e < nv; ++e) rs[i][e] = p[e]; }
        }
        if (f_pe) {
            const float* p = g.pe + (long)(m % g.pe_period) * g.N + n;
            if (v_pe) ldf(p, pev[i]);
            else { for (int e = 0; e < nv; ++e) pev[i][e] = p[e]; }
        }
        if (f_mask) {
            const bf16* p = g.mask + (long)m * g.ldmask + n;
            if (v_msk) {
                if constexpr (SW == 8) { const bf16x8 t = ld8(p); for (int e = 0; e < 8; ++e) mk[i][e] = (float)t[e]; }
                else { const bf16x4 t = *reinterpret_cast<const bf16x4*>(p); for (int e = 0; e < 4; ++e) mk[i][e] = (float)t[e]; }
            } else { for (int e = 0; e < nv; ++e) mk[i][e] = (float)p[e]; }
        }
        if (f_c32 && f_acc) {
            const float* c32 = g.C32 + out_delta + crow(g, m) + n;
            if (v_c32) ldf(c32, old[i]);
            else { for (int e = 0; e < nv; ++e) old[i][e] = c32[e]; }
        }
    }
    // ---- phase 2: arithmetic and stores
#pragma unroll
    for (int ii = 0; ii < NB; ++ii) {
        const int i = ii, row = (tid + (i0 + ii) * NT) / CS, m = m0 + row;
        if (m >= g.M) continue;
        float v[SW];
        ldf(ct + row * LDC + cs, v);
        float ks[SW];                                             // dropout keep-scales of the SW consecutive elements: one hash word per pair
#pragma unroll
        for (int h = 0; h < SW / 4; ++h) {
            float k4[4] = {1.f, 1.f, 1.f, 1.f};
            if (f_drop) dropout_scale4(seed, g.site, (uint32_t)((long)m * g.N + n + 4 * h), g.drop_p, inv_keep, k4);
            ks[4 * h] = k4[0]; ks[4 * h + 1] = k4[1]; ks[4 * h + 2] = k4[2]; ks[4 * h + 3] = k4[3];
        }
#pragma unroll
        for (int e = 0; e < SW; ++e) {
            float x = v[e] + bv[e] + pev[i][e];
            if (f_relu) x = fmaxf(x, 0.f);
            if (f_mask) x = mk[i][e] > 0.f ? x * g.mask_scale : 0.f;
            if (f_drop) x *= ks[e];
            v[e] = x + rs[i][e] + old[i][e];
        }
        if (f_c32) {
            float* c32 = g.C32 + out_delta + crow(g, m) + n;
            if (v_c32) {
#pragma unroll
                for (int h = 0; h < SW / 4; ++h) *reinterpret_cast<f32x4*>(c32 + 4 * h) = f32x4{v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]};
            } else { for (int e = 0; e < nv; ++e) c32[e] = v[e]; }
        }
        if (f_c16) {
            bf16* p = g.C16 + (long)m * g.ldc16 + n;
            if (v_c16) {
                if constexpr (SW == 8) { bf16x8 t; for (int e = 0; e < 8; ++e) t[e] = (bf16)v[e]; st8(p, t); }
                else { bf16x4 t; for (int e = 0; e < 4; ++e) t[e] = (bf16)v[e]; *reinterpret_cast<bf16x4*>(p) = t; }
            } else { for (int e = 0; e < nv; ++e) p[e] = (bf16)v[e]; }
        }
    }
    }
}

template <int BM, int BN, bool RM, int EPI>
__device__ __forceinline__ void gemm_body(const GemmArgs& g, const int bx, const int by, const int bz) {
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int FM = WM / 16, FN = WN / 16;
    using TA = typename std::conditional<RM, TileRM<BM>, TileNT<BM>>::type;
    using TB = typename std::conditional<RM, TileRM<BN>, TileNT<BN>>::type;
    using SA = typename std::conditional<RM, StagerRM<BM>, StagerNT<BM>>::type;
    using SB = typename std::conditional<RM, StagerRM<BN>, StagerNT<BN>>::type;
    constexpr int LDC = BN + 4;                                // fp32 output tile row stride (epilogue staging)
    constexpr size_t TILE_BYTES = 2 * sizeof(TA) + 2 * sizeof(TB), OUT_BYTES = sizeof(float) * BM * LDC;
    __shared__ __attribute__((aligned(16))) char smem[TILE_BYTES > OUT_BYTES ? TILE_BYTES : OUT_BYTES];
    TA* sa_ = reinterpret_cast<TA*>(smem);
    TB* sb_ = reinterpret_cast<TB*>(smem + 2 * sizeof(TA));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = by * BM, n0 = bx * BN;

    // register ring of DEPTH k-tiles: global loads run DEPTH tiles ahead of the MFMAs (a lone workgroup on a CU has
    // nothing else to hide the L2/HBM latency with -- this is what the small-M decoder GEMMs are made of)
    constexpr int DEPTH = 2;
    SA sa[DEPTH]; SB sb[DEPTH];
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fused bias gradient (wgrad form only): colsum[m] = sum_k A(m,k), produced by the n-tile-0 workgroups
    const bool do_colsum = RM && g.colsum && bx == 0;
    float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // split-K (wgrad form): this workgroup reduces rows [kbeg, kend) of the reduction index
    int kbeg = 0, kend = g.K;
    long out_delta = 0;
    if (RM && g.split_k > 1) {
        const int per = ((g.K + g.split_k - 1) / g.split_k + BK - 1) / BK * BK;
        kbeg = bz * per;
        kend = kbeg + per < g.K ? kbeg + per : g.K;
        if (bz > 0) out_delta = g.split_delta + (long)(bz - 1) * g.split_stride;
    }
    const int nk = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {                       // (tiles past the end are clamped reads, never stored)
        sa[d].load(g.A, g.lda, m0, kbeg + d * BK, g.M, kend, tid);
        sb[d].load(g.B, g.ldb, n0, kbeg + d * BK, g.N, kend, tid);
    }
    if constexpr (RM) { if (do_colsum) sa[0].accumulate(csum); }
    sa[0].store(sa_[0].d, tid);
    sb[0].store(sb_[0].d, tid);
    __syncthreads();

    for (int kt0 = 0; kt0 < nk; kt0 += DEPTH) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            const int kt = kt0 + u;
            if (kt >= nk) break;
            static_assert(DEPTH % 2 == 0, "kt0 is a multiple of DEPTH: the LDS buffer parity must be static");
            const int cur = u & 1;                     // == kt & 1, but a compile-time constant (fragment addresses fold to immediates)
            // tile kt already sits in LDS, so ring slot u is free: refill it with tile kt + DEPTH
            sa[u].load(g.A, g.lda, m0, kbeg + (kt + DEPTH) * BK, g.M, kend, tid);
            sb[u].load(g.B, g.ldb, n0, kbeg + (kt + DEPTH) * BK, g.N, kend, tid);
#pragma unroll
            for (int kc = 0; kc < BK / 32; ++kc) {
                bf16x8 af[FM], bfr[FN];
                if (RM) {
#pragma unroll
                    for (int i = 0; i < FM; ++i) af[i] = frag_rm<BM + 16>(sa_[cur].d, wm * WM + i * 16, lane, kc * 32);
#pragma unroll
                    for (int j = 0; j < FN; ++j) bfr[j] = frag_rm<BN + 16>(sb_[cur].d, wn * WN + j * 16, lane, kc * 32);
                } else {
                    const int kq = kc * 32 + (lane >> 4) * 8, rr = lane & 15;
#pragma unroll
                    for (int i = 0; i < FM; ++i) af[i] = ld8(&sa_[cur].d[(wm * WM + i * 16 + rr) * LDK + kq]);
#pragma unroll
                    for (int j = 0; j < FN; ++j) bfr[j] = ld8(&sb_[cur].d[(wn * WN + j * 16 + rr) * LDK + kq]);
                }
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) acc[i][j] = mma16(af[i], bfr[j], acc[i][j]);
            }
            if (kt + 1 < nk) {
                const int nx = (u + 1) % DEPTH;        // static after the full unroll
                if constexpr (RM) { if (do_colsum) sa[nx].accumulate(csum); }
                sa[nx].store(sa_[cur ^ 1].d, tid);
                sb[nx].store(sb_[cur ^ 1].d, tid);
            }
            __syncthreads();
        }
    }
    if constexpr (RM) {
        if (do_colsum) {
            // threads tid, tid + RC, tid + 2RC ... own the same 8 rows: reduce through LDS (tile buffers are free now)
            constexpr int RC = BM / 8;
            float* red = reinterpret_cast<float*>(smem);          // [256][8] floats = 8 KB <= tile size
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 8; ++j) red[tid * 8 + j] = csum[j];
            __syncthreads();
            if (tid < BM) {
                const int rc = tid / 8, j = tid % 8;
                float sum = 0.f;
                for (int t = rc; t < 256; t += RC) sum += red[t * 8 + j];
                if (m0 + tid < g.M) {
                    const int m = m0 + tid;
                    g.colsum[out_delta + (g.cseg_rows ? (long)(m / g.cseg_rows) * g.cseg_stride + m % g.cseg_rows : (long)m)] = sum;
                }
            }
            __syncthreads();
        }
    }

    gemm_epilogue<BM, BN, EPI>(g, acc, smem, m0, n0, out_delta);
}
// XCD-aware tile order (bijective for any grid): workgroup ids are dealt round-robin to the 8 XCDs, each with a private
// L2; every XCD gets a CONTIGUOUS run of tiles (x fastest), so tiles that share an A row-block / B column-block meet in
// one L2 instead of eight.
__device__ __forceinline__ void xcd_tile(int& bx, int& by, int& bz) {
    const int gx = gridDim.x, gy = gridDim.y, total = gx * gy * (int)gridDim.z;
    const int l = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const int c = l & 7, q = total >> 3, r = total & 7;
    const int t = c * q + (c < r ? c : r) + (l >> 3);
    bx = t % gx; by = (t / gx) % gy; bz = t / (gx * gy);
}
template <int BM, int BN, bool RM, int EPI>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (g.xcd_order) xcd_tile(bx, by, bz);
    gemm_body<BM, BN, RM, EPI>(g, bx, by, bz);
}

// ---- the encoder- and decoder-row weight gradients on EIGHT-wave tiles (round 4).  What bounds the 128 x 128 four-wave launch above is
// operand intake: every tile streams a [rows x 128] panel of dY and one of X through its CU (1.21 GB of tile operands for 244 MB of
// distinct bytes, ~28 GB/s per CU), MFMA pipes busy 0.21; the CUs together take in ~8.7 TB/s from the L2s at this access shape
// (tools/wgrad_probe.py: a full round takes the same time over 8 MB of resident operands as over 49 MB), so the way to fewer microseconds
// is fewer operand bytes per FLOP.  A register-staged 128 x 256 tile on eight waves (-25 % bytes per FLOP: 169 -> 134 us on the hkust
// encoder group, whole rounds of big tiles + quarter tiles for the partial last round) was the first step and is gone; the 256 x 256
// LDS-DMA tile below (-50 %) replaced it for every group size: against it, groups of 320 / 384 / 512 such tiles (two rounds, the last
// one partial) take 225 / 219 / 252 us where the 128 x 256 form took 275 / 245 / 341.  Every output element is reduced by ONE workgroup
// over the rows in order whatever the tile, so dW is bit-identical to the four-wave kernel's; the bias gradient differs in fp32
// rounding only.  MASR_ENC_WGRAD_TILE=128 restores the four-wave launch.
// ---- 256 x 256 tiles (2 x 4 waves of 128 x 64) with LDS-DMA staging (hkust: 148 tiles for the encoder rows, 228 for the decoder rows: one
// partial round each).
// The register-staged form of this tile (rounds 3-4) spent, per 64 reduction rows and thread, 8 global loads with
// ~10 vector instructions of clamping each, 8 ds_write_b128 and 64 staging registers (251 VGPRs in all) beside its 64 MFMAs and ran at
// 2.06 us per 64 rows against 0.86 of MFMA time (132 us in the engine).  Here the tiles go L2 -> LDS by global_load_lds_dwordx4 (no
// VGPRs, no LDS stores) into a ring of four 32-row stages; a super-step multiplies two of them while the next two travel (114 us).
// LDS image of an operand tile [32 k][256 cols]: 4 x 4 sub-tiles of [8 k][64 cols] = 1 KiB = ONE DMA instruction (lane l -> k row l >> 3,
// 16-byte chunk l & 7); inside a sub-tile, chunk c of row r sits at chunk position c ^ r (applied to the SOURCE address: the image must be
// lane-linear).  The transposing fragment read (ds_read_b64_tr_b16: lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3)
// then finds the 8 rows of a 32-lane half in 8 different chunk positions, odd and even rows in different bank halves: conflict-free
// (SQ_LDS_BANK_CONFLICT 0).  Reduction rows past the end are zeroed in LDS after they land; rows / columns past the matrix edge are
// clamped re-reads whose products are never stored.
// What each piece bought, same box, engine launch / 128-tile probe (tools/wgrad_probe.py): DMA through the builtin, two 64-row stages,
// __syncthreads: 180 us (the compiler drains the DMAs in front of the first fragment read, see glds16); asm DMA + four-stage ring, barrier per
// 32 rows: 121 / 121; super-steps of 64 rows: - / 118; fragment reads three steps ahead + bias MFMAs spread over the waves: 122 / 110;
// DMA issue spread over steps 0..7: 122 / 99; L2 prefetch three super-steps ahead: 114 / 104; 16-byte dW stores through LDS: 114 / 102.
// The MFMA floor is 54 us; the loop alone (no DMA) runs at 87, the DMAs alone at 50.
constexpr int W16_BK = 32, W16_NST = 4;                               // reduction rows per k tile, LDS stages (NST - 1 tiles in flight)
constexpr int W16_OP = W16_BK * 256 * 2, W16_STAGE = 2 * W16_OP;    // one operand tile 16 KiB, one stage (dY | X) 32 KiB
__device__ __forceinline__ int w16_off(int k, int col) {               // byte offset of element (k, col) inside an operand tile image
    const int r = k & 7, c = (col & 63) >> 3;
    return ((k >> 3) * 4 + (col >> 6)) * 1024 + r * 128 + ((c ^ r) << 4) + ((col & 7) << 1);
}
// fragment of 16 columns starting at r0 (a multiple of 16), k = 0 .. 31.  The lane-dependent part of the address is one of FOUR values
// (`loff[(r0 >> 4) & 3]`: which 16 columns of a 64-column sub-tile) shared by both operands and both k halves -- everything else is a
// constant added to it (kept explicit: left to the compiler the loop-invariant addresses were hoisted into registers and spilled)
__device__ __forceinline__ bf16x8 w16_frag(const char* tile, int r0, const int (&loff)[4]) {
    typedef __attribute__((address_space(3))) bf16x4 lds_b4;
    const char* a = tile + loff[(r0 >> 4) & 3] + (r0 >> 6) * 1024;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)a);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(a + 2 * 4 * 1024));       // 16 k rows = two sub-tile rows down
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}
__device__ __forceinline__ void wgrad16_tile(const WgradDesc& d, const int m0, const int n0, const bool do_colsum, char* smem) {
    constexpr int FM = 8, FN = 4;                                        // wave tile 128 x 64
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 2, wn = wave & 3;
    const int M = d.N, N = d.K, K = d.rows;
    const int nk = (K + W16_BK - 1) / W16_BK;
    // this wave's DMA pieces of a k tile: waves 0-3 the dY tile, 4-7 the X tile; k rows 8 (w & 3) .. + 7, the four 64-column blocks;
    // lane -> (row, chunk) of the sub-tile, source chunk = chunk ^ row
    const int sop = wave >> 2, sr = lane >> 3, sc = (lane & 7) ^ sr;
    const bf16* src[4];
    const long sld = sop ? d.ldx : d.lddy;
    {
        const int last = ((sop ? N : M) + 7) / 8 * 8 - 8;                // last 16-byte chunk of a row (the padded row exists)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int c = (sop ? n0 : m0) + j * 64 + sc * 8;
            c = c < last ? c : last;
            src[j] = (sop ? d.x : d.dy) + c;
        }
    }
    const int krow = (wave & 3) * 8 + sr;
    const unsigned dst0 = lds_addr(smem) + sop * W16_OP + (wave & 3) * 4 * 1024;
    auto stage = [&](int kt) {                                           // (rows past the end of the reduction are clamped re-reads: zeroed after landing)
        const unsigned st = dst0 + (kt % W16_NST) * W16_STAGE;
        long row = (long)kt * W16_BK + krow;
        row = row < K ? row : K - 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(src[j] + row * sld, st + j * 1024);
    };
    f32x4 acc[FM][FN];
    // bias gradient = row sums of the dY tile: dY fragments x ones on the matrix cores, each of the four waves of a row half taking two
    // of its eight fragments (2 wn, 2 wn + 1), re-read at the end of a super-step (one uniform branch there, none inside the pipeline)
    f32x4 accs[2];
    accs[0] = accs[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;
    int loff[4], csoff[2];
    {
        const int g = lane >> 4, ii = lane & 15, q = ii >> 2, p = ii & 3, rowq = 4 * g + q, r = rowq & 7;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) loff[c4] = (rowq >> 3) * 4 * 1024 + r * 128 + (((c4 * 2 + (p >> 1)) ^ r) << 4) + (p & 1) * 8;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int fi = 2 * wn + e;                                   // fragment of the row half: columns wm * 128 + fi * 16 of the dY tile
            csoff[e] = (rowq >> 3) * 4 * 1024 + r * 128 + ((((fi & 3) * 2 + (p >> 1)) ^ r) << 4) + (p & 1) * 8 + (wm * 2 + (fi >> 2)) * 1024;
        }
    }
    // super-steps of two k tiles (64 reduction rows, 64 MFMAs per wave between barriers): tiles 2s and 2s + 1 are multiplied while
    // 2s + 2 and 2s + 3 (issued right after the barrier, into the buffers of the super-step before) travel
    const int ns = (nk + 1) / 2;
    // L2 prefetch W16_PD super-steps ahead (operands that no cache holds take longer than one super-step to arrive): one dword load
    // per thread and super-step, one 128-byte line each (2 operands x 64 rows x 4 lines), never read.  It is issued AFTER the
    // super-step's DMAs and the loop waits with vmcnt(1): the youngest prefetch stays in flight across the barrier.
    constexpr int W16_PD = 3;                                           // (2..4 alike, 6 slower, none: +8 % on the engine's launch)
    const bf16* pf_src; long pf_ld; int pf_row;
    {
        const int op = tid >> 8, line = tid & 3;
        pf_row = (tid >> 2) & 63;
        const int last = ((op ? N : M) + 7) / 8 * 8 - 8;
        int col = (op ? n0 : m0) + line * 64;
        col = col < last ? col : last;
        pf_src = (op ? d.x : d.dy) + col; pf_ld = op ? d.ldx : d.lddy;
    }
    int pf_sink = 0;
    auto prefetch = [&](int s2) {
        long row = (long)s2 * 2 * W16_BK + pf_row;
        row = row < K ? row : K - 1;
        asm volatile("global_load_dword %0, %1, off" : "+v"(pf_sink) : "v"(pf_src + row * pf_ld) : "memory");
    };
    stage(0); stage(1);
    prefetch(W16_PD - 1);
    for (int ss = 0; ss < ns; ++ss) {
        asm volatile("s_waitcnt vmcnt(1)" ::: "memory");                 // this wave's pieces of tiles 2s, 2s + 1 (the prefetch behind them may still travel)
        __builtin_amdgcn_s_barrier();                                    // everyone's; and everyone is done reading the other two buffers
        // the next two tiles' DMAs go out one per step during steps 0..7 (as a burst in front of the first fragment reads they cost
        // every wave ~800 cycles of issue before its first MFMA)
        long roff[2]; unsigned sdst[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kt = 2 * ss + 2 + h;
            long row = (long)kt * W16_BK + krow;
            row = row < K ? row : K - 1;
            roff[h] = row * sld;
            sdst[h] = dst0 + (kt % W16_NST) * W16_STAGE;
        }
        char* t0 = smem + (ss & 1) * 2 * W16_STAGE;
        if (ss == ns - 1 && K - 2 * ss * W16_BK < 2 * W16_BK) {
            // the last super-step holds fewer than 64 valid rows: zero the others (clamped re-reads of the last row), once
            const int valid = K - 2 * ss * W16_BK;
            for (int c = tid; c < 2 * 2 * W16_BK * 32; c += 512) {
                const int h = c / (2 * W16_BK * 32), op = (c / (W16_BK * 32)) & 1, k = (c >> 5) % W16_BK, ch = c & 31;
                if (h * W16_BK + k >= valid) *reinterpret_cast<bf16x8*>(t0 + h * W16_STAGE + op * W16_OP + w16_off(k, ch * 8)) = zero8();
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        // 16 steps of one dY fragment x four X fragments (two k tiles of eight); fragment reads run three steps ahead through a ring of
        // four, the second tile's X fragments arrive during steps 4..7
        const char* ab0 = t0; const char* bb0 = t0 + W16_OP;
        const char* ab1 = t0 + W16_STAGE; const char* bb1 = ab1 + W16_OP;
        bf16x8 b0[FN], b1[FN], ar[4];
#pragma unroll
        for (int j = 0; j < FN; ++j) b0[j] = w16_frag(bb0, wn * 64 + j * 16, loff);
#pragma unroll
        for (int t = 0; t < 3; ++t) ar[t] = w16_frag(ab0, wm * 128 + t * 16, loff);
        // (the order is pinned: left alone, the scheduler folds the ring into one fragment that is read and used at once)
#define W16_STEP(t)                                                                                                                    \
        {                                                                                                                               \
            if ((t) + 3 < 16) ar[((t) + 3) & 3] = w16_frag(((t) + 3) >> 3 ? ab1 : ab0, wm * 128 + (((t) + 3) & 7) * 16, loff);        \
            if ((t) >= 4 && (t) < 8) b1[((t) - 4) & 3] = w16_frag(bb1, wn * 64 + (((t) - 4) & 3) * 16, loff);                           \
            _Pragma("unroll") for (int j = 0; j < FN; ++j) acc[(t) & 7][j] = mma16(ar[(t) & 3], (t) >> 3 ? b1[j] : b0[j], acc[(t) & 7][j]); \
            if ((t) < 8) glds16(src[(t) & 3] + roff[(t) >> 2], sdst[(t) >> 2] + ((t) & 3) * 1024);                                    \
            if ((t) == 8) prefetch(ss + W16_PD);                                                                                       \
            __builtin_amdgcn_sched_group_barrier(0x100, ((t) + 3 < 16 ? 2 : 0) + ((t) >= 4 && (t) < 8 ? 2 : 0), 0);                     \
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                                          \
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 14, 0);
        W16_STEP(0) W16_STEP(1) W16_STEP(2) W16_STEP(3) W16_STEP(4) W16_STEP(5) W16_STEP(6) W16_STEP(7)
        W16_STEP(8) W16_STEP(9) W16_STEP(10) W16_STEP(11) W16_STEP(12) W16_STEP(13) W16_STEP(14) W16_STEP(15)
#undef W16_STEP
        if (do_colsum) {
            typedef __attribute__((address_space(3))) bf16x4 lds_b4;
#pragma unroll
            for (int kc = 0; kc < 2; ++kc)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const char* a = (kc ? ab1 : ab0) + csoff[e];
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)a);
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(a + 2 * 4 * 1024));
                    bf16x8 f;
                    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
                    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
                    accs[e] = mma16(f, ones, accs[e]);
                }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf_sink)::"memory");        // the read-ahead past the last tile lands before the LDS is released
    const int fq = lane >> 4, fr = lane & 15;
    if (do_colsum && fr == 0) {
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int m = m0 + wm * 128 + (2 * wn + e) * 16 + fq * 4 + r; if (m < M) d.db[m] = accs[e][r]; }
    }
    if ((N & 3) == 0 && (reinterpret_cast<size_t>(d.dW) & 15) == 0) {
        // dW rows in whole 256-byte runs: the wave's 128 x 64 block goes through LDS 32 rows at a time (row pitch 68 floats: the four row
        // groups of a fragment land in different banks) and leaves as 16-byte stores, 16 lanes per row (as dword stores straight from
        // the fragments the 148 tiles' 39 MB took ~7 us of store issue at the end of the launch)
        __builtin_amdgcn_s_barrier();                                    // every wave's read-ahead has landed: the LDS is free
        float* st = reinterpret_cast<float*>(smem) + wave * (32 * 68);
#pragma unroll
        for (int c = 0; c < FM / 2; ++c) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < FN; ++j) st[(ii * 16 + fq * 4 + r) * 68 + j * 16 + fr] = acc[2 * c + ii][j][r];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int row = k * 4 + fq, m = m0 + wm * 128 + c * 32 + row, n = n0 + wn * 64 + fr * 4;
                const f32x4 v = *reinterpret_cast<const f32x4*>(st + row * 68 + fr * 4);
                if (m < M && n < N) *reinterpret_cast<f32x4*>(d.dW + (long)m * N + n) = v;
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wm * 128 + i * 16 + fq * 4 + r;
            if (m >= M) continue;
            float* row = d.dW + (long)m * N;
#pragma unroll
            for (int j = 0; j < FN; ++j) { const int n = n0 + wn * 64 + j * 16 + fr; if (n < N) row[n] = acc[i][j][r]; }
        }
}
__global__ __launch_bounds__(512) void gemm_wgrad_grouped16_kernel(WgradGroup grp, int n_first) {
    __shared__ __attribute__((aligned(1024))) char smem[W16_NST * W16_STAGE];
    // XCD-contiguous runs of the tile list (workgroup ids are dealt round-robin to the 8 XCDs; bijective for any tile count): the tiles
    // that share a dY / X panel meet in ONE L2 -- 507 MB of HBM traffic without this, 2.1 x the algorithmic bytes.
    // n_first > 0: the list is two segments, the first n_first tiles (long reductions: the encoder rows) and the rest (short ones: the
    // decoder rows).  Each XCD takes a contiguous run of EACH segment and starts with its long tiles (workgroups are dispatched in id
    // order): the short tiles then fill the CUs the long ones leave idle instead of being a launch of their own.
    const int total = gridDim.x, l = blockIdx.x, xc = l & 7, j = l >> 3;
    auto cnt = [](int n, int x) { return (n >> 3) + (x < (n & 7) ? 1 : 0); };          // what XCD x gets of n items dealt round-robin
    auto start = [](int n, int x) { const int r = n & 7; return x * (n >> 3) + (x < r ? x : r); };
    int t_;
    if (n_first <= 0 || total - n_first < 8) t_ = start(total, xc) + j;
    else {
        const int ca = cnt(n_first, xc);
        if (j < ca) t_ = start(n_first, xc) + j;
        else {
            int sb = 0;                                                   // second-segment tiles of the XCDs before this one
            for (int x = 0; x < xc; ++x) sb += cnt(total, x) - cnt(n_first, x);
            t_ = n_first + sb + (j - ca);
        }
    }
    int p = 0;
    while (p + 1 < grp.n && t_ >= grp.p[p + 1].tile_start) ++p;
    const WgradDesc& d = grp.p[p];
    const int t = t_ - d.tile_start, tiles_x = (d.K + 255) / 256;
    const int bx = t % tiles_x, by = t / tiles_x;
    wgrad16_tile(d, by * 256, bx * 256, d.db != nullptr && bx == 0, smem);
}

#undef HAS

// ---- NT form with LDS-DMA staging (global_load_lds_dwordx4): tiles go HBM/L2 -> LDS without touching VGPRs, the next
// k-tile lands while the current one feeds the MFMAs, and no prefetch ring / vmcnt bookkeeping is left to the compiler.
// The LDS image must be lane-linear per wave-instruction (1 KiB contiguous), so rows are unpadded (128 B) and the
// bank-conflict fix is an XOR swizzle applied to the SOURCE address and to the fragment read: 16-byte chunk c of tile
// row r sits at chunk position c ^ (r & 7).  Needs K % 64 == 0 (out-of-range rows are clamped, never read back).
template <int BM, int BN, int EPI>
__global__ __launch_bounds__(256) void gemm_glds_kernel(GemmArgs g) {
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int FM = WM / 16, FN = WN / 16;
    constexpr int LDC = BN + 4;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, BUF = A_BYTES + B_BYTES;
#ifndef MASR_GLDS_NBUF_SMALL
#define MASR_GLDS_NBUF_SMALL 3
#endif
#ifndef MASR_GLDS_NBUF_MID
#define MASR_GLDS_NBUF_MID 3
#endif
    constexpr int NBUF = (BM * BN >= 128 * 128) ? 2 : (BM * BN >= 128 * 64) ? MASR_GLDS_NBUF_MID : MASR_GLDS_NBUF_SMALL;   // 128x128: 2 x 32 KB keeps two workgroups per CU
    constexpr size_t OUT_BYTES = sizeof(float) * BM * LDC;
    __shared__ __attribute__((aligned(16))) char smem[NBUF * BUF > OUT_BYTES ? NBUF * BUF : OUT_BYTES];
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int bx_ = blockIdx.x, by_ = blockIdx.y, bz_ = blockIdx.z;
    if (g.xcd_order) xcd_tile(bx_, by_, bz_);
    const int m0 = by_ * BM, n0 = bx_ * BN;
    // split_k > 1 (gridDim.z workgroups per tile): workgroup z multiplies the z-th part of the reduction and writes its fp32 partial product to
    // C32 + z * split_stride -- no combine pass: the LayerNorm that follows sums the partials (rowops.hip LnSumArgs)
    const int nk = g.K / 64 / (int)gridDim.z, kt0 = bz_ * nk;

    // per-thread source rows are fixed over the k loop
    constexpr int ACH = BM * 8 / 256, BCH = BN * 8 / 256;
    const bf16* asrc[ACH]; const bf16* bsrc[BCH];
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
        const int c = tid + i * 256, r = c >> 3, ch = (c & 7) ^ (r & 7);
        const int row = m0 + r < g.M ? m0 + r : g.M - 1;
        asrc[i] = g.A + (long)row * g.lda + ch * 8;
    }
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
        const int c = tid + i * 256, r = c >> 3, ch = (c & 7) ^ (r & 7);
        const int row = n0 + r < g.N ? n0 + r : g.N - 1;
        bsrc[i] = g.B + (long)row * g.ldb + ch * 8;
    }
    auto stage = [&](int buf, int kt) {
        char* ab = smem + buf * BUF;
        char* bb = ab + A_BYTES;
#pragma unroll
        for (int i = 0; i < ACH; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t*)(asrc[i] + (kt0 + kt) * 64), (lptr_t*)(ab + (wave * 64 + i * 256) * 16), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < BCH; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t*)(bsrc[i] + (kt0 + kt) * 64), (lptr_t*)(bb + (wave * 64 + i * 256) * 16), 16, 0, 0);
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int rr = lane & 15, q = lane >> 4;
    auto compute = [&](int buf) {
        const bf16* ab = reinterpret_cast<const bf16*>(smem + buf * BUF);
        const bf16* bb = reinterpret_cast<const bf16*>(smem + buf * BUF + A_BYTES);
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            bf16x8 af[FM], bfr[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int row = wm * WM + i * 16 + rr;
                af[i] = ld8(ab + row * 64 + (((kc * 4 + q) ^ (row & 7)) * 8));
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int row = wn * WN + j * 16 + rr;
                bfr[j] = ld8(bb + row * 64 + (((kc * 4 + q) ^ (row & 7)) * 8));
            }
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = mma16(af[i], bfr[j], acc[i][j]);
        }
    };
    if constexpr (NBUF == 2) {
        stage(0, 0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
            compute(cur);
            __syncthreads();                               // (emits vmcnt(0): the next tile has landed) + everyone is done reading
        }
    } else {
        // three LDS buffers, two tiles in flight: a COUNTED wait leaves tile kt+1's DMAs outstanding across the (raw)
        // barrier; __syncthreads() would drain them (its fence emits vmcnt(0) while an LDS-DMA is pending)
        constexpr int LT = ACH + BCH;                      // DMA instructions per thread per tile
        static_assert(LT == 4 || LT == 6, "unexpected tile geometry");
        stage(0, 0);
        if (nk > 1) stage(1, 1);
        int cur = 0;
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) {
                if constexpr (LT == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();                  // tile kt landed for every wave; buffer (kt-1)%3 is no longer read
            if (kt + 2 < nk) stage(cur == 0 ? 2 : cur - 1, kt + 2);
            compute(cur);
            cur = cur == 2 ? 0 : cur + 1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // all fragment reads done before the epilogue reuses the LDS
    }
    gemm_epilogue<BM, BN, EPI>(g, acc, smem, m0, n0, (long)bz_ * g.split_stride);
}

// ---- The same NT tile with the two jobs on different waves (round 5): waves 4-7 only stream (LDS-DMA into a ring of NS stages, counted
// vmcnt), waves 0-3 only read fragments and multiply (software-pipelined by half k-steps: the next half's ds_reads fly under the current
// MFMAs); ONE barrier per k-step is the only synchronisation, and the multiplying waves never execute a vector-memory instruction or wait for
// one.  Why: taken apart, gemm_glds_kernel<128, 64> spends 0.32 us per 64-deep k-step in the DMA alone (24 KB = 75 GB/s per CU, the L2-hit
// ceiling) and 0.30 in reads + MFMAs alone, but 0.43 with both in the same four waves (issue, counted wait and barrier of the stream sit in
// every multiplying wave's k-step); ring depth and read pipelining inside those waves changed nothing.  Split: 0.31 per k-step -- FFN2
// (4000 x 512 x 2048, plain epilogue) 19.4 -> 14.9 us, q/k/v dgrad 16.2 -> 12.3, vgg2enc 24.4 -> 22.4 (vendor library, graph-replayed:
// 15.8 / - / 19.6).  One workgroup per CU (96 KB of ring): for grids of at most one tile per CU, i.e. the 128 x 64 launches of the
// encoder rows; the 128 x 128 launches (K = 512: eight k-steps per tile, prologue and epilogue are half the tile's life) need their second
// workgroup per CU more than the split (q/k/v 17.3 -> 21.8 as one ring workgroup per CU).
template <int BM, int BN, int EPI, int NS>
__global__ __launch_bounds__(512) void gemm_ring_kernel(GemmArgs g) {
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int FM = WM / 16, FN = WN / 16;
    constexpr int LDC = BN + 4;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, BUF = A_BYTES + B_BYTES;
    constexpr size_t OUT_BYTES = sizeof(float) * BM * LDC;
    __shared__ __attribute__((aligned(16))) char smem[NS * BUF > OUT_BYTES ? NS * BUF : OUT_BYTES];
    typedef __attribute__((address_space(1))) const void gptr_t;
    typedef __attribute__((address_space(3))) void lptr_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx_ = blockIdx.x, by_ = blockIdx.y, bz_ = blockIdx.z;
    if (g.xcd_order) xcd_tile(bx_, by_, bz_);
    const int m0 = by_ * BM, n0 = bx_ * BN;
    const int nk = g.K / 64 / (int)gridDim.z, kt0 = bz_ * nk;
    constexpr int ACH = BM * 8 / 256, BCH = BN * 8 / 256, LT = ACH + BCH;
    static_assert(LT * (NS - 1) < 64, "vmcnt range");
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (wave >= 4) {
        // ---- loader waves
        const int lt = tid - 256, lw = wave - 4;
        const bf16* asrc[ACH]; const bf16* bsrc[BCH];
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            const int c = lt + i * 256, r = c >> 3, ch = (c & 7) ^ (r & 7);
            const int row = m0 + r < g.M ? m0 + r : g.M - 1;
            asrc[i] = g.A + (long)row * g.lda + ch * 8 + (long)kt0 * 64;
        }
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            const int c = lt + i * 256, r = c >> 3, ch = (c & 7) ^ (r & 7);
            const int row = n0 + r < g.N ? n0 + r : g.N - 1;
            bsrc[i] = g.B + (long)row * g.ldb + ch * 8 + (long)kt0 * 64;
        }
        auto stage = [&](int buf, int kt) {
            char* ab = smem + buf * BUF;
            char* bb = ab + A_BYTES;
#pragma unroll
            for (int i = 0; i < ACH; ++i)
                __builtin_amdgcn_global_load_lds((gptr_t*)(asrc[i] + kt * 64), (lptr_t*)(ab + (lw * 64 + i * 256) * 16), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < BCH; ++i)
                __builtin_amdgcn_global_load_lds((gptr_t*)(bsrc[i] + kt * 64), (lptr_t*)(bb + (lw * 64 + i * 256) * 16), 16, 0, 0);
        };
        auto wait_stages = [&](int rem) {                  // at most `rem` stages' DMAs of this wave still outstanding
            if (rem >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LT * 3 < 64 ? LT * 3 : 63) : "memory");
            else if (rem == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LT * 2) : "memory");
            else if (rem == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LT * 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        // barrier j (j = 0 .. nk - 1) tells the multiplying waves that stage j has landed; passing it also means they are done with stage j - 1
#pragma unroll
        for (int s0 = 0; s0 < NS - 1; ++s0) if (s0 < nk) stage(s0, s0);
        int issued = nk < NS - 1 ? nk : NS - 1;
        for (int j = 0; j < nk; ++j) {
            const int rem = issued - (j + 1);
            wait_stages(rem > NS - 2 ? NS - 2 : rem);
            __builtin_amdgcn_s_barrier();
            if (issued < nk) { stage(issued % NS, issued); ++issued; }      // into the buffer of stage j - 1
        }
        // the tile's epilogue is shared with the multiplying waves: twice the loads and stores in flight per CU (FFN2 + bias + residual in
        // the step: 22.4 us with the four multiplying waves alone storing)
        __builtin_amdgcn_s_barrier();
        gemm_epilogue<BM, BN, EPI, 512>(g, acc, smem, m0, n0, (long)bz_ * g.split_stride, false);
        return;
    }
    // ---- multiplying waves
    const int wm = wave >> 1, wn = wave & 1;
    const int rr = lane & 15, q = lane >> 4;
    auto load_frags = [&](int buf, int kc, bf16x8 (&af)[FM], bf16x8 (&bfr)[FN]) {
        const bf16* ab = reinterpret_cast<const bf16*>(smem + buf * BUF);
        const bf16* bb = reinterpret_cast<const bf16*>(smem + buf * BUF + A_BYTES);
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int row = wm * WM + i * 16 + rr;
            af[i] = ld8(ab + row * 64 + (((kc * 4 + q) ^ (row & 7)) * 8));
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int row = wn * WN + j * 16 + rr;
            bfr[j] = ld8(bb + row * 64 + (((kc * 4 + q) ^ (row & 7)) * 8));
        }
    };
    auto mma_all = [&](const bf16x8 (&af)[FM], const bf16x8 (&bfr)[FN]) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = mma16(af[i], bfr[j], acc[i][j]);
    };
    bf16x8 a0[FM], b0[FN], a1[FM], b1[FN];
    __builtin_amdgcn_s_barrier();                          // barrier 0: stage 0 has landed
    load_frags(0, 0, a0, b0);
    int cur = 0;
    for (int kt = 0; kt + 1 < nk; ++kt) {
        load_frags(cur, 1, a1, b1);
        mma_all(a0, b0);
        const int nxt = cur == NS - 1 ? 0 : cur + 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave has finished reading stage kt
        __builtin_amdgcn_s_barrier();                      // barrier kt + 1: stage kt + 1 has landed
        load_frags(nxt, 0, a0, b0);
        mma_all(a1, b1);
        cur = nxt;
    }
    load_frags(cur, 1, a1, b1);
    mma_all(a0, b0);
    mma_all(a1, b1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // everybody is done with the ring: the epilogue stages the tile in it
    gemm_epilogue<BM, BN, EPI, 512>(g, acc, smem, m0, n0, (long)bz_ * g.split_stride, true);
}

static int gemm_ncu() {                                         // (task-slot threads call this concurrently: a thread-safe one-time initialisation)
    static const int ncu = [] {
        int dev = 0, n = 0;
        hipGetDevice(&dev); hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    return ncu;
}
template <int BM, int BN, int EPI>
void launch_glds(const GemmArgs& g, dim3 grid, hipStream_t s) {
    if constexpr (BN == 64) {
        // at most one tile per CU: the loader-wave form (one workgroup per CU).  128 x 64: the encoder-row launches with N = 512; 64 x 64: the
        // decoder-row launches of 80-240 workgroups (6.8 -> 6.2 us; with 320 workgroups the plain form's second workgroup per CU wins: 8.8 vs 10.1)
        // Ring depth of the 128 x 64 form: four stages (96 KB) with the GPU to itself (FFN2 14.9 us; single task +2 %), three (72 KB, 15.6 us)
        // when other task streams share the GPU (GemmArgs::lean): with four, the four-slot throughput FELL by 1.5 % -- a 96 KB workgroup on
        // every CU keeps the other slots' GEMM workgroups (48-72 KB) off it -- with three it is unchanged and the lone task keeps +1 %.
        if ((long)grid.x * grid.y * grid.z <= gemm_ncu()) {
            if (BM == 128 && g.lean) hipLaunchKernelGGL((gemm_ring_kernel<BM, BN, EPI, 3>), grid, dim3(512), 0, s, g);
            else hipLaunchKernelGGL((gemm_ring_kernel<BM, BN, EPI, 4>), grid, dim3(512), 0, s, g);
            return;
        }
    }
    hipLaunchKernelGGL((gemm_glds_kernel<BM, BN, EPI>), grid, dim3(256), 0, s, g);
}

template <int BM, int BN, bool RM, int EPI>
void launch_epi(const GemmArgs& g, dim3 grid, hipStream_t s) {
    hipLaunchKernelGGL((gemm_kernel<BM, BN, RM, EPI>), grid, dim3(256), 0, s, g);
}

template <int BM, int BN>
int launch_tile(const GemmArgs& g_in, hipStream_t s) {
    GemmArgs g = g_in;
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, g.split_k > 1 ? g.split_k : 1);
    g.xcd_order = (long)grid.x * grid.y * grid.z >= 64;
    const int epi = (g.bias ? E_BIAS : 0) | (g.pe ? E_PE : 0) | (g.relu ? E_RELU : 0) | (g.mask ? E_MASK : 0) |
                    (g.drop_p > 0.f ? E_DROP : 0) | (g.residual ? E_RES : 0) | (g.accumulate ? E_ACC : 0) |
                    (g.C32 ? E_C32 : 0) | (g.C16 ? E_C16 : 0);
    if (g.reduction_major) {
        if (epi == E_C32) launch_epi<BM, BN, true, E_C32>(g, grid, s);
        else launch_epi<BM, BN, true, -1>(g, grid, s);
    } else {
        const bool glds = g.K % 64 == 0;
        switch (epi & ~E_DROP) {                       // dropout stays a run-time test inside the specialised kernels
#define CASE(mask) case (mask): if (glds) launch_glds<BM, BN, (mask) | E_DROP>(g, grid, s); else launch_epi<BM, BN, false, (mask) | E_DROP>(g, grid, s); break;
            CASE(E_BIAS | E_C16)                       // q/k/v projections
            CASE(E_BIAS | E_RES | E_C32)               // attention out-proj, FFN second layer
            CASE(E_BIAS | E_RELU | E_C16)              // FFN first layer
            CASE(E_BIAS | E_PE | E_C32 | E_C16)        // vgg2enc + positional encoding
            CASE(E_BIAS | E_C32)                       // output projection
            CASE(E_C16)                                // plain dgrad
            CASE(E_MASK | E_C16)                       // dgrad through ReLU (+dropout) mask
            CASE(E_RES | E_C32)                        // dgrad joined with the residual gradient
            CASE(E_ACC | E_C32)                        // memory gradient accumulated over decoder layers
            CASE(E_C32)
#undef CASE
            default: launch_epi<BM, BN, false, -1>(g, grid, s);
        }
    }
    if (hipGetLastError() != hipSuccess) { mk_set_error("mk_gemm", "launch failed"); return -1; }
    return 0;
}

}  // namespace

int mk_gemm_wgrad_grouped(WgradGroup& grp, hipStream_t s, int first_members) {
    if (grp.n <= 0) return 0;
    for (int i = 0; i < grp.n; ++i) {
        const WgradDesc& d = grp.p[i];
        if ((d.lddy & 7) || (d.ldx & 7) || ((uintptr_t)d.dy & 15) || ((uintptr_t)d.x & 15) || d.lddy < (d.N + 7) / 8 * 8 || d.ldx < (d.K + 7) / 8 * 8) {
            mk_set_error("mk_gemm_wgrad_grouped", "operands must be 16-byte aligned with padded rows"); return -1;
        }
        // (the tile loop clamps its reduction rows to rows - 1: an empty reduction would read row -1)
        if (d.rows < 1 || d.N < 1 || d.K < 1) { mk_set_error("mk_gemm_wgrad_grouped", "rows, N and K must be >= 1"); return -1; }
    }
    {
        int t16 = 0, n_first = 0;
        for (int i = 0; i < grp.n; ++i) {
            if (i == first_members) n_first = t16;
            grp.p[i].tile_start = t16; t16 += ((grp.p[i].N + 255) / 256) * ((grp.p[i].K + 255) / 256);
        }
        hipLaunchKernelGGL(gemm_wgrad_grouped16_kernel, dim3(t16), dim3(512), 0, s, grp, first_members > 0 && first_members < grp.n ? n_first : 0);
        if (hipGetLastError() != hipSuccess) { mk_set_error("mk_gemm_wgrad_grouped", "launch failed"); return -1; }
        return 0;
    }
    // (Tried: the 128 x 128 form with LDS-DMA staging -- reduction-major tiles are lane-linear per DMA piece as they are, chunk c of row k
    // at c ^ (k & 15) for the transposing reads, zero line behind the last row, bias gradient as one more MFMA column against ones: 202 us
    // against 160 for the register-staged kernel of that size on the encoder-row launch; both are gone: the 256 x 256 tiles above run it in 111.)
    return 0;
}

int mk_gemm(const GemmArgs& g, hipStream_t s) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return 0;
    // operand contract: 16-byte vector loads along the contiguous index
    if ((g.lda & 7) || (g.ldb & 7) || ((uintptr_t)g.A & 15) || ((uintptr_t)g.B & 15)) {
        mk_set_error("mk_gemm", "A/B must be 16-byte aligned with lda/ldb multiples of 8");
        return -1;
    }
    if (!g.reduction_major && (g.K & 7)) { mk_set_error("mk_gemm", "K must be a multiple of 8"); return -1; }
    if (g.cseg_rows && (g.cseg_rows % 128 || g.M % g.cseg_rows)) { mk_set_error("mk_gemm", "cseg_rows must be a multiple of 128 dividing M"); return -1; }
    // reduction-major tiles are fetched in 8-element chunks along m / n: the padded row must exist
    if (g.reduction_major && (g.lda < (g.M + 7) / 8 * 8 || g.ldb < (g.N + 7) / 8 * 8)) {
        mk_set_error("mk_gemm", "reduction-major form needs lda >= roundup8(M), ldb >= roundup8(N)");
        return -1;
    }
    // largest tile that still yields roughly one workgroup per CU (256 CUs)
    const long z = g.split_k > 1 ? g.split_k : 1;
    if (!g.reduction_major && g.split_k > 1 &&
        (g.K % (64 * g.split_k) || g.bias || g.pe || g.relu || g.mask || g.drop_p > 0.f || g.residual || g.accumulate || !g.C32 || g.C16 || g.cseg_rows)) {
        mk_set_error("mk_gemm", "k-split NT form: K a multiple of 64 * split_k, plain fp32 partial products only (the consumer applies the epilogue)");
        return -1;
    }
    auto wgs = [&](int bm, int bn) { return (long)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn) * z; };
    constexpr long min_wgs = 192;               // (larger tiles on fewer workgroups for the task-slot mode: 96 / 40 measured 0.5 / 1.7 % slower)
    if (wgs(128, 128) >= min_wgs) return launch_tile<128, 128>(g, s);
    if (wgs(128, 64) >= min_wgs) return launch_tile<128, 64>(g, s);
    return launch_tile<64, 64>(g, s);
}
