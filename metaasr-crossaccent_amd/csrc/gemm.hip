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
#include <type_traits>

namespace {

constexpr int BK = 32;
constexpr int LDK = 40;   // NT form: padded LDS row (elements), 80-byte stride

template <int ROWS>
struct TileNT { bf16 d[ROWS * LDK]; };
template <int ROWS>
struct TileRM { bf16 d[BK * (ROWS + 16)]; };   // [k][row], row stride (ROWS+16)*2 B == 8 dwords mod 64

// ---- NT stager: element (r,k) at src[r*ld + k]
template <int ROWS>
struct StagerNT {
    static constexpr int CHUNKS = ROWS * BK / 8;
    static constexpr int PT = CHUNKS / 256 > 0 ? CHUNKS / 256 : 1;
    bf16x8 regs[PT];
    __device__ __forceinline__ void load(const bf16* __restrict__ src, long ld, int r0, int k0, int nrows, int nk, int tid) {
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            int c = tid + i * 256;
            int r = c >> 2, kc = (c & 3) * 8;
            bf16x8 v = zero8();
            if (c < CHUNKS && r0 + r < nrows && k0 + kc < nk) v = ld8(src + (long)(r0 + r) * ld + k0 + kc);
            regs[i] = v;
        }
    }
    __device__ __forceinline__ void store(bf16* __restrict__ dst, int tid) {
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            int c = tid + i * 256;
            int r = c >> 2, kc = (c & 3) * 8;
            if (c < CHUNKS) st8(dst + r * LDK + kc, regs[i]);
        }
    }
};
// ---- reduction-major stager: element (r,k) at src[k*ld + r]; LDS keeps [k][r]
template <int ROWS>
struct StagerRM {
    static constexpr int LDT = ROWS + 16;
    static constexpr int RC = ROWS / 8;
    static constexpr int CHUNKS = BK * RC;
    static constexpr int PT = CHUNKS / 256 > 0 ? CHUNKS / 256 : 1;
    bf16x8 regs[PT];
    __device__ __forceinline__ void load(const bf16* __restrict__ src, long ld, int r0, int k0, int nrows, int nk, int tid) {
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            int c = tid + i * 256;
            int k = c / RC, rc = (c % RC) * 8;
            bf16x8 v = zero8();
            if (c < CHUNKS && k0 + k < nk && r0 + rc < nrows) v = ld8(src + (long)(k0 + k) * ld + r0 + rc);
            regs[i] = v;
        }
    }
    __device__ __forceinline__ void store(bf16* __restrict__ dst, int tid) {
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            int c = tid + i * 256;
            int k = c / RC, rc = (c % RC) * 8;
            if (c < CHUNKS) st8(dst + k * LDT + rc, regs[i]);
        }
    }
    // running column sums of the staged tile (each thread always owns the same 8 rows: 256 % RC == 0)
    __device__ __forceinline__ void accumulate(float (&acc)[8]) {
#pragma unroll
        for (int i = 0; i < PT; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += (float)regs[i][j];
    }
};

// fragment of 16 rows starting at r0 from a reduction-major tile (k permuted, see header)
template <int LDT>
__device__ __forceinline__ bf16x8 frag_rm(const bf16* tile, int r0, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    typedef __attribute__((address_space(3))) bf16x4 lds_b4;
    const bf16* a0 = tile + (4 * g + q) * LDT + r0 + 4 * p;
    const bf16* a1 = a0 + 16 * LDT;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)a0);
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)a1);
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

template <int BM, int BN, bool RM>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int FM = WM / 16, FN = WN / 16;
    using TA = typename std::conditional<RM, TileRM<BM>, TileNT<BM>>::type;
    using TB = typename std::conditional<RM, TileRM<BN>, TileNT<BN>>::type;
    using SA = typename std::conditional<RM, StagerRM<BM>, StagerNT<BM>>::type;
    using SB = typename std::conditional<RM, StagerRM<BN>, StagerNT<BN>>::type;
    __shared__ __attribute__((aligned(16))) TA sa_[2];
    __shared__ __attribute__((aligned(16))) TB sb_[2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    SA sa; SB sb;
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fused bias gradient (wgrad form only): colsum[m] = sum_k A(m,k), produced by the n-tile-0 workgroups
    const bool do_colsum = RM && g.colsum && blockIdx.x == 0;
    float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // split-K (wgrad form): this workgroup reduces rows [kbeg, kend) of the reduction index
    int kbeg = 0, kend = g.K;
    long out_delta = 0;
    if (RM && g.split_k > 1) {
        const int per = ((g.K + g.split_k - 1) / g.split_k + BK - 1) / BK * BK;
        kbeg = blockIdx.z * per;
        kend = kbeg + per < g.K ? kbeg + per : g.K;
        if (blockIdx.z > 0) out_delta = g.split_delta + (long)(blockIdx.z - 1) * g.split_stride;
    }
    const int nk = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
    sa.load(g.A, g.lda, m0, kbeg, g.M, kend, tid);
    sb.load(g.B, g.ldb, n0, kbeg, g.N, kend, tid);
    if constexpr (RM) { if (do_colsum) sa.accumulate(csum); }
    sa.store(sa_[0].d, tid);
    sb.store(sb_[0].d, tid);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            sa.load(g.A, g.lda, m0, kbeg + (kt + 1) * BK, g.M, kend, tid);
            sb.load(g.B, g.ldb, n0, kbeg + (kt + 1) * BK, g.N, kend, tid);
        }
        bf16x8 af[FM], bfr[FN];
        if (RM) {
#pragma unroll
            for (int i = 0; i < FM; ++i) af[i] = frag_rm<BM + 16>(sa_[cur].d, wm * WM + i * 16, lane);
#pragma unroll
            for (int j = 0; j < FN; ++j) bfr[j] = frag_rm<BN + 16>(sb_[cur].d, wn * WN + j * 16, lane);
        } else {
            const int kq = (lane >> 4) * 8, rr = lane & 15;
#pragma unroll
            for (int i = 0; i < FM; ++i) af[i] = ld8(&sa_[cur].d[(wm * WM + i * 16 + rr) * LDK + kq]);
#pragma unroll
            for (int j = 0; j < FN; ++j) bfr[j] = ld8(&sb_[cur].d[(wn * WN + j * 16 + rr) * LDK + kq]);
        }
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = mma16(af[i], bfr[j], acc[i][j]);
        if (kt + 1 < nk) {
            if constexpr (RM) { if (do_colsum) sa.accumulate(csum); }
            sa.store(sa_[cur ^ 1].d, tid);
            sb.store(sb_[cur ^ 1].d, tid);
        }
        __syncthreads();
    }
    if constexpr (RM) {
        if (do_colsum) {
            // threads tid, tid + RC, tid + 2RC ... own the same 8 rows: reduce through LDS (tile buffers are free now)
            constexpr int RC = BM / 8;
            float* red = reinterpret_cast<float*>(sa_[0].d);          // [256][8] floats = 8 KB <= tile size
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 8; ++j) red[tid * 8 + j] = csum[j];
            __syncthreads();
            if (tid < BM) {
                const int rc = tid / 8, j = tid % 8;
                float sum = 0.f;
                for (int t = rc; t < 256; t += RC) sum += red[t * 8 + j];
                if (m0 + tid < g.M) g.colsum[out_delta + m0 + tid] = sum;
            }
            __syncthreads();
        }
    }

    // epilogue: alpha -> bias -> pe -> relu -> mask -> dropout -> residual -> (accumulate) -> store
    const float inv_keep = g.drop_p > 0.f ? 1.0f / (1.0f - g.drop_p) : 1.0f;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wm * WM + i * 16 + (lane >> 4) * 4 + r;
            if (m >= g.M) continue;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int n = n0 + wn * WN + j * 16 + (lane & 15);
                if (n >= g.N) continue;
                float v = acc[i][j][r] * g.alpha;
                if (g.bias) v += g.bias[n];
                if (g.pe) v += g.pe[(long)(m % g.pe_period) * g.N + n];
                if (g.relu) v = fmaxf(v, 0.f);
                if (g.mask) v = ((float)g.mask[(long)m * g.ldmask + n] > 0.f) ? v * g.mask_scale : 0.f;
                if (g.drop_p > 0.f) v *= dropout_scale(g.seed, g.site, (uint32_t)((long)m * g.N + n), g.drop_p, inv_keep);
                if (g.residual) v += g.residual[(long)m * g.ldres + n];
                if (g.C32) {
                    float* p = g.C32 + out_delta + (long)m * g.ldc + n;
                    if (g.accumulate) v += *p;
                    *p = v;
                }
                if (g.C16) g.C16[(long)m * g.ldc16 + n] = (bf16)v;
            }
        }
    }
}

template <int BM, int BN>
int launch_tile(const GemmArgs& g, hipStream_t s) {
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, (g.reduction_major && g.split_k > 1) ? g.split_k : 1);
    if (g.reduction_major)
        hipLaunchKernelGGL((gemm_kernel<BM, BN, true>), grid, dim3(256), 0, s, g);
    else
        hipLaunchKernelGGL((gemm_kernel<BM, BN, false>), grid, dim3(256), 0, s, g);
    if (hipGetLastError() != hipSuccess) { mk_set_error("mk_gemm", "launch failed"); return -1; }
    return 0;
}

}  // namespace

int mk_gemm(const GemmArgs& g, hipStream_t s) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return 0;
    // operand contract: 16-byte vector loads along the contiguous index
    if ((g.lda & 7) || (g.ldb & 7) || ((uintptr_t)g.A & 15) || ((uintptr_t)g.B & 15)) {
        mk_set_error("mk_gemm", "A/B must be 16-byte aligned with lda/ldb multiples of 8");
        return -1;
    }
    if (!g.reduction_major && (g.K & 7)) { mk_set_error("mk_gemm", "K must be a multiple of 8"); return -1; }
    // reduction-major tiles are fetched in 8-element chunks along m / n: the padded row must exist
    if (g.reduction_major && (g.lda < (g.M + 7) / 8 * 8 || g.ldb < (g.N + 7) / 8 * 8)) {
        mk_set_error("mk_gemm", "reduction-major form needs lda >= roundup8(M), ldb >= roundup8(N)");
        return -1;
    }
    // largest tile that still yields roughly one workgroup per CU (256 CUs)
    const long z = (g.reduction_major && g.split_k > 1) ? g.split_k : 1;
    auto wgs = [&](int bm, int bn) { return (long)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn) * z; };
    if (wgs(128, 128) >= 192) return launch_tile<128, 128>(g, s);
    if (wgs(128, 64) >= 192) return launch_tile<128, 64>(g, s);
    return launch_tile<64, 64>(g, s);
}
