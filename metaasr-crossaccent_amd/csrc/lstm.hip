// Bidirectional LSTM layer (forward + backward through time) for the BLSTM-P encoder of the reference's CTC configuration
// (src/modules/encoder.py:59-157 RNNP: nn.LSTM(bidirectional, batch_first) on pack_padded_sequence input, then
// Linear + tanh).
//
// Data layout: activations are batch-first row matrices [B*T][...] (row = b*T + t), as everywhere in libmasr.
//   * gate pre-activations from the input, for all timesteps at once: gx[dir] = X W_ih^T + b_ih + b_hh   (one MFMA GEMM)
//   * the recurrence, one launch per timestep for BOTH directions (blockIdx.z): z = gx[t] + h_{t-1} W_hh^T on 16x16 MFMA
//     tiles with the reduction split over the 4 waves (the decode path's skinny-GEMM scheme), then the gate math in the
//     same kernel.  For that the gate axis is stored UNIT-MAJOR (column u*4 + g instead of torch's g*H + u): the four
//     gates of a hidden unit are adjacent, so a 16-column tile holds 4 complete units.
//   * packed-sequence semantics: a sequence does not take part in steps t >= len (state kept, output row 0); the reverse
//     direction therefore starts from the zero state at each sequence's own last frame, as torch's packed LSTM does.
//   * backward through time: one launch per timestep: dh_rec = dz_{next} W_hh (MFMA) fused with the gate backward of the
//     current step; the weight gradients dW_ih, dW_hh, db and dX are big GEMMs over all timesteps afterwards.
// The hidden size is padded to a multiple of 32 in the bf16 recurrent operands (H = 360 -> 384), pads are zero.
#include "kernels.h"

namespace {

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + __expf(-x)); }

// grid (ceil(4H/16), ceil(B/16), 2); step s: direction 0 handles t = s, direction 1 handles t = T-1-s
__global__ __launch_bounds__(256) void lstm_fwd_step_kernel(LstmStepArgs a, int s) {
    __shared__ float red[4][16][17];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16, dir = blockIdx.z;
    const int t = dir == 0 ? s : a.T - 1 - s;
    const int H = a.H, G = 4 * H, KP = a.KP;
    const bf16* hprev = a.h16[dir][s & 1];
    bf16* hnext = a.h16[dir][(s & 1) ^ 1];
    int row = m0 + (lane & 15); if (row > a.B - 1) row = a.B - 1;
    int col = n0 + (lane & 15); if (col > G - 1) col = G - 1;
    const bf16* __restrict__ ap = hprev + (long)row * KP + 8 * (lane >> 4);
    const bf16* __restrict__ wp = a.whh16[dir] + (long)col * KP + 8 * (lane >> 4);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = wave * 32; k < KP; k += 128) acc = mma16(ld8(ap + k), ld8(wp + k), acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][4 * (lane >> 4) + r][lane & 15] = acc[r];
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const int m = m0 + (threadIdx.x >> 2), ul = threadIdx.x & 3, u = (n0 >> 2) + ul;     // batch row, hidden unit
    if (m >= a.B || u >= H) return;
    const int tm = threadIdx.x >> 2;
    const long r = (long)m * a.T + t;
    bf16* yrow = a.y16 + r * (2 * H) + dir * H + u;
    if (t >= a.lens[m]) {                                       // not part of this sequence: state kept, output 0
        *yrow = (bf16)0.f;
        hnext[(long)m * KP + u] = hprev[(long)m * KP + u];
        return;
    }
    float z[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c = 4 * ul + g;
        z[g] = ((red[0][tm][c] + red[1][tm][c]) + (red[2][tm][c] + red[3][tm][c])) + a.gx[dir][r * G + 4 * u + g];
    }
    const float ig = sigm(z[0]), fg = sigm(z[1]), gg = tanhf(z[2]), og = sigm(z[3]);      // torch gate order i, f, g, o
    float* cst = a.cstate[dir] + (long)m * H + u;
    const float cn = fg * *cst + ig * gg;
    const float h = og * tanhf(cn);
    *cst = cn;
    a.c[dir][r * H + u] = cn;
    float* act = a.act[dir] + r * G + 4 * u;
    act[0] = ig; act[1] = fg; act[2] = gg; act[3] = og;
    hnext[(long)m * KP + u] = (bf16)h;
    *yrow = (bf16)h;
}

// grid (ceil(H/16), ceil(B/16), 2); step s walks the sequence against the forward order:
// direction 0 handles t = T-1-s, direction 1 handles t = s
__global__ __launch_bounds__(256) void lstm_bwd_step_kernel(LstmStepArgs a, int s) {
    __shared__ float red[4][16][17];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16, dir = blockIdx.z;
    const int H = a.H, G = 4 * H, T = a.T;
    const int t = dir == 0 ? T - 1 - s : s;
    const int tnext = dir == 0 ? t + 1 : t - 1;                 // the step processed just before this one (later in forward order)
    const int tprev = dir == 0 ? t - 1 : t + 1;                 // the step whose cell state entered this one in the forward pass
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (s > 0) {                                                // dh_rec[m][n] = sum_k dz_next[m][k] W_hh[k][n]  (k over the 4H gate axis)
        int row = m0 + (lane & 15); if (row > a.B - 1) row = a.B - 1;
        int col = n0 + (lane & 15); if (col > H - 1) col = H - 1;
        const bf16* __restrict__ ap = a.dz16[dir] + ((long)row * T + tnext) * G + 8 * (lane >> 4);
        const bf16* __restrict__ wp = a.whhT16[dir] + (long)col * G + 8 * (lane >> 4);
        for (int k = wave * 32; k < G; k += 128) acc = mma16(ld8(ap + k), ld8(wp + k), acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][4 * (lane >> 4) + r][lane & 15] = acc[r];
    __syncthreads();
    const int tm = threadIdx.x >> 4, tn = threadIdx.x & 15;
    const int m = m0 + tm, u = n0 + tn;
    if (m >= a.B || u >= H) return;
    const long r = (long)m * T + t;
    bf16* dz = a.dz16[dir] + r * G + 4 * u;
    const int len = a.lens[m];
    if (t >= len) { dz[0] = dz[1] = dz[2] = dz[3] = (bf16)0.f; return; }
    const float dh = a.dy[r * (2 * H) + dir * H + u] + ((red[0][tm][tn] + red[1][tm][tn]) + (red[2][tm][tn] + red[3][tm][tn]));
    const float* act = a.act[dir] + r * G + 4 * u;
    const float ig = act[0], fg = act[1], gg = act[2], og = act[3];
    const float cn = a.c[dir][r * H + u];
    const bool first = dir == 0 ? t == 0 : t == len - 1;        // first step of this sequence in the forward pass: c_prev = 0
    const float cp = first ? 0.f : a.c[dir][((long)m * T + tprev) * H + u];
    const float tc = tanhf(cn);
    float* dcst = a.cstate[dir] + (long)m * H + u;              // dL/dc flowing in from the step processed before (zeroed by the launcher)
    const bool last = dir == 0 ? t == len - 1 : t == 0;         // last forward step of the sequence: nothing flows in
    const float dc = dh * og * (1.f - tc * tc) + (last ? 0.f : *dcst);
    *dcst = dc * fg;
    dz[0] = (bf16)(dc * gg * ig * (1.f - ig));
    dz[1] = (bf16)(dc * cp * fg * (1.f - fg));
    dz[2] = (bf16)(dc * ig * (1.f - gg * gg));
    dz[3] = (bf16)(dh * tc * og * (1.f - og));
}

// W [4H][K] fp32 in torch row order (g*H + u) -> bf16 [4H unit-major (u*4+g)][KP] (columns >= K zero)
// pc > 0: the input features come from an NHWC conv map ([d][c], c fastest) while torch's weight columns are c*pd + d
__device__ __forceinline__ int src_col(int k, int pc, int pd) { return pc ? (k % pc) * pd + k / pc : k; }
__global__ void lstm_perm_rows_kernel(const float* __restrict__ w, bf16* __restrict__ out, int H, int K, int KP, int pc, int pd) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)4 * H * KP) return;
    const int k = (int)(i % KP), pr = (int)(i / KP), u = pr >> 2, g = pr & 3;
    out[i] = k < K ? (bf16)w[(long)(g * H + u) * K + src_col(k, pc, pd)] : (bf16)0.f;
}
// the same with permuted columns (pc > 0: layer 0 reads an NHWC conv map), one workgroup per output row: the torch row goes through LDS so
// that both the read (torch column order) and the write (our column order) are contiguous -- the element-per-thread form above reads
// with a stride of pd floats here and took 37 us per direction for the 1440 x 5376 matrix of config/blstm
__global__ __launch_bounds__(256) void lstm_perm_rows_cperm_kernel(const float* __restrict__ w, bf16* __restrict__ out, int H, int K, int KP, int pc, int pd) {
    extern __shared__ float rowbuf[];
    const int pr = blockIdx.x, u = pr >> 2, g = pr & 3;
    const float* src = w + (long)(g * H + u) * K;
    for (int k = threadIdx.x; k < K; k += 256) rowbuf[k] = src[k];
    __syncthreads();
    for (int k = threadIdx.x; k < KP; k += 256) out[(long)pr * KP + k] = k < K ? (bf16)rowbuf[src_col(k, pc, pd)] : (bf16)0.f;
}
// transposed shadows for the dgrad GEMMs, from the row-major ones: out[k][pk] = in[pk][k] for k < K ([K][R] bf16 from [R][ld]); 64 x 64 tiles
// through LDS (the element-per-thread form read the fp32 master with a stride of a whole row: 127 us per direction for layer 0's W_ih)
__global__ __launch_bounds__(256) void transpose16_kernel(const bf16* __restrict__ in, bf16* __restrict__ out, int R, int K, int ld) {
    __shared__ unsigned short tile[64][66];
    const int r0 = blockIdx.y * 64, k0 = blockIdx.x * 64, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const unsigned short* src = reinterpret_cast<const unsigned short*>(in);
    unsigned short* dst = reinterpret_cast<unsigned short*>(out);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + ty * 16 + i, k = k0 + tx;
        tile[ty * 16 + i][tx] = (r < R && k < K) ? src[(long)r * ld + k] : (unsigned short)0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int k = k0 + ty * 16 + i, r = r0 + tx;
        if (k < K && r < R) dst[(long)k * R + r] = tile[tx][ty * 16 + i];
    }
}
// bias[pk] = b_ih[g*H+u] + b_hh[g*H+u]
__global__ void lstm_bias_kernel(const float* __restrict__ bih, const float* __restrict__ bhh, float* __restrict__ out, int H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 4 * H) return;
    const int u = i >> 2, g = i & 3;
    out[i] = bih[g * H + u] + bhh[g * H + u];
}
// gradients back to torch row order: dst[(g*H+u)][k] = src[(u*4+g)][k]   (cols = 1 for the biases: dst2 gets the same values)
__global__ void lstm_unperm_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, float* __restrict__ dst2, int H, int K,
                                        int pc, int pd) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)4 * H * K) return;
    const int k = (int)(i % K), pr = (int)(i / K), u = pr >> 2, g = pr & 3;
    const float v = src[i];
    dst[(long)(g * H + u) * K + src_col(k, pc, pd)] = v;
    if (dst2) dst2[(long)(g * H + u) * K + src_col(k, pc, pd)] = v;
}
// the same with permuted columns (pc > 0), one workgroup per row through LDS: dst[(g*H+u)][c*pd + d] = src[(u*4+g)][d*pc + c]
__global__ __launch_bounds__(256) void lstm_unperm_rows_cperm_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int K, int pc, int pd) {
    extern __shared__ float rowbuf[];
    const int pr = blockIdx.x, u = pr >> 2, g = pr & 3;
    for (int k = threadIdx.x; k < K; k += 256) rowbuf[k] = src[(long)pr * K + k];
    __syncthreads();
    float* drow = dst + (long)(g * H + u) * K;
    for (int kc = threadIdx.x; kc < K; kc += 256) drow[kc] = rowbuf[(kc % pd) * pc + kc / pd];
}
// h_{t-1} as the forward pass saw it, for dW_hh = dz^T h_prev: direction 0 takes y[b][t-1][0:H], direction 1 y[b][t+1][H:2H]
// (rows outside the sequence are zero in y, so the sequence ends need no special case); columns >= H zero
__global__ void lstm_hprev_kernel(const bf16* __restrict__ y16, bf16* __restrict__ hp0, bf16* __restrict__ hp1, int B, int T, int H, int KP) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * T * KP) return;
    const int k = (int)(i % KP); const long r = i / KP; const int t = (int)(r % T);
    bf16 a = (bf16)0.f, b = (bf16)0.f;
    if (k < H) {
        if (t > 0) a = y16[(r - 1) * 2 * H + k];
        if (t + 1 < T) b = y16[(r + 1) * 2 * H + H + k];
    }
    hp0[i] = a; hp1[i] = b;
}
// fp32 [rows][C] -> bf16 [rows][Cp] (columns >= C zero)
__global__ void cast_rows_pad_kernel(const float* __restrict__ x, bf16* __restrict__ y, long rows, int C, int Cp) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * Cp) return;
    const int c = (int)(i % Cp);
    y[i] = c < C ? (bf16)x[(i / Cp) * C + c] : (bf16)0.f;
}
// y = tanh(x) (fp32 + bf16 copy); dx = dy * (1 - y^2) -> bf16
__global__ void tanh_fwd_kernel(const float* __restrict__ x, float* __restrict__ y32, bf16* __restrict__ y16, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = tanhf(x[i]);
    y32[i] = v; y16[i] = (bf16)v;
}
__global__ void tanh_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, bf16* __restrict__ dx16, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = y[i];
    dx16[i] = (bf16)(dy[i] * (1.f - v * v));
}
// rows (b, t >= lens[b]) of x [B*T][C] -> 0   (out.masked_fill(pad_mask, 0), encoder.py:298)
__global__ void mask_rows_kernel(float* __restrict__ x32, bf16* __restrict__ x16, const int* __restrict__ lens, int B, int T, int C) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * T * C) return;
    const long r = i / C;
    if ((int)(r % T) >= lens[r / T]) { if (x32) x32[i] = 0.f; if (x16) x16[i] = (bf16)0.f; }
}

}  // namespace

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? 0 : (mk_set_error(__func__, "launch failed"), -1))
static inline unsigned nblk(long n) { return (unsigned)((n + 255) / 256); }

int mk_lstm_shadows(const float* wih, const float* whh, const float* bih, const float* bhh, int H, int K, int KP_in, int KP_h,
                    bf16* wih16, bf16* wihT16, bf16* whh16, bf16* whhT16, float* bias, int pc, int pd, hipStream_t s) {
    if (pc && (size_t)K * sizeof(float) <= 64 * 1024)
        hipLaunchKernelGGL(lstm_perm_rows_cperm_kernel, dim3(4 * H), dim3(256), (size_t)K * sizeof(float), s, wih, wih16, H, K, KP_in, pc, pd);
    else
        hipLaunchKernelGGL(lstm_perm_rows_kernel, dim3(nblk((long)4 * H * KP_in)), dim3(256), 0, s, wih, wih16, H, K, KP_in, pc, pd);
    hipLaunchKernelGGL(transpose16_kernel, dim3((K + 63) / 64, (4 * H + 63) / 64), dim3(256), 0, s, wih16, wihT16, 4 * H, K, KP_in);
    hipLaunchKernelGGL(lstm_perm_rows_kernel, dim3(nblk((long)4 * H * KP_h)), dim3(256), 0, s, whh, whh16, H, H, KP_h, 0, 0);
    hipLaunchKernelGGL(transpose16_kernel, dim3((H + 63) / 64, (4 * H + 63) / 64), dim3(256), 0, s, whh16, whhT16, 4 * H, H, KP_h);
    hipLaunchKernelGGL(lstm_bias_kernel, dim3(nblk(4 * H)), dim3(256), 0, s, bih, bhh, bias, H);
    return LAUNCH_OK();
}
int mk_lstm_unperm(const float* src, float* dst, float* dst2, int H, int K, int pc, int pd, hipStream_t s) {
    if (pc && !dst2 && (size_t)K * sizeof(float) <= 64 * 1024)
        hipLaunchKernelGGL(lstm_unperm_rows_cperm_kernel, dim3(4 * H), dim3(256), (size_t)K * sizeof(float), s, src, dst, H, K, pc, pd);
    else
        hipLaunchKernelGGL(lstm_unperm_rows_kernel, dim3(nblk((long)4 * H * K)), dim3(256), 0, s, src, dst, dst2, H, K, pc, pd);
    return LAUNCH_OK();
}
int mk_lstm_hprev(const bf16* y16, bf16* hp0, bf16* hp1, int B, int T, int H, int KP, hipStream_t s) {
    hipLaunchKernelGGL(lstm_hprev_kernel, dim3(nblk((long)B * T * KP)), dim3(256), 0, s, y16, hp0, hp1, B, T, H, KP);
    return LAUNCH_OK();
}
int mk_cast_rows_pad(const float* x, bf16* y, long rows, int C, int Cp, hipStream_t s) {
    hipLaunchKernelGGL(cast_rows_pad_kernel, dim3(nblk(rows * Cp)), dim3(256), 0, s, x, y, rows, C, Cp);
    return LAUNCH_OK();
}
int mk_lstm_fwd_steps(const LstmStepArgs& a, hipStream_t s) {
    if (a.KP % 32 || a.KP < a.H) { mk_set_error("mk_lstm_fwd_steps", "padded hidden size must be a multiple of 32"); return -1; }
    for (int d = 0; d < 2; ++d) {
        if (hipMemsetAsync(a.h16[d][0], 0, sizeof(bf16) * (size_t)a.B * a.KP, s) != hipSuccess) return -1;
        if (hipMemsetAsync(a.h16[d][1], 0, sizeof(bf16) * (size_t)a.B * a.KP, s) != hipSuccess) return -1;
        if (hipMemsetAsync(a.cstate[d], 0, sizeof(float) * (size_t)a.B * a.H, s) != hipSuccess) return -1;
    }
    const dim3 grid((4 * a.H + 15) / 16, (a.B + 15) / 16, 2);
    for (int st = 0; st < a.T; ++st) hipLaunchKernelGGL(lstm_fwd_step_kernel, grid, dim3(256), 0, s, a, st);
    return LAUNCH_OK();
}
int mk_lstm_bwd_steps(const LstmStepArgs& a, hipStream_t s) {
    if ((4 * a.H) % 32) { mk_set_error("mk_lstm_bwd_steps", "4H must be a multiple of 32"); return -1; }
    for (int d = 0; d < 2; ++d)
        if (hipMemsetAsync(a.cstate[d], 0, sizeof(float) * (size_t)a.B * a.H, s) != hipSuccess) return -1;
    const dim3 grid((a.H + 15) / 16, (a.B + 15) / 16, 2);
    for (int st = 0; st < a.T; ++st) hipLaunchKernelGGL(lstm_bwd_step_kernel, grid, dim3(256), 0, s, a, st);
    return LAUNCH_OK();
}
int mk_tanh_fwd(const float* x, float* y32, bf16* y16, long n, hipStream_t s) {
    hipLaunchKernelGGL(tanh_fwd_kernel, dim3(nblk(n)), dim3(256), 0, s, x, y32, y16, n);
    return LAUNCH_OK();
}
int mk_tanh_bwd(const float* dy, const float* y, bf16* dx16, long n, hipStream_t s) {
    hipLaunchKernelGGL(tanh_bwd_kernel, dim3(nblk(n)), dim3(256), 0, s, dy, y, dx16, n);
    return LAUNCH_OK();
}
namespace {
__global__ void subsample_rows_kernel(const bf16* __restrict__ y, bf16* __restrict__ ys, int B, int Tin, int Tout, int sub, int C8) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * Tout * C8) return;
    const int c = (int)(i % C8); const long row = i / C8; const int t = (int)(row % Tout); const long b = row / Tout;
    st8(ys + i * 8, ld8(y + ((b * Tin + (long)t * sub) * C8 + c) * 8));
}
__global__ void subsample_rows_bwd_kernel(const float* __restrict__ dys, float* __restrict__ dy, int B, int Tin, int Tout, int sub, int C4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * Tin * C4) return;
    const int c = (int)(i % C4); const long row = i / C4; const int t = (int)(row % Tin); const long b = row / Tin;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (t % sub == 0 && t / sub < Tout) v = *reinterpret_cast<const f32x4*>(dys + ((b * Tout + t / sub) * C4 + c) * 4);
    *reinterpret_cast<f32x4*>(dy + i * 4) = v;
}
}  // namespace
int mk_subsample_rows(const bf16* y, bf16* ys, int B, int Tin, int Tout, int sub, int C, hipStream_t s) {
    if (C % 8 || sub < 1 || Tout != (Tin + sub - 1) / sub) { mk_set_error("mk_subsample_rows", "C % 8 == 0, Tout == ceil(Tin / sub)"); return -1; }
    const long n = (long)B * Tout * (C / 8);
    if (n == 0) return 0;
    hipLaunchKernelGGL(subsample_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, y, ys, B, Tin, Tout, sub, C / 8);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
int mk_subsample_rows_bwd(const float* dys, float* dy, int B, int Tin, int Tout, int sub, int C, hipStream_t s) {
    if (C % 4 || sub < 1 || Tout != (Tin + sub - 1) / sub) { mk_set_error("mk_subsample_rows_bwd", "C % 4 == 0, Tout == ceil(Tin / sub)"); return -1; }
    const long n = (long)B * Tin * (C / 4);
    if (n == 0) return 0;
    hipLaunchKernelGGL(subsample_rows_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dys, dy, B, Tin, Tout, sub, C / 4);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
int mk_mask_rows(float* x32, bf16* x16, const int* lens, int B, int T, int C, hipStream_t s) {
    hipLaunchKernelGGL(mask_rows_kernel, dim3(nblk((long)B * T * C)), dim3(256), 0, s, x32, x16, lens, B, T, C);
    return LAUNCH_OK();
}
