// Incremental (KV-cached) greedy decoding kernels for MyTransformer.recog
// (src/model/transformer_pytorch/mono_transformer_torch.py:143-176).
//
// The reference re-decodes the whole prefix at every step.  Because the target mask is causal, position i of that
// re-decode depends on tokens 0..i only, so the newest position is the only new information per step: these kernels
// compute exactly that one position per utterance against cached keys/values.
//
// Everything that changes from step to step (current length, cache slot, positional-encoding row, token slot) is read
// from ONE device integer `*step`, so the launch sequence of a step is parameter-identical for every step and the engine
// replays it as a hipGraph (engine.hip: masr_recog).
#include "kernels.h"

namespace {

// ---------------------------------------------------------------- skinny GEMM: C[m][n] = epi(sum_k A[m][k] W[n][k]), M small
// One workgroup = one 16x16 output tile; its 4 waves split the reduction (k = wave*32 + 128 i) and combine through LDS.
// Operands go global -> MFMA fragment registers directly (each lane one 16-B load per operand per step): with M <= 16-32
// rows there is no reuse to stage through LDS, the launch is bound by weight bytes and latency.
// (Folding the post-norm LayerNorms into this kernel's prologue was measured and rejected: every workgroup re-deriving
// the row statistics costs more than the ~4 us launch it saves -- 83 ms vs 50 ms per 250-step decode.)
__global__ __launch_bounds__(256) void skinny_gemm_kernel(SkinnyArgs g) {
    __shared__ float red[4][16][17];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
    int row = m0 + (lane & 15); if (row > g.M - 1) row = g.M - 1;
    int col = n0 + (lane & 15); if (col > g.N - 1) col = g.N - 1;
    const bf16* __restrict__ ap = g.A + (long)row * g.lda + 8 * (lane >> 4);
    const bf16* __restrict__ wp = g.W + (long)col * g.ldw + 8 * (lane >> 4);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int k = wave * 32;
    for (; k + 384 < g.K; k += 512) {                      // four independent load pairs in flight
        const bf16x8 a0 = ld8(ap + k), w0 = ld8(wp + k), a1 = ld8(ap + k + 128), w1 = ld8(wp + k + 128);
        const bf16x8 a2 = ld8(ap + k + 256), w2 = ld8(wp + k + 256), a3 = ld8(ap + k + 384), w3 = ld8(wp + k + 384);
        acc = mma16(a0, w0, acc); acc = mma16(a1, w1, acc); acc = mma16(a2, w2, acc); acc = mma16(a3, w3, acc);
    }
    for (; k < g.K; k += 128) acc = mma16(ld8(ap + k), ld8(wp + k), acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][4 * (lane >> 4) + r][lane & 15] = acc[r];
    __syncthreads();
    const int m = m0 + (threadIdx.x >> 4), n = n0 + (threadIdx.x & 15);
    if (m >= g.M || n >= g.N) return;
    const int tm = threadIdx.x >> 4, tn = threadIdx.x & 15;
    float v = (red[0][tm][tn] + red[1][tm][tn]) + (red[2][tm][tn] + red[3][tm][tn]);
    if (g.bias) v += g.bias[n];
    if (g.relu) v = fmaxf(v, 0.f);
    if (g.residual) v += g.residual[(long)m * g.ldres + n];
    if (g.C32) g.C32[(long)m * g.ldc + n] = v;
    if (g.C16) g.C16[(long)m * g.ldc16 + n] = (bf16)v;
}

// ---------------------------------------------------------------- one-query attention against a key/value cache
// grid (H, B), one 256-thread workgroup per (head, utterance).  klen = *step (self-attention: the cache holds step-1 rows,
// the newest key/value row is taken from knew/vnew and appended to the cache at slot step-1) or klens[b] (cross-attention).
// Scores: one key per thread (16-B loads of the bf16 row, fp32 dot).  P.V: thread (g, c) owns the 8 output columns of
// chunk c and every G-th key, so each load is 16 B and the keys of a workgroup are 256/CH-way parallel; partial sums meet
// through shuffles (within a wave) and LDS (across the 4 waves).  fp32 soft-max.
template <int HD>
__global__ __launch_bounds__(256) void attn_decode_kernel(AttnDecodeArgs a) {
    extern __shared__ float sS[];
    __shared__ float s_red[8];
    __shared__ float s_o[4][HD];
    constexpr int CH = HD / 8, G = 256 / CH, GW = 64 / CH;   // column chunks, key groups per workgroup / per wave
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int klen = a.step ? *a.step : a.klens[b];
    const bf16* __restrict__ kc = a.k + (long)b * a.kv_batch_stride + h * HD;
    const bf16* __restrict__ vc = a.v + (long)b * a.kv_batch_stride + h * HD;
    const bf16* kn = a.knew ? a.knew + (long)b * a.ldnew + h * HD : nullptr;
    const bf16* vn = a.knew ? a.vnew + (long)b * a.ldnew + h * HD : nullptr;
    if (kn && tid < HD) {                                    // append (read back only by later launches)
        const_cast<bf16*>(kc)[(long)(klen - 1) * a.ldk + tid] = kn[tid];
        const_cast<bf16*>(vc)[(long)(klen - 1) * a.ldk + tid] = vn[tid];
    }
    const float scale = rsqrtf((float)HD);
    float mx = -3.0e38f;
    if (tid < klen) {
        bf16x8 qv[CH];
        const bf16* qp = a.q + (long)b * a.ldq + h * HD;
#pragma unroll
        for (int c = 0; c < CH; ++c) qv[c] = ld8(qp + 8 * c);
        for (int j = tid; j < klen; j += 256) {
            const bf16* kr = (kn && j == klen - 1) ? kn : kc + (long)j * a.ldk;
            bf16x8 kv[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) kv[c] = ld8(kr + 8 * c);
            float d = 0.f;
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int e = 0; e < 8; ++e) d = fmaf((float)qv[c][e], (float)kv[c][e], d);
            d *= scale;
            sS[j] = d;
            mx = fmaxf(mx, d);
        }
    }
    mx = wave_max(mx);
    if (lane == 0) s_red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    float sum = 0.f;
    for (int j = tid; j < klen; j += 256) { const float p = __expf(sS[j] - mx); sS[j] = p; sum += p; }
    sum = wave_sum(sum);
    if (lane == 0) s_red[4 + wave] = sum;
    __syncthreads();                                         // probabilities and the four partial sums are visible
    sum = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);
    const int c = tid % CH, g = tid / CH;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int j = g; j < klen; j += G) {
        const bf16* vr = (vn && j == klen - 1) ? vn : vc + (long)j * a.ldk;
        const bf16x8 vv = ld8(vr + 8 * c);
        const float p = sS[j];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(p, (float)vv[e], acc[e]);
    }
#pragma unroll
    for (int o = CH; o < 64; o <<= 1)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
    static_assert(GW * CH == 64, "");
    if (lane < CH) {
#pragma unroll
        for (int e = 0; e < 8; ++e) s_o[wave][lane * 8 + e] = acc[e];
    }
    __syncthreads();
    if (tid < HD) a.o[(long)b * a.ldo + h * HD + tid] = (bf16)(((s_o[0][tid] + s_o[1][tid]) + (s_o[2][tid] + s_o[3][tid])) / sum);
}

// ---------------------------------------------------------------- per-step glue
// y[b] = table[tok] + pe[step-1], tok = sos at step 1 else the previous step's arg-max (out[step-2][b])
__global__ __launch_bounds__(256) void recog_embed_step_kernel(const int* __restrict__ step, const int* __restrict__ out,
                                                              const float* __restrict__ table, const float* __restrict__ pe,
                                                              float* __restrict__ y32, bf16* __restrict__ y16, int B, int E, int sos) {
    const int b = blockIdx.x, st = *step;
    const int tok = st == 1 ? sos : out[(long)(st - 2) * B + b];
    for (int e = threadIdx.x; e < E; e += 256) {
        const float v = table[(long)tok * E + e] + pe[(long)(st - 1) * E + e];
        y32[(long)b * E + e] = v;
        y16[(long)b * E + e] = (bf16)v;
    }
}
// out[step-1][b] = first maximal index of logits[b][:C]; block b, 256 threads
__global__ __launch_bounds__(256) void recog_argmax_step_kernel(int* step, const float* __restrict__ logits, long ld,
                                                               int* __restrict__ out, int B, int C) {
    __shared__ float smx[4]; __shared__ int sam[4];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* z = logits + (long)b * ld;
    float mx = -3.4e38f; int am = 0x7fffffff;
    for (int c = threadIdx.x; c < C; c += 256) { const float v = z[c]; if (v > mx) { mx = v; am = c; } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(mx, o, 64); const int oa = __shfl_xor(am, o, 64);
        if (om > mx || (om == mx && oa < am)) { mx = om; am = oa; }
    }
    if (lane == 0) { smx[wave] = mx; sam[wave] = am; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) if (smx[w] > mx || (smx[w] == mx && sam[w] < am)) { mx = smx[w]; am = sam[w]; }
        if (am == 0x7fffffff) am = 0;                        // all-NaN / all -inf row: torch.argmax would still return an index
        const int st = step[0];
        out[(long)(st - 1) * B + b] = am;
        // the last utterance to finish advances the step (every block has read step[0] before taking its ticket)
        __threadfence();
        if (atomicAdd(step + 1, 1) == B - 1) { step[1] = 0; step[0] = st + 1; }
    }
}
// fp32 logits of the greedy decode: z[r][c] = bias[c] + sum_k y[r][k] W[c][k] on the fp32 MASTER weights and the fp32 LayerNorm output
// (the training forward's bf16 operands are fine for a loss, but an arg-max is an index: the last projection is where two near-tied
// classes are told apart, and it is 0.4 MFLOP per row).  grid (rows, ceil(C / 64)), a wave = 16 classes, fixed summation order.
// VEC: 16-byte loads (rows of y and W 16-byte aligned); otherwise the same products in the same order from 4-byte loads
template <bool VEC>
__global__ __launch_bounds__(256) void logits_f32_kernel(const float* __restrict__ y, const float* __restrict__ W,
                                                        const float* __restrict__ bias, float* __restrict__ z, long ldz, int C, int E) {
    const int r = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = blockIdx.y * 64 + wave * 16;
    const float* yr = y + (long)r * E;
    for (int c = c0; c < c0 + 16 && c < C; ++c) {
        const float* w = W + (long)c * E;
        float acc = 0.f;
        for (int k = lane * 4; k < E; k += 256) {
            float4 a, b;
            if constexpr (VEC) { a = *reinterpret_cast<const float4*>(yr + k); b = *reinterpret_cast<const float4*>(w + k); }
            else { a = float4{yr[k], yr[k + 1], yr[k + 2], yr[k + 3]}; b = float4{w[k], w[k + 1], w[k + 2], w[k + 3]}; }
            acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
        }
        acc = wave_sum(acc);
        if (lane == 0) z[(long)r * ldz + c] = acc + bias[c];
    }
}
__global__ void recog_step_set_kernel(int* step, int value, int inc) {
    if (inc) { step[0] += 1; } else { step[0] = value; step[1] = 0; }     // step[1] = arg-max ticket counter
}

}  // namespace

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? 0 : (mk_set_error(__func__, "launch failed"), -1))

int mk_skinny_gemm(const SkinnyArgs& g, hipStream_t s) {
    if (g.K % 32 || (g.lda % 8) || (g.ldw % 8) || ((uintptr_t)g.A & 15) || ((uintptr_t)g.W & 15)) {
        mk_set_error("mk_skinny_gemm", "K % 32 == 0 and 16-byte aligned rows required"); return -1;
    }
    if (g.M <= 0 || g.N <= 0) return 0;
    hipLaunchKernelGGL(skinny_gemm_kernel, dim3((g.N + 15) / 16, (g.M + 15) / 16), dim3(256), 0, s, g);
    return LAUNCH_OK();
}
int mk_attn_decode(const AttnDecodeArgs& a, hipStream_t s) {
    const size_t lds = sizeof(float) * (size_t)a.Tk_cap;
    if (lds > 60 * 1024) { mk_set_error("mk_attn_decode", "more than 15360 keys"); return -1; }
    if ((a.ldq % 8) || (a.ldk % 8) || (a.kv_batch_stride % 8)) { mk_set_error("mk_attn_decode", "strides must be multiples of 8"); return -1; }
    const dim3 grid(a.H, a.B);
    switch (a.hd) {
        case 16: hipLaunchKernelGGL(attn_decode_kernel<16>, grid, dim3(256), lds, s, a); break;
        case 32: hipLaunchKernelGGL(attn_decode_kernel<32>, grid, dim3(256), lds, s, a); break;
        case 64: hipLaunchKernelGGL(attn_decode_kernel<64>, grid, dim3(256), lds, s, a); break;
        default: mk_set_error("mk_attn_decode", "head dim must be 16/32/64"); return -1;
    }
    return LAUNCH_OK();
}
int mk_recog_embed_step(const int* step, const int* out, const float* table, const float* pe, float* y32, bf16* y16, int B, int E, int sos,
                        hipStream_t s) {
    hipLaunchKernelGGL(recog_embed_step_kernel, dim3(B), dim3(256), 0, s, step, out, table, pe, y32, y16, B, E, sos);
    return LAUNCH_OK();
}
int mk_logits_f32(const float* y32, const float* W32, const float* bias, float* logits, long ld, int rows, int C, int E, hipStream_t s) {
    if (E % 4 != 0 || rows <= 0) { mk_set_error("mk_logits_f32", "E % 4 == 0 and rows >= 1 required"); return -1; }
    // (the projection sits at an arbitrary dword offset of the flat parameter buffer as far as this launcher knows: 16-byte loads only
    // when both operands' rows are 16-byte aligned -- always the case for the engine's models, whose d_model is a multiple of 64)
    if (!(((uintptr_t)y32 | (uintptr_t)W32) & 15)) hipLaunchKernelGGL(logits_f32_kernel<true>, dim3(rows, (C + 63) / 64), dim3(256), 0, s, y32, W32, bias, logits, ld, C, E);
    else hipLaunchKernelGGL(logits_f32_kernel<false>, dim3(rows, (C + 63) / 64), dim3(256), 0, s, y32, W32, bias, logits, ld, C, E);
    return LAUNCH_OK();
}
int mk_recog_argmax_step(int* step, const float* logits, long ld, int* out, int B, int C, hipStream_t s) {
    hipLaunchKernelGGL(recog_argmax_step_kernel, dim3(B), dim3(256), 0, s, step, logits, ld, out, B, C);
    return LAUNCH_OK();
}
int mk_recog_step_set(int* step, int value, int inc, hipStream_t s) {
    hipLaunchKernelGGL(recog_step_set_kernel, dim3(1), dim3(1), 0, s, step, value, inc);
    return LAUNCH_OK();
}
