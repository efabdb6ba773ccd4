// Internal launch API of the libmasr kernels (C++ side; the public C ABI is include/masr.h).
// Every launcher is stream-ordered, allocates nothing, never synchronises (graph-capturable).
#pragma once
#include "common.h"

// ---------------------------------------------------------------- GEMM (gemm.hip)
struct GemmArgs {
    // C[m][n] = epi( alpha * sum_k A(m,k) B(n,k) )
    const bf16* A; long lda;      // reduction_major=0: A[m*lda + k]   =1: A[k*lda + m]
    const bf16* B; long ldb;      // reduction_major=0: B[n*ldb + k]   =1: B[k*ldb + n]
    int M, N, K;
    int reduction_major;          // 1 = both operands have the reduction index as the slow index (wgrad)
    float alpha;
    const float* bias;            // [N] or null
    const float* pe; int pe_period;   // add pe[(m % pe_period)*N + n]
    int relu;
    const bf16* mask; long ldmask; float mask_scale;   // v = mask>0 ? v*mask_scale : 0
    float drop_p; uint32_t seed, site;                 // dropout on element index m*N+n
    const uint32_t* seed_ptr;                          // non-null: the seed is read from device memory (graph-replayed steps)
    const float* residual; long ldres;                 // + residual[m*ldres + n]
    float* colsum;                // reduction-major only: colsum[m] = sum_k A(m,k) (fused bias gradient) or null
    int split_k;                  // reduction-major only: >1 splits the reduction over gridDim.z workgroups; split 0 writes
    long split_delta, split_stride;   // C32/colsum, split z>0 writes the same addresses + split_delta + (z-1)*split_stride (mk_split_reduce combines)
    int accumulate;               // C32 += v
    int xcd_order;                // set by mk_gemm: XCD-contiguous tile order
    int lean;                     // the GPU is shared with other task streams (masr_set_concurrency > 1): forms with the smaller LDS footprint
    float* C32; long ldc;         // fp32 output or null
    bf16* C16; long ldc16;        // bf16 output or null
    // segmented fp32 output rows (grouped weight gradients: one GEMM whose M axis spans several Linear layers that sit at a
    // constant distance from each other in the flat gradient buffer): row m lives at C32 + (m / cseg_rows) * cseg_stride +
    // (m % cseg_rows) * ldc, colsum[m] at colsum + (m / cseg_rows) * cseg_stride + m % cseg_rows.  cseg_rows = 0: plain rows;
    // otherwise a multiple of the 128-row tile.
    int cseg_rows; long cseg_stride;
};
int mk_gemm(const GemmArgs& g, hipStream_t s);
// Linear weight gradients as one grouped launch: dW[N][K] = dy[rows][N]^T x[rows][K], db[N] = column sums of dy (or null)
struct WgradDesc { const bf16* dy; const bf16* x; float* dW; float* db; int lddy, ldx, rows, N, K, tile_start; };
constexpr int WGRAD_GROUP_MAX = 40;
struct WgradGroup { int n; WgradDesc p[WGRAD_GROUP_MAX]; };
// ONE grid of 256 x 256 tiles on eight waves (LDS-DMA ring) over all members; fills tile_start.  first_members > 0: members [0, first_members)
// are dispatched first (the engine's merged launch: long encoder-row reductions, then the short decoder-row ones)
int mk_gemm_wgrad_grouped(WgradGroup& grp, hipStream_t s, int first_members = 0);
inline GemmArgs gemm_args() { GemmArgs g{}; g.alpha = 1.f; g.mask_scale = 1.f; return g; }

// ---------------------------------------------------------------- conv front-end (conv.hip)
// activations NHWC bf16: [B][H][W][C]; x input fp32 [B][H][W] (C=1)
// relu_bits (optional): [B][H][W] 64-bit words, bit c = (out[..][c] > 0) -- ConvArgs::mask_bits of the next conv's fused dgrad
int mk_conv1_fwd(const float* x, const float* w /*[64][9]*/, const float* bias, bf16* out, int B, int H, int W, hipStream_t s, unsigned long long* relu_bits = nullptr);
long mk_conv1_wgrad_slab_floats(int B, int H, int W);
// implicit-GEMM 3x3 pad 1: out[p][co] = epi( sum_{tap,ci} in[p+off(tap)][ci] * wk[co][tap*CIN+ci] )
struct ConvArgs {
    const bf16* in; const bf16* wk; const float* bias; int relu;
    // the input map given as a 2x2 max-pool (+ ReLU) backward that is never materialised: in_pooled [B][H/2][W/2][CIN] (the pooled gradient)
    // under the codes in_idx (ConvArgs::pool_idx of the forward launch that pooled).  `in` is then null.  Honoured by the two dgrad launches
    // that sit behind a pool: 64 <- 64 with the fused conv1 weight gradient (x1) and 128 <- 128 with mask_bits; their patch producers expand
    // the 2 x 2 windows while staging (a quarter of the gradient bytes + one code byte per pooled element instead of the map)
    const bf16* in_pooled; const uint8_t* in_idx;
    const bf16* mask;             // optional ReLU mask source (same shape as out): out = mask>0 ? v : 0
    // the same mask as one 64-bit word per pixel [B][H][W] (bit c = mask[..][c] > 0, written by mk_conv1_fwd): what the
    // fused-conv1-wgrad dgrad reads instead of the map (16 x 16 tiles)
    const unsigned long long* mask_bits;
    // the 128-channel layout of the same idea: FOUR dwords per pixel [B][H][W][4], dword q = the sign bytes of channel groups
    // 8q.., 32+8q.., 64+8q.., 96+8q.. (a lane of the streaming kernels masks exactly those).  out_sign_bits: a 128-output-channel
    // forward launch writes them for its ReLU'd output; a masked 128 <- 128 dgrad given them as mask_bits runs on 32-row tiles
    unsigned long long* out_sign_bits;
    bf16* out; int B, H, W, CIN, COUT;
    // 64->64 dgrad of the second conv only: fuse the weight gradient of conv1 (x1 = fp32 network input [B][H][W]) into the
    // epilogue; `out` is then never written, w1_slab receives 640 partial sums per workgroup (mk_conv1_wgrad_fused_reduce)
    const float* x1; float* w1_slab;
    unsigned* sched;              // streaming kernel: 2 zero-initialised counters owned by the calling stream (null: a process-wide pair)
    bf16* pool_out;               // optional: MaxPool2d(2,2) (floor) of the ReLU'd output, [B][H/2][W/2][COUT], written by the same launch
    // optional, with pool_out: one byte per pooled element = window position (0..3, row-major scan) of its FIRST maximum, 4 where
    // that maximum is <= 0 (nothing passes the ReLU): all the pool + ReLU backward needs of the full-resolution map
    uint8_t* pool_idx;
    int out_optional;             // the caller does not need `out`: a launch that pools in its epilogue may skip storing it
    // streaming kernels only: `out` (and `mask`, same shape) has out_cstride channels per pixel and this launch computes channels
    // out_coff .. out_coff + COUT of it (wk / bias already offset by the caller); 0 = a plain [..][COUT] map.  How mk_conv3x3 runs the
    // 256-channel convs of the BLSTM front-end: two passes of 128 output channels over the same patches.
    int out_cstride, out_coff;
};
int mk_conv3x3(const ConvArgs& a, hipStream_t s);
long mk_conv1_wgrad_fused_slab_floats(int B, int H, int W);
int mk_conv1_wgrad_fused_reduce(float* slab, int B, int H, int W, float* dw, float* db, hipStream_t s);
// wgrad: dw[co][ci][3][3] (+ db[co]) from in (NHWC, CIN) and dy (NHWC, COUT)
struct ConvWgradArgs {
    const bf16* in; const bf16* dy; float* dw; float* db; float* slab; int B, H, W, CIN, COUT;
    // optional: dy = the 2x2 max-pool (+ ReLU) backward of dy_pooled [B][H/2][W/2][COUT] under the codes of ConvArgs::pool_idx -- the kernel
    // expands it while staging (a quarter of the gradient bytes + one code byte per pooled element instead of the full-resolution map)
    const bf16* dy_pooled; const uint8_t* pool_idx;
};
int mk_conv3x3_wgrad(const ConvWgradArgs& a, hipStream_t s, int phase = 0);     // phase 1 / 2: the partial-slab kernel / the slab reduce alone
long mk_conv3x3_wgrad_slab_floats(int B, int H, int W, int CIN, int COUT);
int mk_maxpool_fwd(const bf16* in, bf16* out, int B, int H, int W, int C, hipStream_t s, int ceil_mode = 0);
// din = (in is the first max of its window && in > 0) ? dout : 0   (ReLU backward fused)
int mk_maxpool_relu_bwd(const bf16* in, const bf16* dout, bf16* din, int B, int H, int W, int C, hipStream_t s, int ceil_mode = 0);
// ConvArgs::pool_idx computed from a stored map (for the conv kernels that cannot emit the codes in their epilogue)
int mk_maxpool_idx(const bf16* in, uint8_t* idx, int B, int H, int W, int C, hipStream_t s);
// CIN = 1 convs with COUT = 64 n channels
int mk_conv1_fwd_n(const float* x, const float* w, const float* bias, bf16* out, int B, int H, int W, int COUT, hipStream_t s);
int mk_conv1_wgrad_n(const float* x, const bf16* dy, float* dw, float* db, float* slab, int B, int H, int W, int COUT, hipStream_t s);

// ---------------------------------------------------------------- attention (attention.hip)
struct AttnArgs {
    const bf16 *q, *k, *v; long ldq, ldk, ldv;   // row (b*T + t), head h at column h*hd
    bf16* o; long ldo; float* lse;               // lse [B][H][Tq]
    const bf16 *dout; long lddo;                 // backward
    bf16 *dq, *dk, *dv; long lddq, lddk, lddv;
    float* delta;                                // [B][H][Tq] scratch: rowsum(dO*O)
    const int* klens;                            // [B] valid keys per batch or null (all Tk)
    int B, H, Tq, Tk, hd, causal;
    float drop_p; uint32_t seed, site;
    const uint32_t* seed_ptr;                    // non-null: the seed is read from device memory (graph-replayed steps)
};
int mk_attn_fwd(const AttnArgs& a, hipStream_t s);
int mk_attn_bwd(const AttnArgs& a, hipStream_t s);

// ---------------------------------------------------------------- incremental greedy decode (decode.hip)
// C[m][n] = epi(sum_k A[m][k] W[n][k]) for a handful of rows (one per utterance): bias, ReLU, fp32 residual
struct SkinnyArgs {
    const bf16* A; long lda; const bf16* W; long ldw; int M, N, K;
    const float* bias; int relu; const float* residual; long ldres;
    float* C32; long ldc; bf16* C16; long ldc16;
};
int mk_skinny_gemm(const SkinnyArgs& g, hipStream_t s);
// one query row per (utterance, head) against cached keys/values; see decode.hip
struct AttnDecodeArgs {
    const bf16* q; long ldq;                   // [B][ldq], head h at column h*hd
    const bf16 *k, *v; long ldk, kv_batch_stride;   // cache rows: k + b*kv_batch_stride + j*ldk + h*hd
    const bf16 *knew, *vnew; long ldnew;       // self-attention: newest row (appended to the cache at slot *step-1) or null
    const int* step;                           // device scalar: number of valid keys (self-attention) or null
    const int* klens;                          // [B] valid keys (cross-attention) when step == null
    bf16* o; long ldo;
    int B, H, hd, Tk_cap;                      // Tk_cap bounds the key count (sizes the LDS score row)
};
int mk_attn_decode(const AttnDecodeArgs& a, hipStream_t s);
int mk_recog_embed_step(const int* step, const int* out, const float* table, const float* pe, float* y32, bf16* y16, int B, int E, int sos,
                        hipStream_t s);
// also advances step[0] once all B rows are done (ticket counter in step[1])
int mk_recog_argmax_step(int* step, const float* logits, long ld, int* out, int B, int C, hipStream_t s);
int mk_recog_step_set(int* step, int value, int inc, hipStream_t s);     // inc ? *step += 1 : *step = value
// logits[r][c] = bias[c] + <y32[r], W32[c]> in fp32 on the master weights (the decode's last projection: an arg-max follows)
int mk_logits_f32(const float* y32, const float* W32, const float* bias, float* logits, long ld, int rows, int C, int E, hipStream_t s);

// ---------------------------------------------------------------- row ops (rowops.hip)
int mk_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y32, bf16* y16,
                       float* mean, float* rstd, int rows, int E, hipStream_t s);
// dx32 = LN backward; dx16 = bf16 copy (optionally dropout-masked with (seed,site) for the branch GEMMs)
int mk_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                       float* dx32, bf16* dx16, float drop_p, uint32_t seed, uint32_t site,
                       float* dgamma, float* dbeta, float* slab, int rows, int E, hipStream_t s, const uint32_t* seed_ptr = nullptr);
// LayerNorm whose input row (forward) / incoming gradient (backward) is NOT in memory: it is what the epilogue of a k-split GEMM would have
// written -- the sum of n fp32 partial products part[z][row][E] (z-th at part + z * stride) (+ bias) (x dropout at element index row * E + column,
// the GEMM's) + residual.  Forward: the sum is stored to sum_out (the backward pass normalises it again).  Few rows only (< 2048).
struct LnSumArgs { const float* part; long stride; int n; const float* bias; const float* residual; float drop_p; uint32_t seed, site; const uint32_t* seed_ptr; float* sum_out; };
int mk_layernorm_fwd_sum(const LnSumArgs& sm, const float* gamma, const float* beta, float* y32, bf16* y16, float* mean, float* rstd, int rows, int E,
                         hipStream_t s);
int mk_layernorm_bwd_sum(const LnSumArgs& sm, const float* x, const float* gamma, const float* mean, const float* rstd, float* dx32, bf16* dx16,
                         float drop_p, uint32_t seed, uint32_t site, float* slab, int rows, int E, hipStream_t s, const uint32_t* seed_ptr = nullptr);
long mk_layernorm_bwd_slab_floats(int rows, int E);     // capacity for any row count <= rows
int mk_layernorm_bwd_blocks(int rows);                    // partial-sum blocks a launch over `rows` rows writes
// dgamma == null: only the per-block partials are written to `slab` (each LayerNorm its own region) and
// mk_layernorm_bwd_reduce_grouped folds all of them in one launch at the end of the backward pass
struct LnReduceDesc { const float* slab; float* dgamma; float* dbeta; int nblocks; };
constexpr int LN_GROUP_MAX = 64;
struct LnReduceGroup { int n; LnReduceDesc p[LN_GROUP_MAX]; };
int mk_layernorm_bwd_reduce_grouped(const LnReduceGroup& grp, int E, hipStream_t s);
// every fold pass that closes a transformer step's backward as ONE launch (fold.hip): conv[k] = slab reduce of a 3x3-conv weight gradient
// (mk_conv3x3_wgrad phase 2; nsplit = the partial slabs its phase 1 wrote), conv1 = the 640-sum rows of the fused conv1 weight gradient
// (nblocks rows: one per workgroup of that launch), ln = mk_layernorm_bwd_reduce_grouped's list, unperm = mk_vgg2enc_grad_unpermute,
// embed = mk_embed_bwd.  A null output pointer (dw / dtable) leaves the job out.
struct FoldJobs {
    int nconv, E;
    struct { const float* slab; int nsplit; float* dw; float* db; int CIN, COUT; } conv[3];
    struct { const float* slab; int nblocks; float* dw; float* db; } conv1;
    struct { const float* g; float* dw; int E, C, Dp; } unperm;
    struct { const int* order; const int* start; const float* dy; float* dtable; int V, E, accumulate; float drop_p; uint32_t seed, site; const uint32_t* seed_ptr; } embed;
    LnReduceGroup ln;
};
int mk_backward_folds(const FoldJobs& j, hipStream_t s);
int mk_conv3x3_wgrad_nsplit(const ConvWgradArgs& a);                     // the partial slabs mk_conv3x3_wgrad's phase 1 writes for this launch
int mk_conv1_wgrad_fused_rows(int B, int H, int W);                      // the 640-sum rows the fused conv1 weight gradient writes
int mk_embed_fwd(const int* tok, const float* table, const float* pe, float* y32, bf16* y16,
                   int B, int L, int E, float drop_p, uint32_t seed, uint32_t site, hipStream_t s, const uint32_t* seed_ptr = nullptr);
// dtable[v] (+)= sum over rows with tok==v of dy[row]  (deterministic: one block per vocab row)
// order / start: the token positions sorted by token id (ascending position inside a token) and the V + 1 segment starts, from the host
int mk_embed_bwd(const int* order, const int* start, const float* dy, float* dtable, int V, int E, int accumulate,
                 float drop_p, uint32_t seed, uint32_t site, hipStream_t s, const uint32_t* seed_ptr = nullptr);
// greedy decode: build the next decoder input from the previous step's tokens; arg-max of every logits row
int mk_recog_build_tok(int* tok, const int* out, int B, int L, int sos, hipStream_t s);
int mk_recog_argmax(const float* logits, long ld, int* out, int B, int L, int C, hipStream_t s);
// out[i] = the keep-scale (0 or 1/(1-p)) every dropout site applies to element i of its tensor (parity tests)
int mk_dropout_mask(float* out, long n, float p, uint32_t seed, uint32_t site, hipStream_t s);
// y16 = bf16(x32 * dropout_mask)
int mk_cast_dropout(const float* x, bf16* y, long n, float drop_p, uint32_t seed, uint32_t site, hipStream_t s, const uint32_t* seed_ptr = nullptr);
// column sums of x[rows][cols] (cols % 8 == 0, ld >= cols); only the first out_cols are written
int mk_colsum(const bf16* x, long ld, float* out, float* slab, int rows, int cols, int out_cols, hipStream_t s);
long mk_colsum_slab_floats(int rows, int cols);
// label-smoothed CE: logits fp32 [rows][ld]; gold int [rows] (-1 ignored); writes dlogits bf16 [rows][ld]
// stats[0] = loss (already / n_total), stats[1] = n_correct, stats[2] = n_total
int mk_ls_ce(const float* logits, long ld, const int* gold, int rows, int C, float eps, float inv_ntotal,
               bf16* dlogits, float* row_loss, int* row_correct, float* stats, hipStream_t s, const float* inv_ntotal_ptr = nullptr);

// ---------------------------------------------------------------- flat optimiser ops (optim.hip)
int mk_sumsq(const float* x, long n, float* slab, float* out_norm, hipStream_t s);       // out_norm[0] = sqrt(sum x^2)
long mk_sumsq_slab_floats(long n);
// coef = clip(max_norm / (norm+1e-6), <=1) (NaN propagates like torch.clamp)
// norm == nullptr: plain SGD step without clipping
int mk_clip_sgd(float* p, const float* g, float* mom, long n, const float* norm, float max_norm, float lr,
                  float momentum, int nesterov, int first_step, hipStream_t s);     // NaN norm -> step skipped
// (nan_zero: a NaN norm zeroes g / leaves acc alone instead of spreading NaNs -- masr_set_drop_nan_grads)
int mk_clip_scale(float* g, long n, const float* norm, float max_norm, hipStream_t s, int nan_zero = 0);    // g *= coef
int mk_clip_axpy(float* acc, const float* g, long n, const float* norm, float max_norm, hipStream_t s, int nan_zero = 0); // acc += coef*g
int mk_scale(float* x, long n, float a, hipStream_t s);
// weight_decay: decoupled != 0 -> AdamW (p *= 1 - lr*wd first), else torch.optim.Adam's L2 term (g += wd*p); 0 = plain Adam
int mk_adam_guarded(float* p, const float* g, float* m, float* v, long n, float lr_a, int t_a, float lr_b, int t_b, float b1, float b2, float eps,
                    float weight_decay, int decoupled, const float* norm, int* flags, int slot, hipStream_t s);
int mk_adam_sum(float* p, const float* const* grads, int n_grads, float gscale, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                int t, hipStream_t s);
int mk_adam(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, int t, float weight_decay,
            int decoupled, hipStream_t s);
int mk_sum_n(float* out, const float* const* grads, int n_grads, float scale, long n, hipStream_t s);     // out = (g0 + g1 + ...) * scale, n_grads <= 8
int mk_radam(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, int t, float weight_decay, int variant, hipStream_t s);
int mk_axpy(float* y, const float* x, long n, float a, hipStream_t s);
int mk_cast_bf16(const float* x, bf16* y, long n, hipStream_t s);
int mk_transpose_cast_bf16(const float* x /*[R][C]*/, bf16* y /*[C][ldy]*/, int R, int C, long ldy, hipStream_t s);
// conv weight shadows: w [CO][CI][3][3] fp32 -> wk [CO][tap*CI+ci] (fwd) and wd [CI][tap'*CO+co] = w[co][ci][8-tap'] (dgrad)
int mk_conv_weight_shadows(const float* w, bf16* wk, bf16* wd, int CO, int CI, hipStream_t s);       // (BLSTM engine; the transformer uses mk_all_shadows)
// EVERY bf16 operand shadow of the model in ONE launch (masr_refresh after each parameter update), descriptor driven:
//   SH_LINEAR   weight [N][K] at P+src -> k16 [N][K] and its transpose t16 [K][ldt] (64x64 tiles); ptrs[2i] = k16, ptrs[2i+1] = t16
//   SH_CONV     w [CO=N][CI=K][3][3] -> wk [CO][tap*CI+ci] (forward) and wd [CI][tap'*CO+co] = w[co][ci][8-tap'] (dgrad)
//   SH_VGG2ENC  w [E=N][a0*a1] with reference feature index c*Dp+d (a0 = C, a1 = Dp) -> wk [E][d*C+c] (NHWC order) and wt = wk^T
//   SH_COPY32   N floats at P+src -> (float*)ptrs[2i]   (the cross-attention K/V biases gathered into one vector)
enum { SH_LINEAR = 0, SH_CONV = 1, SH_VGG2ENC = 2, SH_COPY32 = 3 };
struct ShadowDesc { long src; int type, N, K, ldt, tile_start, a0, a1; };
// the whole job list travels in the kernel-argument segment (uniform scalar loads; a descriptor table in global memory cost
// every workgroup a chain of dependent loads to find its job)
constexpr int SHADOW_JOBS_MAX = 56;
struct ShadowJobs { int n, blocks; ShadowDesc d[SHADOW_JOBS_MAX]; bf16* p[2 * SHADOW_JOBS_MAX]; };
int mk_shadow_blocks(const ShadowDesc& d);                          // workgroups this job needs (tile_start bookkeeping on the host)
int mk_all_shadows(const float* P, const ShadowJobs& jobs, hipStream_t s);
// inverse map for the weight gradient: g_nhwc [E][d*C+c] fp32 -> dw [E][c*Dp+d]
int mk_vgg2enc_grad_unpermute(const float* g_nhwc, float* dw, int E, int C, int Dp, hipStream_t s);

// ---------------------------------------------------------------- data (data.hip)
// ragged gather + zero pad: rows of feat [sum T_i][D] -> xs [B][Tmax][D]
int mk_gather_pad(const float* feat, const long* row_start, const int* lens, float* xs, int B, int Tmax, int D, hipStream_t s);

// ---------------------------------------------------------------- bidirectional LSTM (lstm.hip) -- BLSTM-P encoder of the CTC config
// rows are batch-first (b*T + t); the gate axis is unit-major (u*4 + g, g in torch order i,f,g,o); index [2] = direction
struct LstmStepArgs {
    int B, T, H, KP;                  // KP = H rounded up to a multiple of 32 (zero-padded bf16 recurrent operands)
    const int* lens;                  // [B] valid frames per sequence (device)
    bf16* h16[2][2];                  // [B][KP] hidden state ping-pong
    const bf16* whh16[2];             // [4H][KP]  W_hh, rows unit-major
    const bf16* whhT16[2];            // [H][4H]   W_hh^T (columns unit-major)
    const float* gx[2];               // [B*T][4H] input contribution + biases
    float* act[2];                    // [B*T][4H] gate activations (saved for backward)
    float* c[2];                      // [B*T][H]  cell states
    float* cstate[2];                 // [B][H]    running cell state (forward) / dL/dc carry (backward)
    bf16* y16;                        // [B*T][2H] layer output (fwd | bwd halves), 0 at padded frames
    const float* dy;                  // [B*T][2H] gradient wrt y (backward)
    bf16* dz16[2];                    // [B*T][4H] gradient wrt the gate pre-activations (backward output)
};
int mk_lstm_fwd_steps(const LstmStepArgs& a, hipStream_t s);
int mk_lstm_bwd_steps(const LstmStepArgs& a, hipStream_t s);
// The same recurrences as ONE launch per layer and pass: workgroups that stay for the whole sequence, W_hh slices in registers, h_t / dz_t
// exchanged between them as self-flagging 8-byte granules (lstm_rec.hip).  `words`: mk_lstm_rec_words() 64-bit words of scratch (start of an
// allocation-aligned block); *err is set to 1 when a workgroup gave up waiting for its peers (never cleared by the kernels).
bool mk_lstm_rec_ok(int B, int H, int KP);
int64_t mk_lstm_rec_words(int B, int H);
int mk_lstm_fwd_rec(const LstmStepArgs& a, unsigned long long* words, int* err, hipStream_t s, int test_stall = 0);   // test_stall: fault injection of the time-out test -- workgroup 0 leaves without publishing
int mk_lstm_bwd_rec(const LstmStepArgs& a, unsigned long long* words, int* err, hipStream_t s);
// W_ih [4H][K], W_hh [4H][H], biases in torch order -> unit-major bf16 shadows (+ transposes for the dgrad GEMMs)
int mk_lstm_shadows(const float* wih, const float* whh, const float* bih, const float* bhh, int H, int K, int KP_in, int KP_h,
                    bf16* wih16, bf16* wihT16, bf16* whh16, bf16* whhT16, float* bias, int pc, int pd, hipStream_t s);
// (pc, pd) != 0: the input features are an NHWC conv map [pd][pc]; torch's weight columns are c*pd + d
int mk_lstm_unperm(const float* src, float* dst, float* dst2, int H, int K, int pc, int pd, hipStream_t s);   // unit-major rows -> torch rows
int mk_lstm_hprev(const bf16* y16, bf16* hp0, bf16* hp1, int B, int T, int H, int KP, hipStream_t s);
int mk_cast_rows_pad(const float* x, bf16* y, long rows, int C, int Cp, hipStream_t s);
int mk_tanh_fwd(const float* x, float* y32, bf16* y16, long n, hipStream_t s);
int mk_tanh_bwd(const float* dy, const float* y, bf16* dx16, long n, hipStream_t s);
int mk_mask_rows(float* x32, bf16* x16, const int* lens, int B, int T, int C, hipStream_t s);
// time sub-sampling between BLSTM-P layers (ys_pad[:, ::sub], src/modules/encoder.py:118-121): ys[b][t'] = y[b][t' * sub], C % 8 == 0;
// backward: dy[b][t] = t % sub == 0 ? dys[b][t / sub] : 0 (fp32, C % 4 == 0)
int mk_subsample_rows(const bf16* y, bf16* ys, int B, int Tin, int Tout, int sub, int C, hipStream_t s);
int mk_subsample_rows_bwd(const float* dys, float* dy, int B, int Tin, int Tout, int sub, int C, hipStream_t s);

// ---------------------------------------------------------------- features (fbank.hip)
// Kaldi-style log-mel filterbank: wav fp32 (PCM scale, utterances concatenated), wav_off [B+1], row_off [B] (first output
// row of each utterance), feat [sum T_b][n_mel]; T_b = 1 + (n_b - 400) / 160; grid covers max_frames frames per utterance
// with_pitch: rows have n_mel + 3 columns (the pitch dims are mk_pitch's) and an utterance has min(fbank, pitch) frames
int mk_fbank(const float* wav, const long* wav_off, const long* row_off, int B, int max_frames, int n_mel, float* feat, hipStream_t s, int with_pitch = 0);
// Kaldi pitch features (pitch.hip): columns n_mel .. n_mel+2 of the same rows
long mk_pitch_work_bytes(long total_samples, int B, int max_frames);
int mk_pitch(const float* wav, const long* wav_off, const long* row_off, long total_samples, long max_samples, int B, int max_frames, int n_mel, float* feat,
             void* work, long work_bytes, hipStream_t s);

// ---------------------------------------------------------------- CTC (ctc.hip)
// logits fp32 [T][B][C] (pre-softmax); targets concatenated int [sum tl]; per-sample offsets tgt_off [B]
// loss_out[0] = mean_b( nll_b / max(tl_b,1) ), zero_infinity; grad wrt logits [T][B][C]
int mk_ctc_loss(const float* logits, const int* targets, const int* tgt_off, const int* in_len, const int* tgt_len,
                  int T, int B, int C, int blank, float* nll /*[B]*/, float* loss_out, float* grad, float* work,
                  int maxS, hipStream_t s, int batch_first = 0);      // batch_first: logits / grad are [B][T][C]
int mk_ctc_status(const float* work, int T, int B, int maxS, hipStream_t s);     // > 0: (index + 1) of an utterance the CTC launch on `work` refused (bad lengths)
long mk_ctc_work_floats(int T, int B, int maxS);
