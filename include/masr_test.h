/* libmasr -- test-only entry points (exported by libmasr.so, NOT part of the operator API of include/masr.h).
 *
 * Standalone launches of single kernels for the parity tests (tests/test_hip_kernels.py, tools/), bf16 passed as uint16_t bit patterns,
 * and one fault-injection hook.  Same conventions as masr.h: plain pointers and sizes, device memory owned by the caller, stream-ordered,
 * 0 on success / < 0 on error (text: the masr_last_error function of masr.h).
 */
#ifndef MASR_TEST_H
#define MASR_TEST_H
#include "masr.h"
#ifdef __cplusplus
extern "C" {
#endif

/* fault injection for the time-out test of the resident LSTM recurrence: while on, the resident forward launches of THIS handle lose one
 * workgroup at start, so that its peers run into the bound of their wait and the step is reported as failed by masr_blstm_read_stats /
 * masr_blstm_check instead of hanging (tests/test_hip_blstm.py) */
void masr_test_blstm_stall(masr_blstm* m, int on);

/* standalone kernel entry points used by the parity tests (bf16 passed as uint16_t bit patterns) */
int masr_test_gemm(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, int M, int N, int K, int reduction_major,
                   const float* bias, int relu, float* C32, int64_t ldc, void* stream);
/* dropout (torch nn.Dropout inside nn.Transformer*Layer / PositionalEncoding, mono_transformer_torch.py:30-32,74-98): the keep-scale
 * (0 or 1/(1-p)) of element i at a site; the GEMM epilogue and the attention-probability sites with their dropout switched on */
int masr_test_dropout_mask(uint32_t seed, uint32_t site, int64_t n, float p, float* out, void* stream);
int masr_test_gemm_dropout(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, int M, int N, int K, float drop_p,
                           uint32_t seed, uint32_t site, float* C32, int64_t ldc, void* stream);
int masr_test_attention_dropout(const uint16_t* q, const uint16_t* k, const uint16_t* v, uint16_t* o, float* lse, int B, int H,
                                int Tq, int Tk, int hd, float drop_p, uint32_t seed, uint32_t site, void* stream);
/* the NT GEMM with any combination of its fused epilogue stages (tools/bench_gemm_epi.py: what each stage costs per launch) */
int masr_test_gemm_epi(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, int M, int N, int K, const float* bias, int relu,
                       float drop_p, const float* residual, const uint16_t* mask, float* C32, uint16_t* C16, void* stream);
/* the operand-shadow pass of masr_refresh on ONE Linear weight: W fp32 [N][K] at P + src (P 16-byte aligned, src any dword offset >= 4 with at
 * least four floats of P behind the tensor -- in the flat parameter buffer the shadowed tensors are neither first nor last) -> k16 bf16 [N][K]
 * and its transpose t16 bf16 [K][ldt] (ldt >= N; the pads of a row stay untouched) */
int masr_test_linear_shadows(const float* P, int64_t src, int N, int K, int ldt, uint16_t* k16, uint16_t* t16, void* stream);
/* the first conv of the VGG front-end (CIN = 1, 64 output channels; mono_transformer_torch.py:49-50) as the engine launches it: x fp32 [B][H][W],
 * w fp32 [64][9], bias fp32 [64] -> out bf16 [B][H][W][64] = ReLU(conv + bias) (fp32 arithmetic on the exact-fp32 MFMA, rounded once) and, when
 * relu_bits is given, one 64-bit word per pixel whose bit c says whether channel c passed the ReLU (what the fused dgrad of the second conv reads) */
int masr_test_conv1_fwd(const float* x, const float* w, const float* bias, uint16_t* out, uint64_t* relu_bits, int B, int H, int W, void* stream);
int masr_test_conv3x3(const uint16_t* in, const uint16_t* wk, const float* bias, int relu, uint16_t* out,
                      int B, int H, int W, int CIN, int COUT, void* stream);
/* dgrad/fused-pool flavours of the same kernel: mask (optional, same shape as out) zeroes outputs where mask <= 0;
   pool_out (optional, [B][H/2][W/2][COUT]) receives MaxPool2d(2,2) of the ReLU'd output */
int masr_test_conv3x3_ex(const uint16_t* in, const uint16_t* wk, const float* bias, int relu, const uint16_t* mask, uint16_t* out,
                         uint16_t* pool_out, int B, int H, int W, int CIN, int COUT, void* stream);
/* 128-channel ReLU masks as sign bits (four dwords per pixel, dword q = the sign bytes of channel groups 8q.., 32+8q.., 64+8q.., 96+8q..):
   a forward launch with 128 output channels writes them for its output (out_sign_bits), a masked 128 <- 128 dgrad reads them
   (mask_bits, next to the bf16 mask it replaces on the streaming path) */
int masr_test_conv3x3_sign_bits(const uint16_t* in, const uint16_t* wk, const float* bias, int relu, const uint16_t* mask, const uint32_t* mask_bits,
                                uint16_t* out, uint32_t* out_sign_bits, int B, int H, int W, int CIN, int COUT, void* stream);
/* the pooling forward conv as the engine launches it: pool_idx ([B][H/2][W/2][COUT] bytes) receives, per pooled element, the
   window position 0..3 (row-major) of its first maximum, or 4 where nothing passed the ReLU; drop_out != 0 allows the launch to
   leave `out` unwritten (the streaming kernels then never store the full-resolution map; the others still do). */
int masr_test_conv3x3_pool_idx(const uint16_t* in, const uint16_t* wk, const float* bias, uint16_t* out, uint16_t* pool_out,
                               uint8_t* pool_idx, int drop_out, int B, int H, int W, int CIN, int COUT, void* stream);
/* The two dgrad launches that sit behind a max-pool (reference: the autograd of nn.MaxPool2d + nn.ReLU in front of nn.Conv2d,
 * mono_transformer_torch.py:51-52,57-58).  Input either as the map dy [B][H][W][C] or -- dy == NULL -- as the pooled gradient dy_pooled
 * [B][H/2][W/2][C] + the codes of masr_test_conv3x3_pool_idx, expanded while the patches are staged (no map in memory): same bits.
 * masr_test_conv3x3_dgrad_pooled: 128 <- 128 channels through the ReLU mask given as sign words (masr_test_conv3x3_sign_bits).
 * masr_test_conv1_wgrad_fused: 64 <- 64 channels whose output is contracted with the network input x1 [B][H][W] inside the launch:
 * dw1 [64][9], db1 [64] = the weight / bias gradient of the FIRST conv; mask_bits = one 64-bit word of sign bits per pixel. */
int masr_test_conv3x3_dgrad_pooled(const uint16_t* dy, const uint16_t* dy_pooled, const uint8_t* pool_idx, const uint16_t* wk, const uint32_t* mask_bits,
                                   uint16_t* out, int B, int H, int W, void* stream);
int64_t masr_test_conv1_wgrad_fused_slab_floats(int B, int H, int W);
int masr_test_conv1_wgrad_fused(const uint16_t* dy, const uint16_t* dy_pooled, const uint8_t* pool_idx, const uint16_t* wk, const uint64_t* mask_bits,
                                const float* x1, float* slab, int64_t slab_floats, float* dw1, float* db1, int B, int H, int W, void* stream);
int masr_test_conv3x3_wgrad(const uint16_t* in, const uint16_t* dy, float* dw, float* slab, int64_t slab_floats,
                            int B, int H, int W, int CIN, int COUT, void* stream);
int64_t masr_test_conv3x3_wgrad_slab_floats(int B, int H, int W, int CIN, int COUT);
/* Linear weight gradients as a grouped launch (mk_gemm_wgrad_grouped: one grid of 256 x 256 tiles): dW[N][K] = dy[rows][N]^T x[rows][K], db[N] =
 * column sums of dy (or null); a second member with the same operands when dW2 is given.
 * _n: `members` members over the same operands (member i writes dW + i * member_stride); first_members > 0: the two-segment tile list of the
 * engine's merged launch -- members [0, first_members) reduce over `rows` rows and are dispatched first, the others over the first rows_rest. */
int masr_test_wgrad_grouped(const uint16_t* dy, int64_t lddy, const uint16_t* x, int64_t ldx, float* dW, float* db, float* dW2, float* db2,
                            int rows, int N, int K, void* stream);
/* the k-split GEMM + summing LayerNorm pair of the decoder (engine.hip ffn_fwd / ln_fwd, ffn_bwd / ln_bwd): forward when x == NULL, backward otherwise */
int masr_test_ksplit_ln(const uint16_t* A, const uint16_t* B, int rows, int E, int K, int split, const float* bias, const float* residual, float drop_p,
                        uint32_t seed, uint32_t site, float* part, const float* gamma, const float* beta, float* sum_out, float* y32, uint16_t* y16,
                        float* mean, float* rstd, const float* x, float* dx32, uint16_t* dx16, float* slab, void* stream);
int masr_test_wgrad_grouped_n(const uint16_t* dy, int64_t lddy, const uint16_t* x, int64_t ldx, float* dW, int64_t member_stride, int members,
                              int first_members, int rows, int rows_rest, int N, int K, void* stream);
/* the same with dy given as the pooled gradient [B][H/2][W/2][COUT] + the pool codes of masr_test_conv3x3_pool_idx (the weight-gradient kernel
 * expands the 2x2 max-pool + ReLU backward while staging; 64->64 and 128->128 channels); db may be null */
int masr_test_conv3x3_wgrad_pooled(const uint16_t* in, const uint16_t* dy_pooled, const uint8_t* pool_idx, float* dw, float* db, float* slab,
                                   int64_t slab_floats, int B, int H, int W, int CIN, int COUT, void* stream);
/* LayerNorm forward + backward of one [rows][E] fp32 matrix (nn.LayerNorm inside nn.Transformer*Layer, mono_transformer_torch.py:74-98):
 * y, y16 (bf16), mean / rstd per row; dx, dgamma, dbeta from dy.  slab: masr_test_layernorm_slab_floats(rows, E) floats of scratch. */
int64_t masr_test_layernorm_slab_floats(int rows, int E);
int masr_test_layernorm(const float* x, const float* gamma, const float* beta, const float* dy, float* y, uint16_t* y16, float* mean,
                        float* rstd, float* dx, uint16_t* dx16, float* dgamma, float* dbeta, float* slab, int rows, int E, float drop_p,
                        uint32_t seed, uint32_t site, void* stream);   /* dx16 = bf16(dx * keep-scale of element row * E + col at `site`) */
/* forward + backward of one attention with dropout on the probabilities (keep-scale of element ((b H + h) Tq + i) Tk + j at `site`,
 * masr_test_dropout_mask): the backward regenerates the masks of the forward from (seed, site, index) */
int masr_test_attention_dropout_bwd(const uint16_t* q, const uint16_t* k, const uint16_t* v, const uint16_t* dout, uint16_t* o, uint16_t* dq, uint16_t* dk,
                                    uint16_t* dv, float* lse, const int32_t* klens, int B, int H, int Tq, int Tk, int hd, int causal, float drop_p,
                                    uint32_t seed, uint32_t site, void* stream);
int masr_test_attention(const uint16_t* q, const uint16_t* k, const uint16_t* v, const uint16_t* dout, uint16_t* o,
                        uint16_t* dq, uint16_t* dk, uint16_t* dv, float* lse, float* delta, const int32_t* klens,
                        int B, int H, int Tq, int Tk, int hd, int causal, void* stream);

#ifdef __cplusplus
}
#endif
#endif
