/* libmasr -- C ABI of the MI355X-native meta-ASR training path.
 *
 * Drop-in boundary for the hot path of sunprinceS/MetaASR-CrossAccent.
 * The reference has no native code and no FFI: its "operator boundary" is the Python mixin
 * contract  get_trainer(cls, config, paras, id2accent) -> solver with
 * solver.run_batch(idx, x, ilens, ys, olens, train) (src/transformer_torch_trainer.py:13-99)
 * driven by FOMetaASRInterface.run_task/_partial_meta_update/_final_meta_update
 * (src/fo_meta_interface.py:128-250).  Each entry point below names the reference code it
 * replaces.  The Python host mirror (metaasr-crossaccent_amd/) binds these with ctypes; see
 * INTEGRATION.md for the stub a maintainer of the reference would add.
 *
 * Conventions: plain pointers and sizes only (no torch types).  All device pointers are HIP
 * device memory owned by the caller.  Every call is ordered on the caller's hipStream_t
 * (passed as void*), allocates nothing and never synchronises unless stated.  Return 0 on
 * success, <0 on error (text via masr_last_error()).  A handle is not thread-safe.
 */
#ifndef MASR_H
#define MASR_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct masr_model masr_model;

/* asr_model block of the reference YAML (config/transformer/pretrain/fometa-hkust.yaml:13-27)
 * + odim (= len(id2units), src/pretrain_interface.py:38-43) + solver.label_smoothing. */
typedef struct masr_config {
    int32_t idim, odim, d_model, nheads, d_inner, enc_layers, dec_layers;
    int32_t tie_weights;          /* tgt_share_weight != 0 (mono_transformer_torch.py:66-70) */
    float dropout, pos_dropout, label_smoothing;
} masr_config;

#define MASR_TRAIN 1              /* run_batch(train=True): forward + loss + backward            */
#define MASR_EVAL 0               /* run_batch(train=False): forward + loss only (dropout off)   */

int masr_version(void);
const char* masr_last_error(void);

/* MyTransformer.__init__ (mono_transformer_torch.py:37-104): builds the parameter table only. */
masr_model* masr_create(const masr_config* cfg);
void masr_destroy(masr_model* m);

/* Flat fp32 parameter buffer layout = the reference state_dict order without
 * the pos_encoder.pe buffer and, when tied, without pre_embed.weight (alias of char_trans.weight). */
int64_t masr_param_numel(const masr_model* m);
int masr_param_count(const masr_model* m);
int masr_param_info(const masr_model* m, int idx, char* name, int name_cap, int64_t shape[4], int* ndim, int64_t* offset);

/* workspace needed for a batch of B utterances x T frames with L = max(olen)+1 target positions */
int64_t masr_workspace_bytes(const masr_model* m, int B, int T, int L);
/* params/grads: fp32 [masr_param_numel]; pe: fp32 [3000][d_model] (PositionalEncoding buffer, :16-28) */
int masr_bind(masr_model* m, float* params, float* grads, const float* pe, void* workspace, int64_t ws_bytes);
/* rebuild the bf16 operand shadows after ANY change of params (load_state_dict, optimizer step) */
int masr_refresh(masr_model* m, void* stream);
void masr_set_seed(masr_model* m, uint64_t seed);      /* dropout stream */
/* hint: this model is one of `slots` task slots running concurrently on the GPU (pretrain.py --tasks_per_gpu).  NO result depends on it,
 * bit for bit: no launch partition that enters a summation order follows the slot count (the k-split below follows masr_set_ksplit only;
 * tests/test_hip_engine.py::test_task_slot_hint_never_changes_bits).  What follows it is the LDS footprint of some launches: with slots > 1
 * the encoder-row GEMMs keep a three-stage operand ring (72 KB) instead of four (96 KB), so that the other slots' workgroups still fit
 * beside them on a CU.  Changing it drops captured step graphs (masr_set_step_graphs). */
void masr_set_concurrency(masr_model* m, int slots);
/* the dropout stream's position: state[0] = seed, state[1] = batches run since masr_set_seed (every run_batch derives its masks
 * from both); set != 0 writes it.  For checkpoints: a resumed run continues the mask stream where the saved one stopped. */
void masr_dropout_state(masr_model* m, uint64_t state[2], int set);

/* TransformerTrainer.run_batch (src/transformer_torch_trainer.py:59-99) = MyTransformer.forward
 * (:178-208) + label-smoothed CE (:64-84) + (train) zero_grad/backward.  xs: device fp32 [B][T][idim];
 * ilens/olens/ys_flat: HOST int64 (ys_flat = concatenated labels, sum(olens) entries).  Gradients
 * are left in `grads`.  olens is NOT mutated (quirk Q6 is reproduced by the Python mirror). */
int masr_run_batch(masr_model* m, const float* xs, const int64_t* ilens, const int64_t* ys_flat,
                   const int64_t* olens, int B, int T, int flags, void* stream);
/* Opt-in: a batch shape (B, T, L, flags, xs pointer) that repeats on a non-null stream is captured into
 * a hipGraph on its second consecutive occurrence and replayed afterwards (one launch instead of ~150; tokens, lengths, dropout
 * seed and 1/n_total reach the kernels through the per-step upload, so replays are bit-identical to direct launches).  It cuts
 * the host's enqueue time 6x and leaves the step time unchanged -- the step is GPU-bound -- hence off by default.
 * counters: out[0] = steps launched kernel by kernel, out[1] = graphs captured, out[2] = steps replayed from a graph, out[3] = k-split GEMM
 * launches of the last step launched or captured (0 = whole reductions).  Captured graphs hold the launch geometry of the settings they were
 * captured under: masr_set_concurrency / masr_set_ksplit / masr_set_split_wgrad_launches drop them (after a device synchronisation). */
void masr_set_step_graphs(masr_model* m, int on);
/* The Linear weight gradients of a step are ONE launch (the decoder-row tiles fill the CUs the encoder-row tiles leave idle); on: two
 * launches, encoder rows then decoder rows (A/B; identical bits -- each element of dW is reduced by one workgroup either way). */
void masr_set_split_wgrad_launches(masr_model* m, int on);
/* The decoder's few-row GEMMs with a long reduction (FFN second layer, its first layer's dgrad, the packed q/k/v dgrad: <= 1024 rows, K >= 1024;
 * and the attention out-projections in two halves) run k-split over K / 512 x as many workgroups; the LayerNorm (backward) that follows sums the
 * fp32 partial products and applies the GEMM's epilogue (bias, dropout, residual) on its way in.  It shortens a lone task's launch chain (+1..3 %)
 * and costs throughput beside other task slots, and it changes the fp32 summation order of those GEMMs -- so it follows THIS call only, never
 * masr_set_concurrency.  Default OFF (whole reductions).  The one-task-per-stream loops turn it on (train.py: the mono / multi trainers); the
 * FOMAML interface leaves it off for every --tasks_per_gpu, so that K slots == the sequential run == N ranks, bit for bit.  Both schedules are
 * pinned to the reference at the headline shape (tests/test_hip_fullsize.py).  masr_step_counters out[3] reports which one ran. */
void masr_set_ksplit(masr_model* m, int on);
void masr_step_counters(const masr_model* m, int64_t out[4]);
/* out[0]=loss, out[1]=n_correct, out[2]=n_total, out[3]=last grad norm.  Synchronises the stream. */
int masr_read_stats(masr_model* m, float out[4], void* stream);
/* the same four floats WITHOUT waiting: masr_stats_post queues their copy into a page-locked block owned by the handle (a ring of
   64) and records an event behind it; it returns a ticket >= 0.  masr_stats_wait(ticket) waits for that event (completion and
   host visibility of the copy) and hands the floats out; a ticket may simply be dropped -- its block is only reused after its
   event has completed.  masr_stats_peek returns the block itself, whose four words hold MASR_STATS_PENDING until the copy lands:
   a host thread may poll them without entering the HIP runtime (the task threads are inside its launch path meanwhile) and call
   masr_stats_wait once they have changed.  Tickets EXPIRE after 64 newer posts on the same handle (masr_stats_wait / _peek then fail
   with "unknown or expired ticket"): a caller that keeps more than that outstanding must read the oldest first (the Python engine
   does so at 48).  Lets the host queue the next tasks while these
   run: the reference reads loss / accuracy / norm only for its log lines (fo_meta_interface.py:147-151). */
#define MASR_STATS_PENDING 0x7FC0DEADu            /* a quiet NaN with a payload no kernel produces */
int64_t masr_stats_post(masr_model* m, void* stream);
const float* masr_stats_peek(masr_model* m, int64_t ticket);
int masr_stats_wait(masr_model* m, int64_t ticket, float out[4]);
/* device view of the last forward's logits: fp32 [rows = B*L][ld], first odim columns valid; and gold */
int masr_last_logits(masr_model* m, const float** logits, const int32_t** gold, int* rows, int* L, int* ld);

/* nn.utils.clip_grad_norm_(parameters, max_norm) (fo_meta_interface.py:148-149,242-243): norm only */
int masr_grad_norm(masr_model* m, void* stream);
/* ... followed by `if not isnan(norm): SGD(lr, momentum, nesterov).step()` (fo_meta_interface.py:228-248).
   first_step: bit 0 = the optimiser's first step (torch creates the momentum buffer as a copy of the gradient: the buffer is not
   read); bit 1 = its last step (run_task drops its SGD after k steps, :228-250: the buffer is not written).  With both bits set
   (meta_k = 1) momentum_buf is not touched and may be null. */
int masr_clip_sgd_step(masr_model* m, float* momentum_buf, float max_norm, float lr, float momentum, int nesterov,
                       int first_step, void* stream);
/* clip in place (multi_interface.py:108-109, mono fine-tune) */
int masr_clip_grads(masr_model* m, float max_norm, void* stream);
/* _partial_meta_update after the val-batch clip (fo_meta_interface.py:148-154,180-198): updates += clip(grads) */
int masr_clip_accumulate(masr_model* m, float* updates, float max_norm, void* stream);
/* Quirk Q5 of the reference (fo_meta_interface.py:151-154): a val-batch gradient whose norm is NaN is only warned about and still accumulated, so
   one bad batch turns the meta weights into NaNs.  Default (off) reproduces that.  On (pretrain.py --fix_nan_meta_grad): masr_clip_grads zeroes
   such a gradient and masr_clip_accumulate leaves `updates` alone -- decided on the device from the norm both already form, no host sync; the
   norm reported by masr_read_stats stays NaN, so the warning is still logged. */
void masr_set_drop_nan_grads(masr_model* m, int on);
/* buf[0..n) *= min(1, max_norm / (*norm + 1e-6)) with the norm read from device memory (what masr_allreduce applies chunk by chunk, as one
   pass: for transports that cannot pipeline it) */
int masr_clip_scale_flat(float* buf, int64_t n, const float* norm, float max_norm, void* stream);

/* flat helpers on arbitrary device buffers */
int masr_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                   float beta1, float beta2, float eps, int step, void* stream);
/* torch.optim.AdamW (decoupled != 0: p *= 1 - lr*weight_decay, then the Adam update) or torch.optim.Adam with its L2
 * weight_decay (decoupled == 0: g += weight_decay*p) -- config/transformer/adapt/hkust-adamw.yaml through
 * getattr(torch.optim, cls) (src/transformer_torch_trainer.py:44-46) */
/* Adam / AdamW step guarded ON THE DEVICE by the gradient norm of the model's stats block (masr_clip_grads leaves it there): a NaN
   norm skips the step, as `if math.isnan(grad_norm): warn else: step()` does (mono_interface.py:141-148, multi_interface.py:108-114),
   without the host having to read the norm first.  The host may be ONE step ahead: (lr_a, t_a) are learning rate and Adam step
   count if the previous guarded step was applied, (lr_b, t_b) if it was skipped -- the previous launch recorded which; pass the same
   pair twice when the previous outcome is known.  `slot` alternates 0 / 1 from step to step. */
int masr_adam_step_guarded(masr_model* m, float* p, const float* g, float* exp_avg, float* exp_avg_sq, int64_t n, float lr_a, int t_a,
                           float lr_b, int t_b, float b1, float b2, float eps, float weight_decay, int decoupled, int slot, void* stream);
/* the meta update of one meta-step in ONE pass: Adam on g = (((g_0 + g_1) + ...) + g_{n-1}) * gscale, the per-task gradients read
   straight from n <= 8 device buffers (`grads` is a HOST array of device pointers).  Replaces the accumulator of
   fo_meta_interface.py:180-202 (zero + n axpy passes + scale pass + Adam pass) with the same additions in the same order. */
int masr_adam_sum_step(float* p, const float* const* grads, int n_grads, float gscale, float* exp_avg, float* exp_avg_sq, int64_t n,
                       float lr, float b1, float b2, float eps, int step, void* stream);
/* out = (((g_0 + g_1) + ...) + g_{n-1}) * scale over n <= 8 device buffers (`grads` is a HOST array of device pointers): the
   rank-local sum of the task gradients of one wave of concurrent tasks = the payload of that wave's ONE RCCL all-reduce
   (the `_updates[n] += p.grad` of fo_meta_interface.py:190-196 for the tasks this rank ran) */
int masr_sum_n(float* out, const float* const* grads, int n_grads, float scale, int64_t n, void* stream);
int masr_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                    float beta1, float beta2, float eps, float weight_decay, int decoupled, int step, void* stream);
/* optimizer_cls 'RAdam' of set_model (src/transformer_torch_trainer.py:36-41).  The reference takes it from `torch_optimizer`, an
 * un-vendored third-party package absent from its tree.  variant 1 = that package's conventions (its authors' published
 * implementation: rectification once N_sma >= 5, denom = sqrt(v) + eps with sqrt(1 - b2^t) folded into the step size, weight decay
 * applied to the weight: p -= lr * wd * p) -- what FlatRAdam uses; pinned against a restatement of that algorithm in the oracle,
 * "parity unpinned" against the package itself.  variant 0 = torch.optim.RAdam's (rho_t > 5, bias-corrected denominator, L2 decay),
 * pinned against torch.optim.RAdam on the CPU (tests/test_hip_misc.py). */
int masr_radam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float b1, float b2,
                    float eps, float weight_decay, int step, int variant, void* stream);
/* torch.optim.SGD(momentum, nesterov) on arbitrary flat buffers */
int masr_sgd_step(float* params, const float* grads, float* momentum_buf, int64_t n, float lr, float momentum,
                  int nesterov, int first_step, void* stream);
int masr_scale(float* x, int64_t n, float a, void* stream);                           /* _updates /= counter (:201-202) */
int masr_axpy(float* y, const float* x, int64_t n, float a, void* stream);
int masr_copy(float* dst, const float* src, int64_t n, void* stream);                 /* load_state_dict(_original) (:226) */

/* ---------------------------------------------------------------------------------------------------------------------
 * The exchange step of sharded meta-training: all-reduce(sum) of the flat meta-gradient over the ranks' GPUs with
 * RCCL over xGMI.  The reference is one process: its `_updates[n] += p.grad` over the tasks of a meta-step and `_updates[n] /=
 * counter` (src/fo_meta_interface.py:190-196,200-202) become, with the tasks sharded one per GPU, local gradient -> all-reduce(sum)
 * -> the same division and the same (replicated) Noam-Adam step on every rank.  librccl is bound with dlopen at init: libmasr has
 * no link-time dependency on it and single-GPU runs never load it.
 *   masr_allreduce_unique_id : rank 0 draws the communicator id (MASR_UNIQUE_ID_BYTES bytes); the caller ships it to the other
 *                              ranks over any side channel (torch.distributed's store, a file, MPI ...).
 *   masr_allreduce_init      : ncclCommInitRank on the CURRENT HIP device (one process per GPU); the communicator owns a
 *                              non-blocking side stream.  NULL + masr_last_error() on failure.
 *   masr_allreduce           : buf[0..n) (device fp32) is summed over all ranks IN PLACE, on the communicator's side stream,
 *                              ordered behind everything queued so far on `producer_stream` -- the caller goes on queueing the
 *                              next task's inner forward on `producer_stream` meanwhile and must not touch buf before
 *                              masr_allreduce_wait.  norm != NULL: buf is first scaled by clip_grad_norm_'s coefficient
 *                              min(1, max_norm / (*norm + 1e-6)) (`norm` = device float, e.g. masr_stats_device(m) + 3 after
 *                              masr_grad_norm; the reference clips every task's gradient before accumulating it, :148-149).  The
 *                              coefficient exists only once the whole backward is done, so the scale pass cannot be bucketed into
 *                              the backward: it is PIPELINED with the collective instead, chunk by chunk (chunk k+1 is scaled on
 *                              the producer stream while chunk k is on the wire).  nchunks (1..16): a few LARGE chunks -- xGMI is
 *                              point-to-point, a ring is bound by one ~153 GB/s link, small buckets only add launch latency.
 *   masr_allreduce_wait      : `stream` waits (on the device, no host sync) for every exchange issued so far.
 * Not thread-safe per communicator; calls on one communicator must come in the same order on every rank (RCCL's rule). */
#define MASR_UNIQUE_ID_BYTES 128
typedef struct masr_comm masr_comm;
int masr_allreduce_unique_id(char* id);
masr_comm* masr_allreduce_init(int rank, int world, const char* id);
void masr_allreduce_destroy(masr_comm* c);
int masr_allreduce(masr_comm* c, float* buf, int64_t n, const float* norm, float max_norm, int nchunks, void* producer_stream);
int masr_allreduce_wait(masr_comm* c, void* stream);
/* host-side health check of the exchange: 0 = the last exchange completed (or none is pending), 1 = still running (timeout_ms == 0: one
 * poll), -1 = RCCL reports an asynchronous error, or timeout_ms ran out -- the communicator is then aborted and the job must end. */
int masr_allreduce_check(masr_comm* c, int timeout_ms);
/* device view of the model's stats block: [0] loss, [1] n_correct, [2] n_total, [3] gradient norm (after masr_grad_norm / masr_clip_*) */
const float* masr_stats_device(masr_model* m);

/* MyTransformer.recog (mono_transformer_torch.py:143-176): greedy decode, out int32 [Ldec][B] (device),
 * Ldec = max(floor(ilens/4)).  Needs workspace for L = Ldec.
 * masr_recog      : KV-cached incremental decode (one new position per step, the step replayed as a hipGraph);
 * masr_recog_full : the reference's literal schedule (the whole prefix is decoded again at every step).
 * Both emit the same token sequences (the target mask is causal). */
int masr_recog(masr_model* m, const float* xs, const int64_t* ilens, int B, int T, int32_t* out, void* stream);
int masr_recog_full(masr_model* m, const float* xs, const int64_t* ilens, int B, int T, int32_t* out, void* stream);

/* Levenshtein distance of two id sequences (host-side; replaces the `editdistance` extension the reference's metric
 * imports, src/monitor/metric.py:4,66,87).  Returns the distance, < 0 on bad arguments. */
int64_t masr_edit_distance(const int32_t* a, int na, const int32_t* b, int nb);

/* ---------------------------------------------------------------------------------------------------------------------
 * BLSTM-CTC model of config/blstm: MonoBLSTM.forward (src/model/blstm/mono_blstm.py:77-92) =
 * BlstmEncoder (src/modules/encoder.py:215-298: VGG 1->128->128 pool(ceil) ->256->256 pool(ceil), RNNP = nlayers x
 * {packed bidirectional LSTM(enc_dim), Linear(2 enc_dim -> proj_dim | enc_odim), tanh}, pad frames zeroed) + Linear head,
 * with BLSTMTrainer.run_batch's loss (src/blstm_trainer.py:55-85): targets [sos] + y + [eos] (sos = eos = odim - 1),
 * log_softmax + CTCLoss(blank 0, mean, zero_infinity).  sample_rate 1 / dropout 0 per layer (the shipped settings).
 * Same conventions as the masr_* calls above: flat fp32 params / grads in the reference's state_dict order (each tensor
 * starts on a 4-float boundary), caller-owned workspace, stream-ordered, int return codes. */
typedef struct masr_blstm masr_blstm;
typedef struct masr_blstm_config {
    int32_t idim, odim;          /* feature width (83), vocabulary incl. <blank> and <eos> (367) */
    int32_t enc_dim, proj_dim;   /* LSTM hidden size per direction, projection width between layers */
    int32_t enc_odim;            /* projection width of the last layer (encoder.odim) */
    int32_t nlayers;             /* <= 8 */
    int32_t sample_rate[8];      /* time sub-sampling behind BLSTM layer i (encoder.sample_rate, e.g. 1_2_2): the layer's output keeps every
                                  * sample_rate[i]-th frame, enc_lens -> (enc_lens + 1) / sample_rate[i] (src/modules/encoder.py:118-121); 0 = 1 */
} masr_blstm_config;
masr_blstm* masr_blstm_create(const masr_blstm_config* cfg);
void masr_blstm_destroy(masr_blstm* m);
int64_t masr_blstm_param_numel(const masr_blstm* m);
int masr_blstm_param_count(const masr_blstm* m);
int masr_blstm_param_info(const masr_blstm* m, int idx, char* name, int name_cap, int64_t shape[4], int* ndim, int64_t* offset);
int64_t masr_blstm_workspace_bytes(const masr_blstm* m, int B, int T, int max_target_len);
int masr_blstm_bind(masr_blstm* m, float* params, float* grads, void* workspace, int64_t ws_bytes);
int masr_blstm_refresh(masr_blstm* m, void* stream);
/* xs: device fp32 [B][T][idim]; ilens / olens: host int64 [B]; ys_flat: host int64 (labels without sos/eos, concatenated).
 * MASR_TRAIN leaves d loss / d params in the bound gradient buffer. */
int masr_blstm_run_batch(masr_blstm* m, const float* xs, const int64_t* ilens, const int64_t* ys_flat, const int64_t* olens,
                         int B, int T, int flags, void* stream);
/* forward only (MonoBLSTM.forward / greedy_decode, mono_blstm.py:63-92): head output readable through masr_blstm_last_logits */
int masr_blstm_forward(masr_blstm* m, const float* xs, const int64_t* ilens, int B, int T, void* stream);
int masr_blstm_read_stats(masr_blstm* m, float out[4], void* stream);            /* out[0] = CTC loss, out[3] = grad norm */
/* The LSTM recurrence of a layer as ONE launch per pass (workgroups resident for the whole sequence, W_hh slices in registers, h_t / dz_t
 * exchanged as self-flagging granules: csrc/lstm_rec.hip) instead of one launch per timestep.  Default ON for the shapes it covers (B <= 32,
 * enc_dim <= 384); off = per-timestep launches (A/B + test; same results to fp32 rounding).  A timed-out exchange is reported by
 * masr_blstm_read_stats. */
void masr_blstm_set_resident_recurrence(masr_blstm* m, int on);
/* forward-only callers (masr_blstm_forward + masr_blstm_last_logits, i.e. the Tester's greedy CTC decode) never read the stats block:
 * this is their check.  Synchronises the stream; -1 (text in masr_last_error(), mark cleared) when the resident recurrence of a launch
 * since the last check timed out -- the logits are then invalid.  After a time-out the remaining resident launches of the step return
 * at once (one bounded wait per step, not one per layer and pass). */
int masr_blstm_check(masr_blstm* m, void* stream);
/* head output (pre-softmax) [B][Tp][odim] fp32 and enc_lens int32 [B] on the device, Tp = ceil(ceil(T/2)/2) */
int masr_blstm_last_logits(masr_blstm* m, float** logits, int32_t** enc_lens, int* B, int* Tp, int* C);
/* nn.utils.clip_grad_norm_(parameters, max_norm) on the flat gradient; the norm is read with masr_blstm_read_stats */
int masr_blstm_clip_grads(masr_blstm* m, float max_norm, void* stream);
/* clip_grad_norm_(max_norm) + SGD(momentum, nesterov) step + shadow refresh (mono_interface.py:141-148) */
int masr_blstm_clip_sgd_step(masr_blstm* m, float* momentum_buf, float max_norm, float lr, float momentum, int nesterov,
                             int first_step, void* stream);

/* Log-mel filterbank features on the GPU, written in the layout of the reference's feat.dat shards
 * (src/io/dataset.py:123-139: one [sum T_b][idim] float matrix per split).  The reference has no extraction code; the
 * algorithm is Kaldi's compute-fbank-feats with the recipe's options (16 kHz, 25 ms / 10 ms, povey window, 512-point FFT,
 * mel bins over 20 Hz - 8 kHz, log) -- oracle/fbank_np.py.  wav: device fp32 on the 16-bit PCM scale, utterances
 * concatenated; wav_off: device int64 [B+1]; row_off: device int64 [B] first output row of each utterance;
 * frames of utterance b: T_b = 1 + (n_b - 400) / 160 (0 if n_b < 400); max_frames >= max T_b; feat: device [sum T_b][n_mel]. */
int masr_fbank(const float* wav, const int64_t* wav_off, const int64_t* row_off, int B, int max_frames, int n_mel,
               float* feat, void* stream);

/* The shipped 83-dim rows: n_mel log-mel bins | 3 Kaldi pitch dims (config/transformer/pretrain/fometa-hkust.yaml:13 `idim: 83`,
 * README.md:24).  The pitch dims follow ESPnet's make_fbank_pitch.sh = compute-kaldi-pitch-feats | process-kaldi-pitch-feats
 * with Kaldi's default options (4 kHz resampling, NCCF at 417 log-spaced lags between 1/400 s and 1/50 s, Viterbi, then
 * [2 * pov feature, 2 * POV-normalised log pitch, 10 * delta log pitch]; oracle/pitch_np.py; the dithering noise Kaldi adds to the
 * delta is omitted).  feat: device [sum T_b][n_mel + 3] with T_b = min(fbank frames, pitch frames of utterance b) -- the pitch
 * tracker needs (ceil(n_b / 4) - 182) / 40 + 1 frames' worth of samples -- as `paste-feats --length-tolerance=2` truncates them;
 * row_off from those T_b; max_frames >= max T_b; total_samples = wav_off[B], max_samples = max n_b (host values);
 * work: masr_fbank_pitch_work_bytes(total_samples, B, max_frames) bytes of device memory. */
int64_t masr_fbank_pitch_work_bytes(int64_t total_samples, int B, int max_frames);
int masr_fbank_pitch(const float* wav, const int64_t* wav_off, const int64_t* row_off, int64_t total_samples, int64_t max_samples, int B, int max_frames,
                     int n_mel, float* feat, void* work, int64_t work_bytes, void* stream);

/* collate_fn zero-padding of CommonVoiceDataset rows (src/io/dataset.py:21-33,147-153) done on the GPU:
 * feat fp32 [sum T_i][D] resident in HBM, row_start int64 [B] (device), lens int32 [B] (device). */
int masr_gather_pad(const float* feat, const int64_t* row_start, const int32_t* lens, float* xs, int B, int Tmax, int D, void* stream);

/* nn.CTCLoss(blank, reduction='mean', zero_infinity=True) on log_softmax(logits) with its gradient wrt the
 * logits (src/blstm_trainer.py:22,62-70).  logits fp32 [T][B][C] device; targets/tgt_off/in_len/tgt_len int32 device.
 * nll [B], loss [1], grad [T][B][C] device outputs; work: masr_ctc_work_floats(T,B,maxS) floats. */
int64_t masr_ctc_work_floats(int T, int B, int maxS);
/* The lengths are DEVICE arrays, so they are vetted by the kernel: an utterance with in_len < 0 or > T, tgt_len < 0 or
 * 2 tgt_len + 1 > maxS is not run -- its nll (hence the mean loss) is NaN, its gradient rows are zero.  in_len == 0 is torch's
 * "no path" case: nll 0, zero gradient (zero_infinity).  masr_ctc_status synchronises and returns 0, or (index + 1) of the last
 * utterance the most recent masr_ctc_loss call ON THIS work buffer (same T, B, maxS) refused, with the text in masr_last_error().  The mark
 * lives in the call's own work buffer (re-armed by every masr_ctc_loss on it): calls on different buffers / streams do not mix. */
int masr_ctc_status(const float* work, int T, int B, int maxS, void* stream);
int masr_ctc_loss(const float* logits, const int32_t* targets, const int32_t* tgt_off, const int32_t* in_len,
                  const int32_t* tgt_len, int T, int B, int C, int blank, float* nll, float* loss, float* grad,
                  float* work, int maxS, void* stream);

/* device timing (HIP events on the launch stream) for bench.py's roofline block: one slot per conv launch of the VGG
 * front-end (each is ONE launch per step, so slot time / launches = that kernel's average duration) and one per kernel
 * class for the rest.  bench.py::PROF_NAMES mirrors this list. */
#define MASR_PROF_CONV1_FWD 0     /* 1 -> 64, direct fp32 (HBM-bound) */
#define MASR_PROF_CONV2_FWD 1     /* 64 -> 64 on the full-resolution map + fused 2x2 max-pool */
#define MASR_PROF_CONV3_FWD 2     /* 64 -> 128 */
#define MASR_PROF_CONV4_FWD 3     /* 128 -> 128 + fused 2x2 max-pool */
#define MASR_PROF_CONV2_DGRAD 4   /* 64 <- 64 with conv1's weight gradient fused into the epilogue */
#define MASR_PROF_CONV3_DGRAD 5
#define MASR_PROF_CONV4_DGRAD 6
#define MASR_PROF_CONV2_WGRAD 7
#define MASR_PROF_CONV3_WGRAD 8
#define MASR_PROF_CONV4_WGRAD 9
#define MASR_PROF_CONV1_WGRAD 10  /* folds of the per-workgroup weight-gradient partials: conv1's fused sums and the slab reduces of conv2..4 */
#define MASR_PROF_GEMM_ENC 11     /* Linear forward / dgrad over the B*T' encoder rows (incl. vgg2enc, grouped cross-attention K/V) */
#define MASR_PROF_GEMM_DEC 12     /* ... over the B*L decoder rows */
#define MASR_PROF_WGRAD_ENC 13    /* Linear weight gradients reducing over encoder rows (split-K) */
#define MASR_PROF_WGRAD_DEC 14    /* grouped launch of the decoder-row weight gradients */
#define MASR_PROF_ATTN_ENC 15
#define MASR_PROF_ATTN_DEC 16
#define MASR_PROF_LAYERNORM 17
#define MASR_PROF_POOL 18         /* max-pool/ReLU backward */
#define MASR_PROF_OPTIM 19        /* grad norm, clip + SGD */
#define MASR_PROF_SHADOWS 20      /* bf16 operand shadows after a parameter update */
#define MASR_PROF_MISC 21         /* loss, embedding, casts, split-K combine */
#define MASR_PROF_N 22
int masr_profile_enable(masr_model* m, int on);
/* sums since the last call: ms[MASR_PROF_N], launches[MASR_PROF_N]; synchronises */
int masr_profile_read(masr_model* m, float* ms, int* launches);

#ifdef __cplusplus
}
#endif
#endif
