// libmasr engine: the VGG-Transformer encoder-decoder training step of the reference
// (MyTransformer.forward, src/model/transformer_pytorch/mono_transformer_torch.py:178-208, plus
// run_batch's loss/backward, src/transformer_torch_trainer.py:59-99) orchestrated as a fixed
// sequence of hand-written gfx950 kernels on one HIP stream, behind the C ABI of include/masr.h.
//
// Memory model (sized for 288 GB HBM3E): ONE flat fp32 parameter buffer and ONE flat gradient buffer
// (caller-owned; every optimiser / clip / all-reduce is a single streaming pass), bf16 operand shadows
// of the weights (refreshed after each parameter update), and a bump-allocated activation arena that
// keeps every activation of the step resident (nothing is recomputed, nothing is re-read through host).
// Activations are batch-first row matrices [B*T][E]; conv activations are NHWC bf16.
#include <cmath>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>

#include "../../include/masr.h"
#include "../../include/masr_test.h"
#include "kernels.h"

static thread_local std::string g_err;
void mk_set_error(const char* what, const char* detail) { g_err = std::string(what) + ": " + detail; }

namespace {

struct PInfo { std::string name; int64_t shape[4]; int ndim; int64_t off; int64_t numel; };
struct Lin { int64_t w, b; int N, K; bf16 *k16, *t16; };          // weight [N][K]; k16 = bf16 copy, t16 = bf16 [K][Npad]
struct Norm { int64_t w, b; };
struct Attn { Lin in, out; bf16 *q_k16, *q_t16; };             // q_*: cross-attention only -- the query third of in_proj on its own
struct EncL { Attn sa; Lin l1, l2; Norm n1, n2; };
struct DecL { Attn sa, ca; Lin l1, l2; Norm n1, n2, n3; };
struct Conv { int64_t w, b; int CO, CI; bf16 *k16, *d16; };

struct Arena {
    char* base; int64_t cap, off;
    template <class T> T* get(int64_t n) {
        const int64_t bytes = (n * (int64_t)sizeof(T) + 255) & ~(int64_t)255;
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += bytes;
        return p;
    }
};

struct EncAct { float *s1, *x1_32, *s2, *m1, *r1, *m2, *r2, *lse; bf16 *qkv, *ao, *x1_16, *f; uint32_t site[4]; };
struct DecAct {
    float *s1, *y1_32, *s2, *y2_32, *s3, *m1, *r1, *m2, *r2, *m3, *r3, *lse_s, *lse_c;
    bf16 *qkv, *ao, *y1_16, *q, *kv, *co, *y2_16, *f; uint32_t site[6];      // kv: this layer's 2E columns of Acts::kv_all (row stride NK)
};
// per-decoder-layer bf16 gradient operands of the deferred (grouped) weight-gradient launch
struct DecGrad { bf16 *g3, *g2, *g1, *gf, *gq, *gqkv; };
struct EncGrad { bf16 *g2, *g1, *gf, *gqkv; };          // per-layer gradient operands of the encoder-row weight gradients (kept for the grouped launch)
struct Acts {
    int B, T, D, H2, W2, Tp, Dp, L, rows_e, rows_d;
    int *tok_in, *gold, *enc_lens, *step_dev;
    int *tok_order, *tok_start;             // decoder-input token positions sorted by token id + the C + 1 segment starts (embedding backward)
    uint32_t* meta;                                        // [8] behind enc_lens, same upload: [0] dropout seed of the step, [1] 1/n_total (float bits)
    bf16* step_qkv;                                        // incremental decode: the newest position's q|k|v [B][3E]
    bf16 *a1, *p1, *a3, *p2;
    unsigned long long *a1_bits, *a3_bits;                           // ReLU mask of a1, one word per pixel (written by conv1's forward, read by conv2's fused dgrad)
    uint8_t *i1, *i2;                                  // ConvArgs::pool_idx of the two pools (a2 / a4 are only written by conv kernels that cannot emit them)
    std::vector<float*> x32; std::vector<bf16*> x16;        // encoder layer inputs/outputs [NE+1]
    std::vector<EncAct> enc;
    float *mf, *rf; bf16* mem16; bf16* kv_all;             // kv_all [rows_e][ND*2E]: K|V of every decoder layer's cross-attention
    std::vector<float*> y32; std::vector<bf16*> y16;        // decoder layer inputs/outputs [ND+1]
    std::vector<DecAct> dec;
    float *mdf, *rdf; bf16* yf16;
    float* logits; bf16* dlogits; float* row_loss; int* row_correct;
    uint32_t site_v2e, site_emb;
    // backward scratch
    float *ge_a, *ge_b, *gd_a, *gd_b, *dmem32, *v2e_g32;
    bf16 *ge16, *gao_e, *gao_d, *gkv_all, *dp2, *da3, *dp1;      // (d(a4), d(a2) exist only as pooled gradient + codes; d(a1) never)
    float *delta_e, *delta_d;
    std::vector<DecGrad> dgr;
    std::vector<EncGrad> egr;
    float* part;                                           // fp32 partial products of a k-split few-row GEMM, summed by the LayerNorm that follows ([<= 8][rows_d][E])
    float* slab; int64_t slab_floats;
    float *cw_slab[3], *c1_slab;                           // partial slabs of the conv weight gradients: each its own, all folded by ONE launch at the end of the pass
    float* ln_slab; int64_t ln_slab_floats;                // one region per LayerNorm backward (grouped reduce)
};

}  // namespace

constexpr int KSPLIT_MAX = 8;
struct masr_model {
    masr_config cfg;
    int E, H, hd, Fi, NE, ND, C, Cp, D, Dp, F;
    std::vector<PInfo> params; int64_t nparams = 0;
    Conv conv[4]; Lin v2e, ct; int64_t embed_w; std::vector<EncL> enc; Norm enc_norm; std::vector<DecL> dec; Norm dec_norm;
    float *P = nullptr, *G = nullptr; const float* pe = nullptr;
    char* ws = nullptr; int64_t ws_bytes = 0, persist_bytes = 0;
    bf16 *v2e_k = nullptr;                    // permuted vgg2enc weight (NHWC feature order)
    // cross-attention K/V projections of ALL decoder layers as one operand: the encoder memory is projected once by one GEMM
    // with N = ND*2E (forward), its gradient comes back through one GEMM with K = ND*2E and the ND weight gradients are one
    // reduction-major GEMM with M = ND*2E (segmented output rows).  kv_k16 [ND*2E][E], kvT [E][ND*2E], kv_bias [ND*2E].
    bf16 *kv_k16 = nullptr, *kvT = nullptr; float* kv_bias = nullptr; int NK = 0;
    std::vector<ShadowJobs> shadows;                       // job list(s) of the operand-shadow refresh: one launch per <= SHADOW_JOBS_MAX jobs (hkust: one)
    float* stats = nullptr;                   // device [8]: loss, n_correct, n_total, grad_norm
    unsigned* conv_sched = nullptr;           // tile counters of the streaming conv kernel (this model's stream only)
    float* h_stats = nullptr;                 // pinned
    // ring of page-locked stats blocks owned by the handle, one event each (masr_stats_post / masr_stats_wait): a block is only
    // handed out again after its previous copy's event has completed, whatever became of the ticket
    static constexpr int RING = 64;
    float* h_ring = nullptr; hipEvent_t ring_ev[RING]; bool ring_used[RING]; int64_t ring_next = 0;
    int* h_stage = nullptr; int64_t stage_ints = 0; int stage_slot = 0; hipEvent_t stage_ev[4];
    uint64_t seed = 0x1234; uint64_t step = 0;
    Acts acts; bool have_acts = false;
    LnReduceGroup lng; int64_t ln_slab_used = 0;           // LayerNorm dgamma/dbeta partials, folded by one grouped launch
    bool split_wgrad = false;                              // masr_set_split_wgrad_launches
    int slots = 1;                                         // masr_set_concurrency: task slots sharing the GPU
    int drop_nan_grads = 0;                                // masr_set_drop_nan_grads: masr_clip_grads / masr_clip_accumulate turn a NaN-norm gradient into zeros (opt-out of quirk Q5)
    bool ksplit = false;                                   // masr_set_ksplit: few-row long-reduction GEMMs k-split, partials summed by the LayerNorm behind them
    int64_t n_ksplit = 0;                                  // k-split GEMM launches of the last masr_run_batch (masr_step_counters out[3])
    WgradGroup wg, wge;                                    // decoder-row / encoder-row weight gradients collected for the grouped launch (lin_wgrad)
    // captured training / evaluation steps (masr_run_batch, opt-in): a batch shape that repeats is replayed as ONE graph launch
    // instead of ~150 kernel launches.  Measured: host enqueue 0.61 -> 0.11 ms per step, step time unchanged (the GPU, not the
    // launch path, bounds both the single-task and the 4-task mode: tools/host_launch_cost.py) -- hence off by default
    struct StepGraph { int B, T, L, train; const void *ws, *P, *xs; hipGraph_t g; hipGraphExec_t e; uint64_t used; };
    std::vector<StepGraph> step_graphs; int last_key[4] = {0, 0, 0, -1}; const void* last_xs = nullptr; uint64_t graph_clock = 0;
    int64_t n_direct = 0, n_captured = 0, n_replayed = 0;  // masr_step_counters
    bool step_graphs_on = false;                           // masr_set_step_graphs
    // cached hipGraph of one incremental decode step (masr_recog)
    hipGraphExec_t dec_exec = nullptr; hipGraph_t dec_graph = nullptr; hipEvent_t dec_done = nullptr;
    int dec_key[3] = {0, 0, 0}; const void* dec_key_ptr[3] = {nullptr, nullptr, nullptr};
    // profiling
    bool prof = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_ev[MASR_PROF_N]; int prof_used[MASR_PROF_N] = {0};
};

namespace {

int64_t add_param(masr_model* m, const std::string& name, std::initializer_list<int64_t> shape) {
    PInfo p; p.name = name; p.ndim = (int)shape.size(); p.numel = 1;
    int i = 0; for (auto s : shape) { p.shape[i++] = s; p.numel *= s; }
    for (; i < 4; ++i) p.shape[i] = 1;
    p.off = m->nparams; m->nparams += p.numel;
    m->params.push_back(p);
    return p.off;
}
Lin add_linear(masr_model* m, const std::string& pre, int N, int K, const char* wname = ".weight", const char* bname = ".bias") {
    Lin l{}; l.N = N; l.K = K;
    l.w = add_param(m, pre + wname, {N, K});
    l.b = add_param(m, pre + bname, {N});
    return l;
}
Norm add_norm(masr_model* m, const std::string& pre) {
    Norm n; n.w = add_param(m, pre + ".weight", {m->E}); n.b = add_param(m, pre + ".bias", {m->E}); return n;
}
Attn add_attn(masr_model* m, const std::string& pre) {
    Attn a;
    a.in = add_linear(m, pre, 3 * m->E, m->E, ".in_proj_weight", ".in_proj_bias");
    a.out = add_linear(m, pre + ".out_proj", m->E, m->E);
    return a;
}

// ------------------------------------------------------------------ persistent region (shadows, stats)
void plan_persistent(masr_model* m, Arena& ar) {
    for (int i = 1; i < 4; ++i) {
        m->conv[i].k16 = ar.get<bf16>((int64_t)m->conv[i].CO * 9 * m->conv[i].CI);
        m->conv[i].d16 = ar.get<bf16>((int64_t)m->conv[i].CO * 9 * m->conv[i].CI);
    }
    auto lin = [&](Lin& l) {
        const int Np = (l.N + 7) / 8 * 8;
        l.k16 = ar.get<bf16>((int64_t)Np * l.K);
        l.t16 = ar.get<bf16>((int64_t)l.K * Np);
    };
    m->v2e_k = ar.get<bf16>((int64_t)m->E * m->F);
    m->v2e.t16 = ar.get<bf16>((int64_t)m->F * m->E);
    m->ct.k16 = ar.get<bf16>((int64_t)m->Cp * m->E);
    m->ct.t16 = ar.get<bf16>((int64_t)m->E * m->Cp);
    for (auto& e : m->enc) { lin(e.sa.in); lin(e.sa.out); lin(e.l1); lin(e.l2); }
    for (auto& d : m->dec) {
        lin(d.sa.in); lin(d.sa.out); lin(d.ca.out); lin(d.l1); lin(d.l2);
        d.ca.q_k16 = ar.get<bf16>((int64_t)m->E * m->E); d.ca.q_t16 = ar.get<bf16>((int64_t)m->E * m->E);
        d.ca.in.k16 = d.ca.in.t16 = nullptr;                        // (the packed in_proj of a cross-attention has no shadow of its own)
    }
    m->NK = m->ND * 2 * m->E;
    m->kv_k16 = ar.get<bf16>((int64_t)m->NK * m->E); m->kvT = ar.get<bf16>((int64_t)m->E * m->NK); m->kv_bias = ar.get<float>(m->NK);
    m->stats = ar.get<float>(64);
    m->conv_sched = ar.get<unsigned>(64);
}

// ------------------------------------------------------------------ activation plan
void plan_acts(const masr_model* m, Arena& ar, Acts& a, int B, int T, int L, bool train) {
    const int E = m->E, Fi = m->Fi, H = m->H;
    a.B = B; a.T = T; a.D = m->D; a.H2 = T / 2; a.W2 = m->D / 2; a.Tp = a.H2 / 2; a.Dp = a.W2 / 2; a.L = L;
    a.rows_e = B * a.Tp; a.rows_d = B * L;
    const int64_t re = a.rows_e, rd = a.rows_d;
    a.tok_in = ar.get<int>(3 * rd + B + 8 + m->C + 1); a.gold = a.tok_in + rd; a.enc_lens = a.gold + rd;   // one block: one H2D copy per step
    a.meta = reinterpret_cast<uint32_t*>(a.enc_lens + B);
    a.tok_order = a.enc_lens + B + 8; a.tok_start = a.tok_order + rd;
    a.step_dev = ar.get<int>(4);
    a.step_qkv = ar.get<bf16>((int64_t)B * 3 * E);
    const int64_t P1 = (int64_t)B * T * m->D, P2 = (int64_t)B * a.H2 * a.W2;
    a.a1 = ar.get<bf16>(P1 * 64); a.p1 = ar.get<bf16>(P2 * 64);      // (the maps in front of the pools, a2 and a4, are never stored: pooled map + codes)
    a.a1_bits = ar.get<unsigned long long>(P1); a.a3_bits = ar.get<unsigned long long>(P2 * 2);      // (64 / 128 sign bits per pixel)
    a.a3 = ar.get<bf16>(P2 * 128); a.p2 = ar.get<bf16>(re * m->F);
    a.i1 = ar.get<uint8_t>(P2 * 64); a.i2 = ar.get<uint8_t>(re * m->F);
    a.x32.resize(m->NE + 1); a.x16.resize(m->NE + 1); a.enc.resize(m->NE);
    for (int l = 0; l <= m->NE; ++l) { a.x32[l] = ar.get<float>(re * E); a.x16[l] = ar.get<bf16>(re * E); }
    for (auto& e : a.enc) {
        e.qkv = ar.get<bf16>(re * 3 * E); e.ao = ar.get<bf16>(re * E); e.lse = ar.get<float>((int64_t)B * H * a.Tp);
        e.s1 = ar.get<float>(re * E); e.x1_32 = ar.get<float>(re * E); e.x1_16 = ar.get<bf16>(re * E);
        e.m1 = ar.get<float>(re); e.r1 = ar.get<float>(re); e.f = ar.get<bf16>(re * Fi);
        e.s2 = ar.get<float>(re * E); e.m2 = ar.get<float>(re); e.r2 = ar.get<float>(re);
    }
    a.mf = ar.get<float>(re); a.rf = ar.get<float>(re); a.mem16 = ar.get<bf16>(re * E);
    a.kv_all = ar.get<bf16>(re * m->NK);
    a.y32.resize(m->ND + 1); a.y16.resize(m->ND + 1); a.dec.resize(m->ND);
    for (int l = 0; l <= m->ND; ++l) { a.y32[l] = ar.get<float>(rd * E); a.y16[l] = ar.get<bf16>(rd * E); }
    for (auto& d : a.dec) {
        d.qkv = ar.get<bf16>(rd * 3 * E); d.ao = ar.get<bf16>(rd * E); d.lse_s = ar.get<float>((int64_t)B * H * L);
        d.s1 = ar.get<float>(rd * E); d.y1_32 = ar.get<float>(rd * E); d.y1_16 = ar.get<bf16>(rd * E);
        d.m1 = ar.get<float>(rd); d.r1 = ar.get<float>(rd);
        d.q = ar.get<bf16>(rd * E); d.kv = nullptr; d.co = ar.get<bf16>(rd * E); d.lse_c = ar.get<float>((int64_t)B * H * L);
        d.s2 = ar.get<float>(rd * E); d.y2_32 = ar.get<float>(rd * E); d.y2_16 = ar.get<bf16>(rd * E);
        d.m2 = ar.get<float>(rd); d.r2 = ar.get<float>(rd);
        d.f = ar.get<bf16>(rd * Fi); d.s3 = ar.get<float>(rd * E); d.m3 = ar.get<float>(rd); d.r3 = ar.get<float>(rd);
    }
    if (a.kv_all) for (int l = 0; l < m->ND; ++l) a.dec[l].kv = a.kv_all + (int64_t)l * 2 * E;
    a.mdf = ar.get<float>(rd); a.rdf = ar.get<float>(rd); a.yf16 = ar.get<bf16>(rd * E);
    a.logits = ar.get<float>(rd * m->Cp); a.dlogits = ar.get<bf16>(rd * m->Cp);
    a.row_loss = ar.get<float>(rd); a.row_correct = ar.get<int>(rd);
    // slab: max over all users
    int64_t sl = mk_sumsq_slab_floats(m->nparams);
    auto mx = [&](int64_t v) { if (v > sl) sl = v; };
    mx(mk_layernorm_bwd_slab_floats((int)(re > rd ? re : rd), E));
    mx(mk_colsum_slab_floats((int)P1, 64)); mx(mk_colsum_slab_floats((int)P2, 128));
    mx(mk_colsum_slab_floats((int)(re > rd ? re : rd), 3 * E > Fi ? 3 * E : Fi));
    a.slab_floats = sl; a.slab = ar.get<float>(sl);
    a.part = ar.get<float>((int64_t)KSPLIT_MAX * rd * E);
    if (train) {
        const int nln = 2 * m->NE + 1 + 3 * m->ND + 1;
        a.ln_slab_floats = 0;
        for (int i = 0; i < nln; ++i) a.ln_slab_floats += mk_layernorm_bwd_slab_floats((int)(i < 2 * m->NE + 1 ? re : rd), E);
        a.ln_slab = ar.get<float>(a.ln_slab_floats);
        a.cw_slab[0] = ar.get<float>(mk_conv3x3_wgrad_slab_floats(B, T, m->D, 64, 64));
        a.cw_slab[1] = ar.get<float>(mk_conv3x3_wgrad_slab_floats(B, a.H2, a.W2, 64, 128));
        a.cw_slab[2] = ar.get<float>(mk_conv3x3_wgrad_slab_floats(B, a.H2, a.W2, 128, 128));
        a.c1_slab = ar.get<float>(mk_conv1_wgrad_fused_slab_floats(B, T, m->D));
        a.ge_a = ar.get<float>(re * E); a.ge_b = ar.get<float>(re * E); a.gd_a = ar.get<float>(rd * E); a.gd_b = ar.get<float>(rd * E);
        a.dmem32 = ar.get<float>(re * E); a.v2e_g32 = ar.get<float>((int64_t)E * m->F);
        a.ge16 = ar.get<bf16>(re * E);
        a.gao_e = ar.get<bf16>(re * E); a.gao_d = ar.get<bf16>(rd * E);
        a.gkv_all = ar.get<bf16>(re * m->NK);
        a.delta_e = ar.get<float>((int64_t)B * H * a.Tp); a.delta_d = ar.get<float>((int64_t)B * H * L);
        a.egr.resize(m->NE);
        for (auto& g : a.egr) { g.g2 = ar.get<bf16>(re * E); g.g1 = ar.get<bf16>(re * E); g.gf = ar.get<bf16>(re * Fi); g.gqkv = ar.get<bf16>(re * 3 * E); }
        a.dgr.resize(m->ND);
        for (auto& g : a.dgr) {
            g.g3 = ar.get<bf16>(rd * E); g.g2 = ar.get<bf16>(rd * E); g.g1 = ar.get<bf16>(rd * E);
            g.gf = ar.get<bf16>(rd * Fi); g.gq = ar.get<bf16>(rd * E); g.gqkv = ar.get<bf16>(rd * 3 * E);
        }
        a.dp2 = ar.get<bf16>(re * m->F); a.da3 = ar.get<bf16>(P2 * 128); a.dp1 = ar.get<bf16>(P2 * 64);
    }
}

// ------------------------------------------------------------------ profiling scope
struct Prof {
    masr_model* m; int cat; hipStream_t s; bool on;
    Prof(masr_model* m_, int cat_, hipStream_t s_) : m(m_), cat(cat_), s(s_), on(m_->prof) {
        if (!on) return;
        auto& v = m->prof_ev[cat];
        if (m->prof_used[cat] == (int)v.size()) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); v.push_back({a, b});
        }
        hipEventRecord(v[m->prof_used[cat]].first, s);
    }
    ~Prof() {
        if (!on) return;
        hipEventRecord(m->prof_ev[cat][m->prof_used[cat]].second, s);
        m->prof_used[cat]++;
    }
};

#define CK(expr) do { if ((expr) != 0) return -1; } while (0)

// Y = X W^T (+bias ...) with the bf16 shadow of W
GemmArgs lin_fwd_args(const bf16* x, long ldx, const bf16* wk, int M, int N, int K, const float* bias) {
    GemmArgs g = gemm_args();
    g.A = x; g.lda = ldx; g.B = wk; g.ldb = K; g.M = M; g.N = N; g.K = K; g.bias = bias;
    return g;
}

// seed_ptr / inv_ptr: non-null while a step is being captured into a graph -- the dropout seed and 1/n_total of the step
// then live in device memory (Acts::meta, uploaded with the tokens), so one captured launch sequence serves every step
struct Ctx { masr_model* m; hipStream_t s; uint32_t seed; bool train; float p_drop, p_pos; const uint32_t* seed_ptr = nullptr; const float* inv_ptr = nullptr; };

int gemm(Ctx& c, const GemmArgs& g) {
    const int re = c.m->acts.rows_e;
    const int cat = g.reduction_major ? (g.K == re ? MASR_PROF_WGRAD_ENC : MASR_PROF_WGRAD_DEC) : (g.M == re ? MASR_PROF_GEMM_ENC : MASR_PROF_GEMM_DEC);
    masr_model* m = c.m;
    Prof p(c.m, cat, c.s);
    GemmArgs h = g; h.seed_ptr = c.seed_ptr; h.lean = m->slots > 1;
    return mk_gemm(h, c.s);
}

// weight/bias gradients of a Linear: dW[N][K] = dy^T x, db = colsum(dy).  They feed nothing but the optimiser, so they are not launched one by
// one: every layer keeps its own dY operand and ONE grid of 256 x 256 tiles (gemm.hip gemm_wgrad_grouped16_kernel) computes them all at the end
// of the backward pass -- the encoder-row members (reduction over B*T' rows: `enc`) first, the decoder-row ones (B*L rows) filling the CUs
// those leave idle.  A member that does not fit the descriptor list (very deep models) runs as a plain reduction-major GEMM at once.
int lin_wgrad(Ctx& c, const bf16* dy, long lddy, const bf16* x, long ldx, int rows, int N, int K, float* dW, float* db, bool enc = false) {
    masr_model* m = c.m;
    WgradGroup& grp = enc ? m->wge : m->wg;
    if (m->wge.n + m->wg.n < WGRAD_GROUP_MAX) {
        WgradDesc& d = grp.p[grp.n++];
        d.dy = dy; d.x = x; d.dW = dW; d.db = db; d.lddy = (int)lddy; d.ldx = (int)ldx; d.rows = rows; d.N = N; d.K = K; d.tile_start = 0;
        return 0;
    }
    GemmArgs g = gemm_args();
    g.reduction_major = 1; g.A = dy; g.lda = lddy; g.B = x; g.ldb = ldx; g.M = N; g.N = K; g.K = rows;
    g.C32 = dW; g.ldc = K; g.colsum = db;
    return gemm(c, g);
}
// hkust: 148 tiles over 4000 rows + 228 over 592 rows.  As two launches the first leaves 108 CUs idle for ~110 us and the second takes ~40 us
// of its own; as one the short tiles run on those CUs (125 us).  masr_set_split_wgrad_launches: two launches (A/B; same bits -- every element
// of dW is reduced by one workgroup over its rows in order either way: tests/test_hip_engine.py).
int flush_wgrads(Ctx& c) {
    masr_model* m = c.m;
    int rc = 0;
    if (!m->split_wgrad) {
        const int first = m->wge.n;                                        // the encoder-row members go first (long reductions)
        for (int i = 0; i < m->wg.n; ++i) m->wge.p[m->wge.n++] = m->wg.p[i];
        m->wg.n = 0;
        if (m->wge.n) { Prof p(m, MASR_PROF_WGRAD_ENC, c.s); rc = mk_gemm_wgrad_grouped(m->wge, c.s, first); }
    } else {
        if (m->wge.n) { Prof p(m, MASR_PROF_WGRAD_ENC, c.s); rc = mk_gemm_wgrad_grouped(m->wge, c.s); }
        if (rc == 0 && m->wg.n) { Prof p(m, MASR_PROF_WGRAD_DEC, c.s); rc = mk_gemm_wgrad_grouped(m->wg, c.s); }
    }
    m->wge.n = 0; m->wg.n = 0;
    return rc;
}
// dX = dy W via the transposed shadow t16 [K][ldt]
GemmArgs lin_dgrad_args(const bf16* dy, long lddy, const bf16* t16, long ldt, int rows, int N, int K) {
    GemmArgs g = gemm_args();
    g.A = dy; g.lda = lddy; g.B = t16; g.ldb = ldt; g.M = rows; g.N = K; g.K = N;
    return g;
}

// The few-row GEMMs with a long reduction (decoder rows: FFN second layer and the first layer's dgrad, K = d_inner; packed q/k/v dgrad, K = 3E) are
// 80 workgroups with a chain of 24-32 k steps each -- 16 us where their K = 512 siblings take 8.  They run k-split over K / 512 x as many
// workgroups; each writes its fp32 partial product and the LayerNorm that always follows sums them (and applies what the GEMM's epilogue would
// have: bias, dropout, residual) on its way in: no combine pass, no extra launch (rowops.hip LnSumArgs).  0: not this shape.
// It buys latency with occupancy -- 320 workgroups x 7.4 us instead of 80 x 16 -- so it pays with the GPU to the task alone (train.py: +3 %) and costs
// beside other task slots, where occupancy counts (four-slot throughput 9 920 -> 10 000 utt/s with whole reductions).  It changes the fp32 summation
// order, so it follows ONLY masr_set_ksplit (default off), never the slot count: the caller that runs one task per GPU turns it on (mono / multi
// interface), the FOMAML interface leaves it off for every --tasks_per_gpu (K slots == the sequential run == N ranks, bit for bit).
static bool ksplit_on(const masr_model* m) { return m->ksplit; }
static int ksplit_of(const masr_model* m, int rows, int K) { return (ksplit_on(m) && rows <= 1024 && K >= 1024 && K % 512 == 0 && K / 512 <= KSPLIT_MAX) ? K / 512 : 0; }
static GemmArgs ksplit_args(masr_model* m, GemmArgs g, int S, int rows, int N) {
    ++m->n_ksplit;
    g.bias = nullptr; g.drop_p = 0.f; g.residual = nullptr; g.C16 = nullptr;
    g.C32 = m->acts.part; g.ldc = N; g.split_k = S; g.split_stride = (long)rows * N;
    return g;
}
int attn_block_fwd(Ctx& c, const Attn& at, const bf16* xq, const bf16* xkv, int rows_q, int rows_kv, int Tq, int Tk, bool self,
                   bool causal, const int* klens, bf16* qkv_or_q, bf16* kv, bf16* ao, float* lse, const float* resid, float* s_out,
                   uint32_t site_p, uint32_t site_o, LnSumArgs* defer = nullptr) {
    masr_model* m = c.m; const int E = m->E; const float* P = m->P;
    AttnArgs a{};
    if (self) {
        GemmArgs g = lin_fwd_args(xq, E, at.in.k16, rows_q, 3 * E, E, P + at.in.b); g.C16 = qkv_or_q; g.ldc16 = 3 * E;
        CK(gemm(c, g));
        a.q = qkv_or_q; a.k = qkv_or_q + E; a.v = qkv_or_q + 2 * E; a.ldq = a.ldk = a.ldv = 3 * E;
    } else {
        GemmArgs g = lin_fwd_args(xq, E, at.q_k16, rows_q, E, E, P + at.in.b); g.C16 = qkv_or_q; g.ldc16 = E;
        CK(gemm(c, g));
        // K|V of the encoder memory were projected for all layers at once (project_memory_kv); kv = this layer's columns
        (void)xkv; (void)rows_kv;
        a.q = qkv_or_q; a.ldq = E; a.k = kv; a.v = kv + E; a.ldk = a.ldv = m->NK;
    }
    a.o = ao; a.ldo = E; a.lse = lse; a.klens = klens; a.B = m->acts.B; a.H = m->H; a.Tq = Tq; a.Tk = Tk; a.hd = m->hd;
    a.causal = causal; a.drop_p = c.p_drop; a.seed = c.seed; a.seed_ptr = c.seed_ptr; a.site = site_p;
    { Prof p(m, Tk == m->acts.Tp && Tq == Tk ? MASR_PROF_ATTN_ENC : MASR_PROF_ATTN_DEC, c.s); CK(mk_attn_fwd(a, c.s)); }
    GemmArgs o = lin_fwd_args(ao, E, at.out.k16, rows_q, E, E, P + at.out.b);
    o.drop_p = c.p_drop; o.seed = c.seed; o.site = site_o; o.residual = resid; o.ldres = E; o.C32 = s_out; o.ldc = E;
    // few rows: the reduction over E runs as two halves on twice the workgroups, the LayerNorm behind the block sums them (see ksplit_of):
    // out-projection 8.6 -> 5.9 us, the LayerNorm 4.8 -> 5.6 with the second partial to read
    const int S = (defer && ksplit_on(m) && rows_q <= 1024 && E >= 512 && E % 128 == 0) ? 2 : 0;
    if (S) {
        *defer = LnSumArgs{m->acts.part, (long)rows_q * E, S, P + at.out.b, resid, c.p_drop, c.seed, site_o, c.seed_ptr, s_out};
        return gemm(c, ksplit_args(m, o, S, rows_q, E));
    }
    if (defer) defer->n = 0;
    CK(gemm(c, o));
    return 0;
}

int ffn_fwd(Ctx& c, const Lin& l1, const Lin& l2, const bf16* x16, const float* x32, int rows, bf16* f, float* s_out, uint32_t site_i, uint32_t site_o,
            LnSumArgs* defer = nullptr) {
    masr_model* m = c.m; const int E = m->E, Fi = m->Fi; const float* P = m->P;
    GemmArgs g = lin_fwd_args(x16, E, l1.k16, rows, Fi, E, P + l1.b); g.relu = 1; g.drop_p = c.p_drop; g.seed = c.seed; g.site = site_i;
    g.C16 = f; g.ldc16 = Fi;
    CK(gemm(c, g));
    GemmArgs h = lin_fwd_args(f, Fi, l2.k16, rows, E, Fi, P + l2.b); h.drop_p = c.p_drop; h.seed = c.seed; h.site = site_o;
    h.residual = x32; h.ldres = E; h.C32 = s_out; h.ldc = E;
    const int S = defer ? ksplit_of(m, rows, Fi) : 0;
    if (S) {                                                   // (the caller's LayerNorm takes `defer`: ln_fwd below)
        *defer = LnSumArgs{m->acts.part, (long)rows * E, S, P + l2.b, x32, c.p_drop, c.seed, site_o, c.seed_ptr, s_out};
        return gemm(c, ksplit_args(m, h, S, rows, E));
    }
    if (defer) defer->n = 0;
    CK(gemm(c, h));
    return 0;
}
int ln_fwd(Ctx& c, const Norm& n, const float* x, float* y32, bf16* y16, float* mean, float* rstd, int rows, const LnSumArgs* sum = nullptr) {
    masr_model* m = c.m;
    if (sum && sum->n > 0) {                                   // x = the partial products of a k-split GEMM (ffn_fwd): summed on the way in
        Prof p(c.m, MASR_PROF_LAYERNORM, c.s);
        return mk_layernorm_fwd_sum(*sum, m->P + n.w, m->P + n.b, y32, y16, mean, rstd, rows, m->E, c.s);
    }
    Prof p(c.m, MASR_PROF_LAYERNORM, c.s);
    return mk_layernorm_fwd(x, c.m->P + n.w, c.m->P + n.b, y32, y16, mean, rstd, rows, c.m->E, c.s);
}
int ln_bwd(Ctx& c, const Norm& n, const float* dy, const float* x, const float* mean, const float* rstd, float* dx32, bf16* dx16,
           uint32_t site, int rows, const LnSumArgs* sum = nullptr) {
    masr_model* m = c.m;
    // the dgamma/dbeta partials of every LayerNorm go to their own slab region; the fold launch at the end of the pass folds them all at once
    const int64_t need = (int64_t)mk_layernorm_bwd_blocks(rows) * 2 * m->E;
    if (sum && sum->n > 0) {                                   // dy = the partial products of a k-split dgrad GEMM + its residual gradient
        if (m->lng.n >= LN_GROUP_MAX || m->ln_slab_used + need > m->acts.ln_slab_floats) { mk_set_error("ln_bwd", "no room for the LayerNorm partials"); return -1; }
        float* slab = m->acts.ln_slab + m->ln_slab_used;
        m->ln_slab_used += need;
        LnReduceDesc& d = m->lng.p[m->lng.n++];
        d.slab = slab; d.dgamma = m->G + n.w; d.dbeta = m->G + n.b; d.nblocks = (rows + 3) / 4;
        Prof p(c.m, MASR_PROF_LAYERNORM, c.s);
        return mk_layernorm_bwd_sum(*sum, x, m->P + n.w, mean, rstd, dx32, dx16, dx16 ? c.p_drop : 0.f, c.seed, site, slab, rows, m->E, c.s, c.seed_ptr);
    }
    Prof p(c.m, MASR_PROF_LAYERNORM, c.s);
    if (m->lng.n < LN_GROUP_MAX && m->ln_slab_used + need <= m->acts.ln_slab_floats) {
        float* slab = m->acts.ln_slab + m->ln_slab_used;
        m->ln_slab_used += need;
        LnReduceDesc& d = m->lng.p[m->lng.n++];
        d.slab = slab; d.dgamma = m->G + n.w; d.dbeta = m->G + n.b; d.nblocks = (int)(need / (2 * m->E));
        return mk_layernorm_bwd(dy, x, m->P + n.w, mean, rstd, dx32, dx16, dx16 ? c.p_drop : 0.f, c.seed, site, nullptr, nullptr, slab, rows,
                                m->E, c.s, c.seed_ptr);
    }
    return mk_layernorm_bwd(dy, x, m->P + n.w, mean, rstd, dx32, dx16, dx16 ? c.p_drop : 0.f, c.seed, site, m->G + n.w, m->G + n.b,
                            m->acts.slab, rows, m->E, c.s, c.seed_ptr);
}
int flush_ln_reduce(Ctx& c) {
    masr_model* m = c.m;
    Prof p(m, MASR_PROF_LAYERNORM, c.s);
    const int rc = mk_layernorm_bwd_reduce_grouped(m->lng, m->E, c.s);
    m->lng.n = 0; m->ln_slab_used = 0;
    return rc;
}

// backward of  s_out = x + drop(ffn(x16))  given d s_out (gs32 fp32, gs16 bf16 already dropout-masked for the ffn output site)
// writes d x (fp32) = gs32 + ffn-branch gradient into gout
int ffn_bwd(Ctx& c, const Lin& l1, const Lin& l2, const bf16* x16, const bf16* f, const float* gs32, const bf16* gs16, int rows,
            bf16* gf, float* gout, bool split, LnSumArgs* defer = nullptr) {
    masr_model* m = c.m; const int E = m->E, Fi = m->Fi;
    CK(lin_wgrad(c, gs16, E, f, Fi, rows, E, Fi, m->G + l2.w, m->G + l2.b, split));
    GemmArgs g = lin_dgrad_args(gs16, E, l2.t16, E, rows, E, Fi);
    g.mask = f; g.ldmask = Fi; g.mask_scale = c.p_drop > 0.f ? 1.f / (1.f - c.p_drop) : 1.f; g.C16 = gf; g.ldc16 = Fi;
    CK(gemm(c, g));
    CK(lin_wgrad(c, gf, Fi, x16, E, rows, Fi, E, m->G + l1.w, m->G + l1.b, split));
    GemmArgs h = lin_dgrad_args(gf, Fi, l1.t16, Fi, rows, Fi, E);
    h.residual = gs32; h.ldres = E; h.C32 = gout; h.ldc = E;
    const int S = defer ? ksplit_of(m, rows, Fi) : 0;
    if (S) {                                                   // (gout is not written: the LayerNorm backward that follows takes `defer`)
        *defer = LnSumArgs{m->acts.part, (long)rows * E, S, nullptr, gs32, 0.f, 0u, 0u, nullptr, nullptr};
        return gemm(c, ksplit_args(m, h, S, rows, E));
    }
    if (defer) defer->n = 0;
    CK(gemm(c, h));
    return 0;
}


}  // namespace

// =========================================================================== C ABI
extern "C" {

int masr_version(void) { return 1; }
const char* masr_last_error(void) { return g_err.c_str(); }

masr_model* masr_create(const masr_config* cfg) {
    if (!cfg || cfg->d_model % cfg->nheads || cfg->d_model % 8 || cfg->d_inner % 8 || cfg->idim < 4) {
        mk_set_error("masr_create", "bad config"); return nullptr;
    }
    const int hd = cfg->d_model / cfg->nheads;
    if (hd != 16 && hd != 32 && hd != 64) { mk_set_error("masr_create", "head dim must be 16/32/64"); return nullptr; }
    if (cfg->d_model % 64) { mk_set_error("masr_create", "d_model must be a multiple of 64"); return nullptr; }
    if (cfg->d_model > 1024 || cfg->d_inner > 2048 || 3 * cfg->d_model > 2048) {
        mk_set_error("masr_create", "d_model <= 682, d_inner <= 2048 supported"); return nullptr;
    }
    masr_model* m = new masr_model();
    m->cfg = *cfg; m->E = cfg->d_model; m->H = cfg->nheads; m->hd = hd; m->Fi = cfg->d_inner; m->NE = cfg->enc_layers;
    m->ND = cfg->dec_layers; m->C = cfg->odim; m->Cp = (cfg->odim + 127) / 128 * 128; m->D = cfg->idim;
    m->Dp = (cfg->idim / 2) / 2; m->F = 128 * m->Dp;
    // state_dict order of the reference
    const int idx[4] = {0, 2, 5, 7}; const int co[4] = {64, 64, 128, 128}, ci[4] = {1, 64, 64, 128};
    for (int i = 0; i < 4; ++i) {
        Conv& c = m->conv[i]; c.CO = co[i]; c.CI = ci[i]; c.k16 = c.d16 = nullptr;
        const std::string pre = "feat_extractor." + std::to_string(idx[i]);
        c.w = add_param(m, pre + ".weight", {co[i], ci[i], 3, 3});
        c.b = add_param(m, pre + ".bias", {co[i]});
    }
    m->v2e = add_linear(m, "vgg2enc", m->E, 128 * (cfg->idim / 4));
    m->ct = add_linear(m, "char_trans", m->C, m->E);
    m->embed_w = cfg->tie_weights ? m->ct.w : add_param(m, "pre_embed.weight", {m->C, m->E});
    m->enc.resize(m->NE);
    for (int l = 0; l < m->NE; ++l) {
        const std::string pre = "encoder.layers." + std::to_string(l);
        EncL& e = m->enc[l];
        e.sa = add_attn(m, pre + ".self_attn");
        e.l1 = add_linear(m, pre + ".linear1", m->Fi, m->E); e.l2 = add_linear(m, pre + ".linear2", m->E, m->Fi);
        e.n1 = add_norm(m, pre + ".norm1"); e.n2 = add_norm(m, pre + ".norm2");
    }
    m->enc_norm = add_norm(m, "encoder.norm");
    m->dec.resize(m->ND);
    for (int l = 0; l < m->ND; ++l) {
        const std::string pre = "decoder.layers." + std::to_string(l);
        DecL& d = m->dec[l];
        d.sa = add_attn(m, pre + ".self_attn"); d.ca = add_attn(m, pre + ".multihead_attn");
        d.l1 = add_linear(m, pre + ".linear1", m->Fi, m->E); d.l2 = add_linear(m, pre + ".linear2", m->E, m->Fi);
        d.n1 = add_norm(m, pre + ".norm1"); d.n2 = add_norm(m, pre + ".norm2"); d.n3 = add_norm(m, pre + ".norm3");
    }
    m->dec_norm = add_norm(m, "decoder.norm");
    Arena ar{nullptr, 0, 0};
    plan_persistent(m, ar);
    m->persist_bytes = ar.off;
    m->stage_ints = 1 << 16;
    for (auto& e : m->stage_ev) e = nullptr;
    for (int i = 0; i < masr_model::RING; ++i) { m->ring_ev[i] = nullptr; m->ring_used[i] = false; }
    return m;
}

void masr_destroy(masr_model* m) {
    if (!m) return;
    if (m->h_stage) hipHostFree(m->h_stage);
    if (m->h_stats) hipHostFree(m->h_stats);
    for (int i = 0; i < masr_model::RING; ++i) if (m->ring_ev[i]) { if (m->ring_used[i]) hipEventSynchronize(m->ring_ev[i]); hipEventDestroy(m->ring_ev[i]); }
    if (m->h_ring) hipHostFree(m->h_ring);
    if (m->dec_done) { hipEventSynchronize(m->dec_done); hipEventDestroy(m->dec_done); }
    if (m->dec_exec) hipGraphExecDestroy(m->dec_exec);
    if (m->dec_graph) hipGraphDestroy(m->dec_graph);
    for (auto& sg : m->step_graphs) { hipGraphExecDestroy(sg.e); hipGraphDestroy(sg.g); }
    for (auto& e : m->stage_ev) if (e) hipEventDestroy(e);
    for (auto& v : m->prof_ev) for (auto& p : v) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    delete m;
}

int64_t masr_param_numel(const masr_model* m) { return m->nparams; }
int masr_param_count(const masr_model* m) { return (int)m->params.size(); }
int masr_param_info(const masr_model* m, int idx, char* name, int cap, int64_t shape[4], int* ndim, int64_t* offset) {
    if (idx < 0 || idx >= (int)m->params.size()) { mk_set_error("masr_param_info", "index out of range"); return -1; }
    const PInfo& p = m->params[idx];
    if (name && cap > 0) { std::strncpy(name, p.name.c_str(), cap - 1); name[cap - 1] = 0; }
    for (int i = 0; i < 4; ++i) shape[i] = p.shape[i];
    *ndim = p.ndim; *offset = p.off;
    return 0;
}

int64_t masr_workspace_bytes(const masr_model* m, int B, int T, int L) {
    Arena ar{nullptr, 0, 0};
    Acts a;
    plan_acts(m, ar, a, B, T, L, true);
    return m->persist_bytes + ar.off + 4096;
}

int masr_bind(masr_model* m, float* params, float* grads, const float* pe, void* workspace, int64_t ws_bytes) {
    if (!params || !grads || !pe || !workspace || ws_bytes < m->persist_bytes) { mk_set_error("masr_bind", "null pointer or workspace too small"); return -1; }
    if (((uintptr_t)workspace & 255) || ((uintptr_t)params & 15) || ((uintptr_t)grads & 15)) { mk_set_error("masr_bind", "misaligned buffers"); return -1; }
    m->P = params; m->G = grads; m->pe = pe; m->ws = (char*)workspace; m->ws_bytes = ws_bytes;
    Arena ar{m->ws, ws_bytes, 0};
    plan_persistent(m, ar);
    if (!m->h_stage) {
        HIP_CHECK_RET(hipHostMalloc((void**)&m->h_stage, sizeof(int) * m->stage_ints * 4, hipHostMallocDefault));
        HIP_CHECK_RET(hipHostMalloc((void**)&m->h_stats, sizeof(float) * 64, hipHostMallocDefault));
        for (auto& e : m->stage_ev) HIP_CHECK_RET(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIP_CHECK_RET(hipHostMalloc((void**)&m->h_ring, sizeof(float) * 4 * masr_model::RING, hipHostMallocDefault));
        for (auto& e : m->ring_ev) HIP_CHECK_RET(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    {   // device tables: the job list of the one-launch shadow refresh
        // the list travels BY VALUE in the kernel arguments (kernels.h), which caps one launch at SHADOW_JOBS_MAX jobs: deeper
        // models (8e4d = 69 jobs) simply take a second launch
        m->shadows.clear();
        auto job = [&](int type, long src, int N, int K, int ldt, int a0, int a1, void* p0, void* p1) {
            if (m->shadows.empty() || m->shadows.back().n == SHADOW_JOBS_MAX) { m->shadows.emplace_back(); m->shadows.back().n = 0; m->shadows.back().blocks = 0; }
            ShadowJobs& J = m->shadows.back();
            ShadowDesc d{}; d.src = src; d.type = type; d.N = N; d.K = K; d.ldt = ldt; d.a0 = a0; d.a1 = a1; d.tile_start = J.blocks;
            J.blocks += mk_shadow_blocks(d);
            J.d[J.n] = d; J.p[2 * J.n] = (bf16*)p0; J.p[2 * J.n + 1] = (bf16*)p1; ++J.n;
        };
        auto lin = [&](const Lin& l) { job(SH_LINEAR, l.w, l.N, l.K, (l.N + 7) / 8 * 8, 0, 0, l.k16, l.t16); };
        const int E = m->E;
        for (int i = 1; i < 4; ++i) job(SH_CONV, m->conv[i].w, m->conv[i].CO, m->conv[i].CI, 0, 0, 0, m->conv[i].k16, m->conv[i].d16);
        job(SH_VGG2ENC, m->v2e.w, E, 0, 0, 128, m->Dp, m->v2e_k, m->v2e.t16);
        job(SH_LINEAR, m->ct.w, m->C, E, m->Cp, 0, 0, m->ct.k16, m->ct.t16);     // pads (rows / columns >= odim) stay zero, see below
        for (auto& e : m->enc) { lin(e.sa.in); lin(e.sa.out); lin(e.l1); lin(e.l2); }
        for (int l = 0; l < m->ND; ++l) {
            const DecL& d = m->dec[l];
            lin(d.sa.in); lin(d.sa.out); lin(d.ca.out); lin(d.l1); lin(d.l2);
            job(SH_LINEAR, d.ca.in.w, E, E, E, 0, 0, d.ca.q_k16, d.ca.q_t16);                                              // query third
            job(SH_LINEAR, d.ca.in.w + (long)E * E, 2 * E, E, m->NK, 0, 0, m->kv_k16 + (long)l * 2 * E * E, m->kvT + (long)l * 2 * E);   // key|value thirds
            job(SH_COPY32, d.ca.in.b + E, 2 * E, 0, 0, 0, 0, m->kv_bias + (long)l * 2 * E, nullptr);
        }
    }
    // pads of the char_trans shadows must be zero (rows/cols >= odim); the refresh kernels only write the odim part
    HIP_CHECK_RET(hipMemset(m->conv_sched, 0, sizeof(unsigned) * 64));   // tile counters of the streaming conv (re-armed by the kernel itself)
    HIP_CHECK_RET(hipMemset(m->ct.k16, 0, sizeof(bf16) * (size_t)m->Cp * m->E));
    HIP_CHECK_RET(hipMemset(m->ct.t16, 0, sizeof(bf16) * (size_t)m->E * m->Cp));
    m->have_acts = false;
    return 0;
}

void masr_set_seed(masr_model* m, uint64_t seed) { m->seed = seed; m->step = 0; }
// captured steps hold the launch geometry of the settings they were captured under: every setter that changes it drops them (after the
// device has drained: nothing of a graph may be in flight when it is destroyed)
static void drop_step_graphs(masr_model* m) {
    if (m->step_graphs.empty()) return;
    hipDeviceSynchronize();
    for (auto& sg : m->step_graphs) { hipGraphExecDestroy(sg.e); hipGraphDestroy(sg.g); }
    m->step_graphs.clear();
    m->last_key[3] = -1; m->last_xs = nullptr;
}
// (no RESULT depends on the number of task slots -- "K task slots == the sequential run, bit for bit" is a guarantee of --tasks_per_gpu, and
// every partition into partial sums is fixed per shape and per masr_set_ksplit; what follows the hint is the LDS footprint of a few launches:
// GemmArgs::lean)
void masr_set_concurrency(masr_model* m, int slots) { slots = slots < 1 ? 1 : slots; if (slots != m->slots) drop_step_graphs(m); m->slots = slots; }
void masr_dropout_state(masr_model* m, uint64_t state[2], int set) {
    if (set) { m->seed = state[0]; m->step = state[1]; } else { state[0] = m->seed; state[1] = m->step; }
}

int masr_refresh(masr_model* m, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!m->P) { mk_set_error("masr_refresh", "not bound"); return -1; }
    Prof p(m, MASR_PROF_SHADOWS, s);
    // every bf16 operand shadow (conv forward/dgrad layouts, permuted vgg2enc, all Linear weights and their transposes, the
    // gathered cross-attention K/V operand) in ONE launch; the pads of the char_trans shadows are zeroed once in masr_bind
    for (const ShadowJobs& J : m->shadows) CK(mk_all_shadows(m->P, J, s));
    return 0;
}

static int forward_encoder(Ctx& c, const float* xs) {
    masr_model* m = c.m; Acts& a = m->acts; hipStream_t s = c.s; const float* P = m->P;
    const int B = a.B, T = a.T, D = a.D, E = m->E;
    uint32_t site = 1;
    {
        Prof p(m, MASR_PROF_CONV1_FWD, s);
        CK(mk_conv1_fwd(xs, P + m->conv[0].w, P + m->conv[0].b, a.a1, B, T, D, s, c.train ? a.a1_bits : nullptr));
    }
    // the maps in front of the two pools are needed by nothing but the pool + ReLU backward, and that needs one byte per POOLED
    // element (which window position won, or that none passed the ReLU): the pooling convs store those and drop the map
    auto conv = [&](const bf16* in, const Conv& cv, bf16* out, int H, int W, bf16* pooled, uint8_t* idx) -> int {
        Prof p(m, MASR_PROF_CONV2_FWD + (int)(&cv - &m->conv[1]), s);
        ConvArgs ca{}; ca.sched = m->conv_sched; ca.in = in; ca.wk = cv.k16; ca.bias = P + cv.b; ca.relu = 1; ca.mask = nullptr; ca.out = out;
        ca.B = B; ca.H = H; ca.W = W; ca.CIN = cv.CI; ca.COUT = cv.CO; ca.pool_out = pooled;      // MaxPool2d written by the producing conv's epilogue
        if (ca.pool_out) { ca.pool_idx = c.train ? idx : nullptr; ca.out_optional = 1; }
        if (&cv == &m->conv[2] && c.train) ca.out_sign_bits = a.a3_bits;      // conv3's ReLU mask as sign bits for conv4's masked dgrad
        return mk_conv3x3(ca, s);
    };
    CK(conv(a.a1, m->conv[1], nullptr, T, D, a.p1, a.i1));
    CK(conv(a.p1, m->conv[2], a.a3, a.H2, a.W2, nullptr, nullptr));
    CK(conv(a.a3, m->conv[3], nullptr, a.H2, a.W2, a.p2, a.i2));
    // vgg2enc + positional encoding + pos dropout
    {
        GemmArgs g = lin_fwd_args(a.p2, m->F, m->v2e_k, a.rows_e, E, m->F, P + m->v2e.b);
        g.pe = m->pe; g.pe_period = a.Tp; g.drop_p = c.p_pos; g.seed = c.seed; g.site = a.site_v2e = site++;
        g.C32 = a.x32[0]; g.ldc = E; g.C16 = a.x16[0]; g.ldc16 = E;
        CK(gemm(c, g));
    }
    for (int l = 0; l < m->NE; ++l) {
        EncAct& e = a.enc[l]; const EncL& w = m->enc[l];
        for (int i = 0; i < 4; ++i) e.site[i] = site++;
        CK(attn_block_fwd(c, w.sa, a.x16[l], nullptr, a.rows_e, 0, a.Tp, a.Tp, true, false, a.enc_lens, e.qkv, nullptr, e.ao, e.lse,
                          a.x32[l], e.s1, e.site[0], e.site[1]));
        CK(ln_fwd(c, w.n1, e.s1, e.x1_32, e.x1_16, e.m1, e.r1, a.rows_e));
        CK(ffn_fwd(c, w.l1, w.l2, e.x1_16, e.x1_32, a.rows_e, e.f, e.s2, e.site[2], e.site[3]));
        CK(ln_fwd(c, w.n2, e.s2, a.x32[l + 1], a.x16[l + 1], e.m2, e.r2, a.rows_e));
    }
    CK(ln_fwd(c, m->enc_norm, a.x32[m->NE], nullptr, a.mem16, a.mf, a.rf, a.rows_e));
    return 0;
}

// K|V projections of the encoder memory for the cross-attention of EVERY decoder layer: one GEMM, N = ND*2E
static int project_memory_kv(Ctx& c) {
    masr_model* m = c.m; Acts& a = m->acts;
    GemmArgs h = lin_fwd_args(a.mem16, m->E, m->kv_k16, a.rows_e, m->NK, m->E, m->kv_bias);
    h.C16 = a.kv_all; h.ldc16 = m->NK;
    return gemm(c, h);
}

static int forward_decoder(Ctx& c, bool project_kv = true, bool logits_f32 = false) {
    masr_model* m = c.m; Acts& a = m->acts; hipStream_t s = c.s; const float* P = m->P;
    const int E = m->E, L = a.L;
    uint32_t site = 100;
    a.site_emb = site++;
    if (project_kv) CK(project_memory_kv(c));
    { Prof p(m, MASR_PROF_MISC, s); CK(mk_embed_fwd(a.tok_in, P + m->embed_w, m->pe, a.y32[0], a.y16[0], a.B, L, E, c.p_pos, c.seed, a.site_emb, s, c.seed_ptr)); }
    for (int l = 0; l < m->ND; ++l) {
        DecAct& d = a.dec[l]; const DecL& w = m->dec[l];
        for (int i = 0; i < 6; ++i) d.site[i] = site++;
        LnSumArgs ks{};                                        // (k-split GEMMs: their partial products are summed by the LayerNorm behind them)
        CK(attn_block_fwd(c, w.sa, a.y16[l], nullptr, a.rows_d, 0, L, L, true, true, nullptr, d.qkv, nullptr, d.ao, d.lse_s, a.y32[l], d.s1,
                          d.site[0], d.site[1], &ks));
        CK(ln_fwd(c, w.n1, d.s1, d.y1_32, d.y1_16, d.m1, d.r1, a.rows_d, &ks));
        CK(attn_block_fwd(c, w.ca, d.y1_16, a.mem16, a.rows_d, a.rows_e, L, a.Tp, false, false, a.enc_lens, d.q, d.kv, d.co, d.lse_c, d.y1_32,
                          d.s2, d.site[2], d.site[3], &ks));
        CK(ln_fwd(c, w.n2, d.s2, d.y2_32, d.y2_16, d.m2, d.r2, a.rows_d, &ks));
        CK(ffn_fwd(c, w.l1, w.l2, d.y2_16, d.y2_32, a.rows_d, d.f, d.s3, d.site[4], d.site[5], &ks));
        CK(ln_fwd(c, w.n3, d.s3, a.y32[l + 1], a.y16[l + 1], d.m3, d.r3, a.rows_d, &ks));
    }
    if (logits_f32) {
        // greedy decode: the last projection in fp32 on the master weights (see mk_logits_f32); layer 0's pre-LayerNorm sum is free by now
        float* yf32 = a.dec[0].s1;
        CK(ln_fwd(c, m->dec_norm, a.y32[m->ND], yf32, nullptr, a.mdf, a.rdf, a.rows_d));
        return mk_logits_f32(yf32, P + m->ct.w, P + m->ct.b, a.logits, m->Cp, a.rows_d, m->C, E, c.s);
    }
    CK(ln_fwd(c, m->dec_norm, a.y32[m->ND], nullptr, a.yf16, a.mdf, a.rdf, a.rows_d));
    GemmArgs g = lin_fwd_args(a.yf16, E, m->ct.k16, a.rows_d, m->C, E, P + m->ct.b);
    g.C32 = a.logits; g.ldc = m->Cp;
    CK(gemm(c, g));
    return 0;
}

// backward of an attention block  s = resid + drop(out_proj(attn(...)))
//   gs32/gs16: d s (bf16 copy already masked with the out-proj dropout site)
//   self : writes d x (fp32) = gs32 + qkv-proj dgrad into gout
//   cross: writes d xq (fp32) = gs32 + q-proj dgrad into gout and accumulates d memory into dmem
static int attn_block_bwd(Ctx& c, const Attn& at, const bf16* xq16, const bf16* xkv16, int rows_q, int rows_kv, int Tq, int Tk, bool self,
                          bool causal, const int* klens, const bf16* qkv_or_q, const bf16* kv, const bf16* ao, const float* lse,
                          const float* gs32, const bf16* gs16, bf16* gao, bf16* gqkv_or_q, bf16* gkv, float* delta, float* gout,
                          float* dmem, int dmem_accumulate, uint32_t site_p, bool split, LnSumArgs* defer = nullptr) {
    masr_model* m = c.m; const int E = m->E; float* G = m->G;
    CK(lin_wgrad(c, gs16, E, ao, E, rows_q, E, E, G + at.out.w, G + at.out.b, split));
    { GemmArgs g = lin_dgrad_args(gs16, E, at.out.t16, E, rows_q, E, E); g.C16 = gao; g.ldc16 = E; CK(gemm(c, g)); }
    AttnArgs a{};
    if (self) {
        a.q = qkv_or_q; a.k = qkv_or_q + E; a.v = qkv_or_q + 2 * E; a.ldq = a.ldk = a.ldv = 3 * E;
        a.dq = gqkv_or_q; a.dk = gqkv_or_q + E; a.dv = gqkv_or_q + 2 * E; a.lddq = a.lddk = a.lddv = 3 * E;
    } else {
        a.q = qkv_or_q; a.ldq = E; a.k = kv; a.v = kv + E; a.ldk = a.ldv = m->NK;
        a.dq = gqkv_or_q; a.lddq = E; a.dk = gkv; a.dv = gkv + E; a.lddk = a.lddv = m->NK;     // this layer's columns of gkv_all
    }
    a.o = const_cast<bf16*>(ao); a.ldo = E; a.lse = const_cast<float*>(lse); a.dout = gao; a.lddo = E; a.delta = delta; a.klens = klens;
    a.B = m->acts.B; a.H = m->H; a.Tq = Tq; a.Tk = Tk; a.hd = m->hd; a.causal = causal; a.drop_p = c.p_drop; a.seed = c.seed; a.seed_ptr = c.seed_ptr; a.site = site_p;
    { Prof p(m, Tk == m->acts.Tp && Tq == Tk ? MASR_PROF_ATTN_ENC : MASR_PROF_ATTN_DEC, c.s); CK(mk_attn_bwd(a, c.s)); }
    if (self) {
        CK(lin_wgrad(c, gqkv_or_q, 3 * E, xq16, E, rows_q, 3 * E, E, G + at.in.w, G + at.in.b, split));
        GemmArgs g = lin_dgrad_args(gqkv_or_q, 3 * E, at.in.t16, 3 * E, rows_q, 3 * E, E);
        g.residual = gs32; g.ldres = E; g.C32 = gout; g.ldc = E;
        const int S = defer ? ksplit_of(m, rows_q, 3 * E) : 0;
        if (S) {                                               // (gout is not written: the next LayerNorm backward takes `defer`)
            *defer = LnSumArgs{m->acts.part, (long)rows_q * E, S, nullptr, gs32, 0.f, 0u, 0u, nullptr, nullptr};
            return gemm(c, ksplit_args(m, g, S, rows_q, E));
        }
        if (defer) defer->n = 0;
        CK(gemm(c, g));
    } else {
        CK(lin_wgrad(c, gqkv_or_q, E, xq16, E, rows_q, E, E, G + at.in.w, G + at.in.b));
        GemmArgs g = lin_dgrad_args(gqkv_or_q, E, at.q_t16, E, rows_q, E, E);
        g.residual = gs32; g.ldres = E; g.C32 = gout; g.ldc = E;
        CK(gemm(c, g));
        // the K|V halves (weight gradients, gradient of the encoder memory) are handled for all layers at once by
        // memory_kv_bwd after the decoder layer loop
        (void)xkv16; (void)rows_kv; (void)dmem; (void)dmem_accumulate;
    }
    return 0;
}

// backward of project_memory_kv for all decoder layers at once: the ND weight gradients dW_l = gkv_l^T mem are ONE
// reduction-major GEMM with M = ND*2E whose output rows are segmented over the layers' in_proj_weight blocks (constant
// distance in the flat gradient buffer), and d(memory) = sum_l gkv_l Wkv_l is ONE GEMM with K = ND*2E
static int memory_kv_bwd(Ctx& c) {
    masr_model* m = c.m; Acts& a = m->acts; float* G = m->G;
    const int E = m->E;
    const DecL& d0 = m->dec[0];
    if (m->wge.n + m->wg.n + m->ND <= WGRAD_GROUP_MAX) {
        // one descriptor per decoder layer in the grouped encoder-row launch (gkv_all stays untouched until the end of the pass)
        for (int l = 0; l < m->ND; ++l)
            CK(lin_wgrad(c, a.gkv_all + (int64_t)l * 2 * E, m->NK, a.mem16, E, a.rows_e, 2 * E, E, G + m->dec[l].ca.in.w + (long)E * E, G + m->dec[l].ca.in.b + E, true));
    } else {
        GemmArgs g = gemm_args();
        g.reduction_major = 1; g.A = a.gkv_all; g.lda = m->NK; g.B = a.mem16; g.ldb = E; g.M = m->NK; g.N = E; g.K = a.rows_e;
        g.C32 = G + d0.ca.in.w + (long)E * E; g.ldc = E; g.colsum = G + d0.ca.in.b + E;
        g.cseg_rows = 2 * E; g.cseg_stride = m->ND > 1 ? m->dec[1].ca.in.w - d0.ca.in.w : 0;
        CK(gemm(c, g));
    }
    GemmArgs h = lin_dgrad_args(a.gkv_all, m->NK, m->kvT, m->NK, a.rows_e, m->NK, E);
    h.C32 = a.dmem32; h.ldc = E;
    CK(gemm(c, h));
    return 0;
}

static int backward(Ctx& c, const float* xs) {
    masr_model* m = c.m; Acts& a = m->acts; hipStream_t s = c.s; float* G = m->G;
    const int E = m->E, L = a.L, B = a.B;
    // ---- output projection.  The weight gradients of the decoder-row Linears (reduction over only B*L rows) are not
    // launched one by one: their operands are kept per layer and ONE grouped launch computes them after the layer loop
    m->wg.n = 0; m->wge.n = 0;
    m->lng.n = 0; m->ln_slab_used = 0;
    CK(lin_wgrad(c, a.dlogits, m->Cp, a.yf16, E, a.rows_d, m->C, E, G + m->ct.w, G + m->ct.b));
    { GemmArgs g = lin_dgrad_args(a.dlogits, m->Cp, m->ct.t16, m->Cp, a.rows_d, m->Cp, E); g.C32 = a.gd_a; g.ldc = E; CK(gemm(c, g)); }
    float *gcur = a.gd_b, *gs = a.gd_a;
    CK(ln_bwd(c, m->dec_norm, a.gd_a, a.y32[m->ND], a.mdf, a.rdf, gcur, nullptr, 0, a.rows_d));
    // ---- decoder layers
    LnSumArgs ks{};                                            // pending partial products of a k-split dgrad (the LayerNorm backward behind it sums them)
    for (int l = m->ND - 1; l >= 0; --l) {
        DecAct& d = a.dec[l]; const DecL& w = m->dec[l]; const DecGrad& dg = a.dgr[l];
        CK(ln_bwd(c, w.n3, gcur, d.s3, d.m3, d.r3, gs, dg.g3, d.site[5], a.rows_d, &ks));
        CK(ffn_bwd(c, w.l1, w.l2, d.y2_16, d.f, gs, dg.g3, a.rows_d, dg.gf, gcur, false, &ks));
        CK(ln_bwd(c, w.n2, gcur, d.s2, d.m2, d.r2, gs, dg.g2, d.site[3], a.rows_d, &ks));
        ks.n = 0;
        CK(attn_block_bwd(c, w.ca, d.y1_16, a.mem16, a.rows_d, a.rows_e, L, a.Tp, false, false, a.enc_lens, d.q, d.kv, d.co, d.lse_c, gs,
                          dg.g2, a.gao_d, dg.gq, a.gkv_all + (int64_t)l * 2 * E, a.delta_d, gcur, a.dmem32, 0, d.site[2], false));
        CK(ln_bwd(c, w.n1, gcur, d.s1, d.m1, d.r1, gs, dg.g1, d.site[1], a.rows_d));
        // (layer 0's input gradient goes to the embedding backward, not to a LayerNorm: its q/k/v dgrad runs whole)
        CK(attn_block_bwd(c, w.sa, a.y16[l], nullptr, a.rows_d, 0, L, L, true, true, nullptr, d.qkv, nullptr, d.ao, d.lse_s, gs, dg.g1,
                          a.gao_d, dg.gqkv, nullptr, a.delta_d, gcur, nullptr, 0, d.site[0], false, l > 0 ? &ks : nullptr));
    }
    CK(memory_kv_bwd(c));
    float* g_dec_in = gcur;                                  // d(decoder input): consumed by embed_bwd after the split-K combine
    // ---- encoder
    gcur = a.ge_b; gs = a.ge_a;
    CK(ln_bwd(c, m->enc_norm, a.dmem32, a.x32[m->NE], a.mf, a.rf, gcur, nullptr, 0, a.rows_e));
    for (int l = m->NE - 1; l >= 0; --l) {
        EncAct& e = a.enc[l]; const EncL& w = m->enc[l];
        // (grouped weight gradients read their dY operands at the END of the pass: every layer keeps its own)
        const EncGrad eg = a.egr[l];
        CK(ln_bwd(c, w.n2, gcur, e.s2, e.m2, e.r2, gs, eg.g2, e.site[3], a.rows_e));
        CK(ffn_bwd(c, w.l1, w.l2, e.x1_16, e.f, gs, eg.g2, a.rows_e, eg.gf, gcur, true));
        CK(ln_bwd(c, w.n1, gcur, e.s1, e.m1, e.r1, gs, eg.g1, e.site[1], a.rows_e));
        CK(attn_block_bwd(c, w.sa, a.x16[l], nullptr, a.rows_e, 0, a.Tp, a.Tp, true, false, a.enc_lens, e.qkv, nullptr, e.ao, e.lse, gs, eg.g1,
                          a.gao_e, eg.gqkv, nullptr, a.delta_e, gcur, nullptr, 0, e.site[0], true));
    }
    // ---- vgg2enc (through the positional dropout)
    { Prof p(m, MASR_PROF_MISC, s); CK(mk_cast_dropout(gcur, a.ge16, (long)a.rows_e * E, c.p_pos, c.seed, a.site_v2e, s, c.seed_ptr)); }
    CK(lin_wgrad(c, a.ge16, E, a.p2, m->F, a.rows_e, E, m->F, a.v2e_g32, G + m->v2e.b, true));
    CK(flush_wgrads(c));                                     // every Linear weight gradient of the step, one grid
    { GemmArgs g = lin_dgrad_args(a.ge16, E, m->v2e.t16, E, a.rows_e, E, m->F); g.C16 = a.dp2; g.ldc16 = m->F; CK(gemm(c, g)); }
    // ---- VGG
    FoldJobs folds{};
    // The two maps behind a max-pool, d(a4) and d(a2), are never materialised: their consumers -- the weight-gradient kernels and the
    // dgrad kernels -- take the POOLED gradient + the one-byte pool codes of the forward launch and expand the 2 x 2 windows while staging
    // (a quarter of the gradient bytes; the maxpool backward launches and their 338 MB per step are gone)
    auto wgrad = [&](const bf16* in, const bf16* dy, const Conv& cv, int H, int W, const bf16* dy_pooled = nullptr, const uint8_t* idx = nullptr) -> int {
        const int k = (int)(&cv - &m->conv[1]);
        ConvWgradArgs wa{}; wa.in = in; wa.dy = dy; wa.dw = G + cv.w; wa.db = G + cv.b; wa.slab = a.cw_slab[k]; wa.B = B; wa.H = H; wa.W = W; wa.CIN = cv.CI; wa.COUT = cv.CO;
        wa.dy_pooled = dy_pooled; wa.pool_idx = idx;
        { Prof p(m, MASR_PROF_CONV2_WGRAD + k, s); CK(mk_conv3x3_wgrad(wa, s, 1)); }     // the partial slabs; their reduce rides in the fold launch below
        folds.conv[folds.nconv++] = {a.cw_slab[k], mk_conv3x3_wgrad_nsplit(wa), G + cv.w, G + cv.b, cv.CI, cv.CO};
        return 0;
    };
    auto dgrad = [&](const bf16* dy, const Conv& cv, bf16* out, int H, int W, const bf16* dy_pooled = nullptr, const uint8_t* idx = nullptr) -> int {
        Prof p(m, MASR_PROF_CONV2_DGRAD + (int)(&cv - &m->conv[1]), s);
        ConvArgs ca{}; ca.sched = m->conv_sched; ca.in = dy; ca.in_pooled = dy_pooled; ca.in_idx = idx; ca.wk = cv.d16; ca.out = out; ca.B = B; ca.H = H; ca.W = W;
        ca.CIN = cv.CO; ca.COUT = cv.CI;
        if (&cv == &m->conv[3]) { ca.mask = a.a3; ca.mask_bits = a.a3_bits; }      // conv3's ReLU mask: the sign words its forward launch wrote
        if (&cv == &m->conv[1]) {
            // d(conv1 output) is consumed only by conv1's weight gradient: contracted inside the dgrad epilogue, never stored
            ca.mask = a.a1; ca.mask_bits = a.a1_bits; ca.out = nullptr; ca.x1 = xs; ca.w1_slab = a.c1_slab;
        }
        return mk_conv3x3(ca, s);
    };
    CK(wgrad(a.a3, nullptr, m->conv[3], a.H2, a.W2, a.dp2, a.i2));
    CK(dgrad(nullptr, m->conv[3], a.da3, a.H2, a.W2, a.dp2, a.i2));
    CK(wgrad(a.p1, a.da3, m->conv[2], a.H2, a.W2));
    CK(dgrad(a.da3, m->conv[2], a.dp1, a.H2, a.W2));
    CK(wgrad(a.a1, nullptr, m->conv[1], a.T, a.D, a.dp1, a.i1));
    CK(dgrad(nullptr, m->conv[1], nullptr, a.T, a.D, a.dp1, a.i1));
    // ---- every fold of the pass as ONE launch (fold.hip): the conv / conv1 slab reduces, the LayerNorm dgamma / dbeta partials, vgg2enc's weight
    // gradient back in the reference's feature order, and the embedding rows added into the (tied) table -- after the grouped launch wrote it
    folds.E = E;
    folds.conv1 = {a.c1_slab, mk_conv1_wgrad_fused_rows(B, a.T, a.D), G + m->conv[0].w, G + m->conv[0].b};
    folds.unperm = {a.v2e_g32, G + m->v2e.w, E, 128, m->Dp};
    folds.embed = {a.tok_order, a.tok_start, g_dec_in, G + m->embed_w, m->C, E, m->cfg.tie_weights ? 1 : 0, c.p_pos, c.seed, a.site_emb, c.seed_ptr};
    folds.ln = m->lng;
    { Prof p(m, MASR_PROF_CONV1_WGRAD, s); CK(mk_backward_folds(folds, s)); }
    m->lng.n = 0; m->ln_slab_used = 0;
    return 0;
}

int masr_run_batch(masr_model* m, const float* xs, const int64_t* ilens, const int64_t* ys_flat, const int64_t* olens, int B, int T,
                   int flags, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!m->P) { mk_set_error("masr_run_batch", "not bound"); return -1; }
    if (B <= 0 || T < 4) { mk_set_error("masr_run_batch", "need B >= 1 and T >= 4"); return -1; }
    const bool train = (flags & MASR_TRAIN) != 0;
    int maxo = 0; int64_t ntot = 0;
    for (int b = 0; b < B; ++b) { if (olens[b] > maxo) maxo = (int)olens[b]; ntot += olens[b] + 1; }
    const int L = maxo + 1;
    Arena ar{m->ws, m->ws_bytes, m->persist_bytes};
    plan_acts(m, ar, m->acts, B, T, L, train);
    if (ar.off > m->ws_bytes) { mk_set_error("masr_run_batch", "workspace too small (see masr_workspace_bytes)"); return -2; }
    Acts& a = m->acts; m->have_acts = true;
    const int64_t stage_n = (int64_t)3 * B * L + B + 8 + m->C + 1;        // tok_in | gold | enc_lens | meta | tok_order | tok_start
    if (stage_n > m->stage_ints) {
        // the pinned staging ring grows with the batch (B * L) and the vocabulary (C): drain the copies in flight, then re-allocate
        for (auto& ev : m->stage_ev) HIP_CHECK_RET(hipEventSynchronize(ev));
        int* grown = nullptr;
        const int64_t want = stage_n + stage_n / 2;
        HIP_CHECK_RET(hipHostMalloc((void**)&grown, sizeof(int) * want * 4, hipHostMallocDefault));
        hipHostFree(m->h_stage);
        m->h_stage = grown; m->stage_ints = want;
    }
    // ---- MyTransformer.preprocess (:124-141): ys_in = [sos]+y padded with eos, ys_out = y+[eos] padded with -1
    const int slot = m->stage_slot; m->stage_slot = (slot + 1) & 3;
    HIP_CHECK_RET(hipEventSynchronize(m->stage_ev[slot]));
    int* h = m->h_stage + (int64_t)slot * m->stage_ints;
    int* h_in = h; int* h_out = h + (int64_t)B * L; int* h_len = h + (int64_t)2 * B * L;
    const int sos = 0, eos = m->C - 1;
    int64_t off = 0;
    for (int b = 0; b < B; ++b) {
        const int n = (int)olens[b];
        for (int l = 0; l < L; ++l) { h_in[b * L + l] = eos; h_out[b * L + l] = -1; }
        h_in[b * L] = sos;
        for (int l = 0; l < n; ++l) {
            const int tok = (int)ys_flat[off + l];
            if (tok < 0 || tok >= m->C) { mk_set_error("masr_run_batch", "label out of range"); return -1; }
            h_in[b * L + l + 1] = tok; h_out[b * L + l] = tok;
        }
        h_out[b * L + n] = eos;
        off += n;
        h_len[b] = (int)(ilens[b] / 4);                             // enc_lens = floor(ilens/4) (:117)
        if (h_len[b] < 1 || ilens[b] > T) { mk_set_error("masr_run_batch", "ilens must be in [4, T]"); return -1; }
    }
    Ctx c{m, s, (uint32_t)(m->seed * 0x9E3779B97F4A7C15ull >> 32) + (uint32_t)m->step * 7919u, train,
          train ? m->cfg.dropout : 0.f, train ? m->cfg.pos_dropout : 0.f};
    m->step++;
    const float inv_ntot = 1.0f / (float)ntot;
    std::memcpy(h_len + B, &c.seed, 4); std::memcpy(h_len + B + 1, &inv_ntot, 4);     // Acts::meta
    {   // the decoder-input positions grouped by token (counting sort, stable: ascending position inside a token) for the embedding backward.
        // Only the positions 0 .. olens[b] of an utterance: behind them the inputs are eos padding whose gradient is exactly zero (their
        // outputs carry no loss, and the causal mask keeps every valid output from reading them) -- hundreds of hits on ONE table row
        // that a single workgroup column would sum for nothing.
        int* h_order = h_len + B + 8; int* h_start = h_order + (int64_t)B * L;
        const int V = m->C;
        for (int v = 0; v <= V; ++v) h_start[v] = 0;
        for (int b = 0; b < B; ++b) for (int l = 0; l <= (int)olens[b]; ++l) h_start[h_in[b * L + l] + 1]++;
        for (int v = 0; v < V; ++v) h_start[v + 1] += h_start[v];
        // (fill with a running cursor kept in the start array itself, then shift it back)
        for (int b = 0; b < B; ++b) for (int l = 0; l <= (int)olens[b]; ++l) h_order[h_start[h_in[b * L + l]]++] = b * L + l;
        for (int v = V; v > 0; --v) h_start[v] = h_start[v - 1];
        h_start[0] = 0;
    }
    HIP_CHECK_RET(hipMemcpyAsync(a.tok_in, h, sizeof(int) * (size_t)stage_n, hipMemcpyHostToDevice, s));   // tok_in | gold | enc_lens | meta | tok_order | tok_start
    HIP_CHECK_RET(hipEventRecord(m->stage_ev[slot], s));

    auto run = [&](Ctx& cc) -> int {
        m->n_ksplit = 0;
        CK(forward_encoder(cc, xs));
        CK(forward_decoder(cc));
        { Prof p(m, MASR_PROF_MISC, s);
          CK(mk_ls_ce(a.logits, m->Cp, a.gold, a.rows_d, m->C, m->cfg.label_smoothing, inv_ntot, a.dlogits, a.row_loss, a.row_correct,
                      m->stats, s, cc.inv_ptr)); }
        if (train) CK(backward(cc, xs));
        return 0;
    };
    // ---- a batch shape seen twice in a row is captured once and replayed from then on (everything that changes from step to
    // step -- tokens, lengths, dropout seed, 1/n_total -- reaches the kernels through the upload above)
    const bool graphs_on = m->step_graphs_on;
    const int key[4] = {B, T, L, train ? 1 : 0};
    const bool repeat = !memcmp(key, m->last_key, sizeof key) && m->last_xs == (const void*)xs;
    memcpy(m->last_key, key, sizeof key); m->last_xs = xs;
    if (!graphs_on || s == nullptr || m->prof || !repeat) { ++m->n_direct; return run(c); }
    masr_model::StepGraph* sg = nullptr;
    for (auto& g : m->step_graphs)
        if (g.B == B && g.T == T && g.L == L && g.train == key[3] && g.ws == m->ws && g.P == m->P && g.xs == (const void*)xs) { sg = &g; break; }
    if (!sg) {
        if (m->step_graphs.size() >= 8) {                            // evict the least recently used (nothing of it may be in flight)
            HIP_CHECK_RET(hipStreamSynchronize(s));
            size_t lru = 0;
            for (size_t i = 1; i < m->step_graphs.size(); ++i) if (m->step_graphs[i].used < m->step_graphs[lru].used) lru = i;
            hipGraphExecDestroy(m->step_graphs[lru].e); hipGraphDestroy(m->step_graphs[lru].g);
            m->step_graphs.erase(m->step_graphs.begin() + lru);
        }
        masr_model::StepGraph ng{B, T, L, key[3], m->ws, m->P, xs, nullptr, nullptr, 0};
        Ctx cc = c; cc.seed_ptr = a.meta; cc.inv_ptr = reinterpret_cast<const float*>(a.meta + 1);
        HIP_CHECK_RET(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        const int rc = run(cc);
        const hipError_t e = hipStreamEndCapture(s, &ng.g);
        if (rc || e != hipSuccess) { mk_set_error("masr_run_batch", "stream capture of the step failed"); return -1; }
        HIP_CHECK_RET(hipGraphInstantiate(&ng.e, ng.g, nullptr, nullptr, 0));
        m->step_graphs.push_back(ng);
        sg = &m->step_graphs.back();
        ++m->n_captured;
    }
    sg->used = ++m->graph_clock;
    HIP_CHECK_RET(hipGraphLaunch(sg->e, s));
    ++m->n_replayed;
    return 0;
}

void masr_set_step_graphs(masr_model* m, int on) { m->step_graphs_on = on != 0; }
void masr_set_split_wgrad_launches(masr_model* m, int on) { if ((on != 0) != m->split_wgrad) drop_step_graphs(m); m->split_wgrad = on != 0; }
void masr_set_ksplit(masr_model* m, int on) { if ((on != 0) != m->ksplit) drop_step_graphs(m); m->ksplit = on != 0; }
void masr_set_drop_nan_grads(masr_model* m, int on) { m->drop_nan_grads = on != 0; }
void masr_step_counters(const masr_model* m, int64_t out[4]) { out[0] = m->n_direct; out[1] = m->n_captured; out[2] = m->n_replayed; out[3] = m->n_ksplit; }

int masr_read_stats(masr_model* m, float out[4], void* stream) {
    hipStream_t s = (hipStream_t)stream;
    HIP_CHECK_RET(hipMemcpyAsync(m->h_stats, m->stats, sizeof(float) * 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK_RET(hipStreamSynchronize(s));
    for (int i = 0; i < 4; ++i) out[i] = m->h_stats[i];
    return 0;
}

const float* masr_stats_device(masr_model* m) { return m ? m->stats : nullptr; }
int64_t masr_stats_post(masr_model* m, void* stream) {
    if (!m->h_ring) { mk_set_error("masr_stats_post", "not bound"); return -1; }
    const int slot = (int)(m->ring_next % masr_model::RING);
    if (m->ring_used[slot]) HIP_CHECK_RET(hipEventSynchronize(m->ring_ev[slot]));     // the block's previous copy has landed (ticket long dropped or read)
    uint32_t* w = reinterpret_cast<uint32_t*>(m->h_ring + 4 * slot);
    for (int i = 0; i < 4; ++i) w[i] = MASR_STATS_PENDING;
    HIP_CHECK_RET(hipMemcpyAsync(m->h_ring + 4 * slot, m->stats, sizeof(float) * 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_CHECK_RET(hipEventRecord(m->ring_ev[slot], (hipStream_t)stream));
    m->ring_used[slot] = true;
    return m->ring_next++;
}
const float* masr_stats_peek(masr_model* m, int64_t ticket) {
    if (!m->h_ring || ticket < 0 || ticket >= m->ring_next || m->ring_next - ticket > masr_model::RING) { mk_set_error("masr_stats_peek", "unknown or expired ticket"); return nullptr; }
    return m->h_ring + 4 * (ticket % masr_model::RING);
}
int masr_stats_wait(masr_model* m, int64_t ticket, float out[4]) {
    const float* p = masr_stats_peek(m, ticket);
    if (!p) return -1;
    HIP_CHECK_RET(hipEventSynchronize(m->ring_ev[ticket % masr_model::RING]));          // completion AND host visibility of the copy
    for (int i = 0; i < 4; ++i) out[i] = p[i];
    return 0;
}

int masr_last_logits(masr_model* m, const float** logits, const int32_t** gold, int* rows, int* L, int* ld) {
    if (!m->have_acts) { mk_set_error("masr_last_logits", "no forward has run"); return -1; }
    *logits = m->acts.logits; *gold = m->acts.gold; *rows = m->acts.rows_d; *L = m->acts.L; *ld = m->Cp;
    return 0;
}

static float* slab_of(masr_model* m) {
    // optimiser passes may run before any batch: fall back to the tail of the persistent stats block
    return m->have_acts ? m->acts.slab : nullptr;
}

int masr_grad_norm(masr_model* m, void* stream) {
    float* slab = slab_of(m);
    if (!slab) { mk_set_error("masr_grad_norm", "run a batch first"); return -1; }
    Prof p(m, MASR_PROF_OPTIM, (hipStream_t)stream);
    return mk_sumsq(m->G, m->nparams, slab, m->stats + 3, (hipStream_t)stream);
}
int masr_clip_sgd_step(masr_model* m, float* mom, float max_norm, float lr, float momentum, int nesterov, int first_step, void* stream) {
    CK(masr_grad_norm(m, stream));
    { Prof p(m, MASR_PROF_OPTIM, (hipStream_t)stream);
      CK(mk_clip_sgd(m->P, m->G, mom, m->nparams, m->stats + 3, max_norm, lr, momentum, nesterov, first_step, (hipStream_t)stream)); }
    return masr_refresh(m, stream);
}
int masr_clip_grads(masr_model* m, float max_norm, void* stream) {
    CK(masr_grad_norm(m, stream));
    return mk_clip_scale(m->G, m->nparams, m->stats + 3, max_norm, (hipStream_t)stream, m->drop_nan_grads);
}
int masr_clip_scale_flat(float* buf, int64_t n, const float* norm, float max_norm, void* stream) {
    return mk_clip_scale(buf, n, norm, max_norm, (hipStream_t)stream);
}
int masr_clip_accumulate(masr_model* m, float* updates, float max_norm, void* stream) {
    CK(masr_grad_norm(m, stream));
    return mk_clip_axpy(updates, m->G, m->nparams, m->stats + 3, max_norm, (hipStream_t)stream, m->drop_nan_grads);
}
int masr_adam_step(float* p, const float* g, float* ea, float* eas, int64_t n, float lr, float b1, float b2, float eps, int step, void* stream) {
    return mk_adam(p, g, ea, eas, n, lr, b1, b2, eps, step, 0.f, 0, (hipStream_t)stream);
}
int masr_adam_step_guarded(masr_model* m, float* p, const float* g, float* ea, float* eas, int64_t n, float lr_a, int t_a, float lr_b, int t_b,
                           float b1, float b2, float eps, float weight_decay, int decoupled, int slot, void* stream) {
    Prof pr(m, MASR_PROF_OPTIM, (hipStream_t)stream);
    return mk_adam_guarded(p, g, ea, eas, n, lr_a, t_a, lr_b, t_b, b1, b2, eps, weight_decay, decoupled, m->stats + 3,
                           reinterpret_cast<int*>(m->stats + 16), slot, (hipStream_t)stream);
}
int masr_adam_sum_step(float* p, const float* const* grads, int n_grads, float gscale, float* ea, float* eas, int64_t n, float lr, float b1,
                       float b2, float eps, int step, void* stream) {
    return mk_adam_sum(p, grads, n_grads, gscale, ea, eas, n, lr, b1, b2, eps, step, (hipStream_t)stream);
}
int masr_sum_n(float* out, const float* const* grads, int n_grads, float scale, int64_t n, void* stream) {
    return mk_sum_n(out, grads, n_grads, scale, n, (hipStream_t)stream);
}
int masr_adamw_step(float* p, const float* g, float* ea, float* eas, int64_t n, float lr, float b1, float b2, float eps, float weight_decay,
                    int decoupled, int step, void* stream) {
    return mk_adam(p, g, ea, eas, n, lr, b1, b2, eps, step, weight_decay, decoupled, (hipStream_t)stream);
}
int masr_radam_step(float* p, const float* g, float* ea, float* eas, int64_t n, float lr, float b1, float b2, float eps, float weight_decay,
                    int step, int variant, void* stream) {
    return mk_radam(p, g, ea, eas, n, lr, b1, b2, eps, step, weight_decay, variant, (hipStream_t)stream);
}
int masr_sgd_step(float* p, const float* g, float* mom, int64_t n, float lr, float momentum, int nesterov, int first_step, void* stream) {
    return mk_clip_sgd(p, g, mom, n, nullptr, 0.f, lr, momentum, nesterov, first_step, (hipStream_t)stream);
}
int masr_scale(float* x, int64_t n, float a, void* stream) { return mk_scale(x, n, a, (hipStream_t)stream); }
int masr_axpy(float* y, const float* x, int64_t n, float a, void* stream) { return mk_axpy(y, x, n, a, (hipStream_t)stream); }
int masr_copy(float* dst, const float* src, int64_t n, void* stream) {
    HIP_CHECK_RET(hipMemcpyAsync(dst, src, sizeof(float) * (size_t)n, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

// One incremental decode step (the newest target position of every utterance) -- SURVEY 8(f).1.  Every launch below has
// step-independent arguments; the step itself lives in *a.step_dev, so the sequence is captured once and replayed.
static int decode_step(Ctx& c, int* out) {
    masr_model* m = c.m; Acts& a = m->acts; hipStream_t s = c.s; const float* P = m->P;
    const int E = m->E, Fi = m->Fi, B = a.B;
    auto lin = [&](const bf16* x, long ldx, const bf16* wk, int N, int K, const float* bias) {
        SkinnyArgs g{}; g.A = x; g.lda = ldx; g.W = wk; g.ldw = K; g.M = B; g.N = N; g.K = K; g.bias = bias; return g;
    };
    auto att = [&]() { AttnDecodeArgs t{}; t.B = B; t.H = m->H; t.hd = m->hd; t.ldo = E; return t; };
    CK(mk_recog_embed_step(a.step_dev, out, P + m->embed_w, m->pe, a.y32[0], a.y16[0], B, E, 0, s));
    for (int l = 0; l < m->ND; ++l) {
        DecAct& d = a.dec[l]; const DecL& w = m->dec[l];
        // causal self-attention: keys/values of earlier positions live in d.qkv ([B][Ldec][3E], the layout of the full decode)
        SkinnyArgs g = lin(a.y16[l], E, w.sa.in.k16, 3 * E, E, P + w.sa.in.b); g.C16 = a.step_qkv; g.ldc16 = 3 * E;
        CK(mk_skinny_gemm(g, s));
        AttnDecodeArgs t = att();
        t.q = a.step_qkv; t.ldq = 3 * E; t.k = d.qkv + E; t.v = d.qkv + 2 * E; t.ldk = 3 * E; t.kv_batch_stride = (long)a.L * 3 * E;
        t.knew = a.step_qkv + E; t.vnew = a.step_qkv + 2 * E; t.ldnew = 3 * E; t.step = a.step_dev; t.o = d.ao; t.Tk_cap = a.L;
        CK(mk_attn_decode(t, s));
        g = lin(d.ao, E, w.sa.out.k16, E, E, P + w.sa.out.b); g.residual = a.y32[l]; g.ldres = E; g.C32 = d.s1; g.ldc = E;
        CK(mk_skinny_gemm(g, s));
        CK(ln_fwd(c, w.n1, d.s1, d.y1_32, d.y1_16, d.m1, d.r1, B));
        // cross-attention over the encoder memory: d.kv was projected once, before the first step
        g = lin(d.y1_16, E, w.ca.q_k16, E, E, P + w.ca.in.b); g.C16 = d.q; g.ldc16 = E;
        CK(mk_skinny_gemm(g, s));
        t = att();
        t.q = d.q; t.ldq = E; t.k = d.kv; t.v = d.kv + E; t.ldk = m->NK; t.kv_batch_stride = (long)a.Tp * m->NK;
        t.klens = a.enc_lens; t.o = d.co; t.Tk_cap = a.Tp;
        CK(mk_attn_decode(t, s));
        g = lin(d.co, E, w.ca.out.k16, E, E, P + w.ca.out.b); g.residual = d.y1_32; g.ldres = E; g.C32 = d.s2; g.ldc = E;
        CK(mk_skinny_gemm(g, s));
        CK(ln_fwd(c, w.n2, d.s2, d.y2_32, d.y2_16, d.m2, d.r2, B));
        g = lin(d.y2_16, E, w.l1.k16, Fi, E, P + w.l1.b); g.relu = 1; g.C16 = d.f; g.ldc16 = Fi;
        CK(mk_skinny_gemm(g, s));
        g = lin(d.f, Fi, w.l2.k16, E, Fi, P + w.l2.b); g.residual = d.y2_32; g.ldres = E; g.C32 = d.s3; g.ldc = E;
        CK(mk_skinny_gemm(g, s));
        CK(ln_fwd(c, w.n3, d.s3, a.y32[l + 1], a.y16[l + 1], d.m3, d.r3, B));
    }
    // the last projection in fp32 on the master weights (an arg-max follows: mk_logits_f32); layer 0's pre-LayerNorm sum is free by now
    float* yf32 = a.dec[0].s1;
    CK(ln_fwd(c, m->dec_norm, a.y32[m->ND], yf32, nullptr, a.mdf, a.rdf, B));
    CK(mk_logits_f32(yf32, P + m->ct.w, P + m->ct.b, a.logits, m->Cp, B, m->C, E, s));
    CK(mk_recog_argmax_step(a.step_dev, a.logits, m->Cp, out, B, m->C, s));       // also advances *step_dev
    return 0;
}

// shared front half of the two decoders: argument checks, activation plan, enc_lens upload, encoder
static int recog_prepare(masr_model* m, const float* xs, const int64_t* ilens, int B, int T, hipStream_t s, int* Ldec_out) {
    if (!m->P) { mk_set_error("masr_recog", "not bound"); return -1; }
    if (B <= 0 || T < 4) { mk_set_error("masr_recog", "need B >= 1 and T >= 4"); return -1; }
    int Ldec = 0;
    for (int b = 0; b < B; ++b) {
        if (ilens[b] < 4 || ilens[b] > T) { mk_set_error("masr_recog", "ilens must be in [4, T]"); return -1; }
        if ((int)(ilens[b] / 4) > Ldec) Ldec = (int)(ilens[b] / 4);
    }
    Arena ar{m->ws, m->ws_bytes, m->persist_bytes};
    plan_acts(m, ar, m->acts, B, T, Ldec, false);
    if (ar.off > m->ws_bytes) { mk_set_error("masr_recog", "workspace too small (masr_workspace_bytes(B, T, max(ilens)/4))"); return -2; }
    Acts& a = m->acts; m->have_acts = true;
    const int slot = m->stage_slot; m->stage_slot = (slot + 1) & 3;
    HIP_CHECK_RET(hipEventSynchronize(m->stage_ev[slot]));
    int* h_len = m->h_stage + (int64_t)slot * m->stage_ints;
    for (int b = 0; b < B; ++b) h_len[b] = (int)(ilens[b] / 4);
    HIP_CHECK_RET(hipMemcpyAsync(a.enc_lens, h_len, sizeof(int) * (size_t)B, hipMemcpyHostToDevice, s));
    HIP_CHECK_RET(hipEventRecord(m->stage_ev[slot], s));
    Ctx c{m, s, 0u, false, 0.f, 0.f};
    CK(forward_encoder(c, xs));
    *Ldec_out = Ldec;
    return 0;
}

int masr_recog_full(masr_model* m, const float* xs, const int64_t* ilens, int B, int T, int32_t* out, void* stream) {
    // MyTransformer.recog (mono_transformer_torch.py:143-176) literally: the encoder runs once; then, for step = 1 .. max(enc_lens),
    // the WHOLE prefix [sos, out_0 .. out_{step-2}] is decoded again (no KV cache, exactly as the reference) and every
    // position's arg-max becomes the new `out`.  The result after the last step is out[Ldec][B].
    hipStream_t s = (hipStream_t)stream;
    int Ldec = 0;
    { const int rc = recog_prepare(m, xs, ilens, B, T, s, &Ldec); if (rc) return rc; }
    Acts& a = m->acts;
    Ctx c{m, s, 0u, false, 0.f, 0.f};
    CK(project_memory_kv(c));                               // (the memory does not change between steps)
    for (int step = 1; step <= Ldec; ++step) {
        a.L = step; a.rows_d = B * step;
        CK(mk_recog_build_tok(a.tok_in, out, B, step, 0, s));
        CK(forward_decoder(c, false, true));
        CK(mk_recog_argmax(a.logits, m->Cp, out, B, step, m->C, s));
    }
    m->have_acts = false;                                   // logits/gold views are not meaningful after a decode
    return 0;
}

int masr_recog(masr_model* m, const float* xs, const int64_t* ilens, int B, int T, int32_t* out, void* stream) {
    // Same token sequences as masr_recog_full with O(L) instead of O(L^2) decoder work: the target mask is causal, so the
    // re-decode of earlier positions reproduces what is already in `out`; only the newest position is computed per step,
    // against cached self-attention keys/values and encoder-memory keys/values projected once.  The per-step launch
    // sequence is captured into a hipGraph and replayed Ldec times (direct launches on the legacy NULL stream, while
    // profiling, or with MASR_RECOG_NO_GRAPH set).
    hipStream_t s = (hipStream_t)stream;
    int Ldec = 0;
    { const int rc = recog_prepare(m, xs, ilens, B, T, s, &Ldec); if (rc) return rc; }
    Acts& a = m->acts; const int E = m->E;
    Ctx c{m, s, 0u, false, 0.f, 0.f};
    CK(project_memory_kv(c));
    CK(mk_recog_step_set(a.step_dev, 1, 0, s));
    const bool use_graph = s != nullptr && !m->prof && !getenv("MASR_RECOG_NO_GRAPH");
    if (!use_graph) {
        for (int step = 1; step <= Ldec; ++step) CK(decode_step(c, out));
    } else {
        const int key[3] = {B, T, Ldec}; const void* kp[3] = {m->ws, m->P, out};
        const bool hit = m->dec_exec && !memcmp(key, m->dec_key, sizeof key) && !memcmp(kp, m->dec_key_ptr, sizeof kp);
        if (!hit) {
            if (m->dec_done) HIP_CHECK_RET(hipEventSynchronize(m->dec_done));      // no replay of the old graph in flight
            else HIP_CHECK_RET(hipEventCreateWithFlags(&m->dec_done, hipEventDisableTiming));
            if (m->dec_exec) { hipGraphExecDestroy(m->dec_exec); m->dec_exec = nullptr; }
            if (m->dec_graph) { hipGraphDestroy(m->dec_graph); m->dec_graph = nullptr; }
            HIP_CHECK_RET(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            const int rc = decode_step(c, out);
            const hipError_t e = hipStreamEndCapture(s, &m->dec_graph);
            if (rc || e != hipSuccess) { mk_set_error("masr_recog", "stream capture of the decode step failed"); return -1; }
            HIP_CHECK_RET(hipGraphInstantiate(&m->dec_exec, m->dec_graph, nullptr, nullptr, 0));
            memcpy(m->dec_key, key, sizeof key); memcpy(m->dec_key_ptr, kp, sizeof kp);
        }
        for (int step = 1; step <= Ldec; ++step) HIP_CHECK_RET(hipGraphLaunch(m->dec_exec, s));
        HIP_CHECK_RET(hipEventRecord(m->dec_done, s));
    }
    m->have_acts = false;
    return 0;
}

// Levenshtein distance of two id sequences (host code; the reference's metric imports the `editdistance` C extension,
// src/monitor/metric.py:4,66,87).  Two-row DP, unit costs.
int64_t masr_edit_distance(const int32_t* a, int na, const int32_t* b, int nb) {
    if (na < 0 || nb < 0 || (na > 0 && !a) || (nb > 0 && !b)) { mk_set_error("masr_edit_distance", "bad arguments"); return -1; }
    std::vector<int64_t> row((size_t)nb + 1);
    for (int j = 0; j <= nb; ++j) row[j] = j;
    for (int i = 1; i <= na; ++i) {
        int64_t diag = row[0];
        row[0] = i;
        for (int j = 1; j <= nb; ++j) {
            const int64_t sub = diag + (a[i - 1] != b[j - 1]);
            diag = row[j];
            int64_t v = row[j] + 1;
            if (row[j - 1] + 1 < v) v = row[j - 1] + 1;
            if (sub < v) v = sub;
            row[j] = v;
        }
    }
    return row[nb];
}

int masr_gather_pad(const float* feat, const int64_t* row_start, const int32_t* lens, float* xs, int B, int Tmax, int D, void* stream) {
    return mk_gather_pad(feat, (const long*)row_start, lens, xs, B, Tmax, D, (hipStream_t)stream);
}
int masr_fbank(const float* wav, const int64_t* wav_off, const int64_t* row_off, int B, int max_frames, int n_mel, float* feat, void* stream) {
    return mk_fbank(wav, (const long*)wav_off, (const long*)row_off, B, max_frames, n_mel, feat, (hipStream_t)stream);
}
int64_t masr_fbank_pitch_work_bytes(int64_t total_samples, int B, int max_frames) { return mk_pitch_work_bytes(total_samples, B, max_frames); }
int masr_fbank_pitch(const float* wav, const int64_t* wav_off, const int64_t* row_off, int64_t total_samples, int64_t max_samples, int B, int max_frames,
                     int n_mel, float* feat, void* work, int64_t work_bytes, void* stream) {
    if (mk_fbank(wav, (const long*)wav_off, (const long*)row_off, B, max_frames, n_mel, feat, (hipStream_t)stream, 1) != 0) return -1;
    return mk_pitch(wav, (const long*)wav_off, (const long*)row_off, total_samples, max_samples, B, max_frames, n_mel, feat, work, work_bytes, (hipStream_t)stream);
}
int64_t masr_ctc_work_floats(int T, int B, int maxS) { return mk_ctc_work_floats(T, B, maxS); }
int masr_ctc_status(const float* work, int T, int B, int maxS, void* stream) { return mk_ctc_status(work, T, B, maxS, (hipStream_t)stream); }
int masr_ctc_loss(const float* logits, const int32_t* targets, const int32_t* tgt_off, const int32_t* in_len, const int32_t* tgt_len, int T,
                  int B, int C, int blank, float* nll, float* loss, float* grad, float* work, int maxS, void* stream) {
    return mk_ctc_loss(logits, targets, tgt_off, in_len, tgt_len, T, B, C, blank, nll, loss, grad, work, maxS, (hipStream_t)stream);
}

int masr_profile_enable(masr_model* m, int on) {
    m->prof = on != 0;
    for (int i = 0; i < MASR_PROF_N; ++i) m->prof_used[i] = 0;
    return 0;
}
int masr_profile_read(masr_model* m, float* ms, int* launches) {
    HIP_CHECK_RET(hipDeviceSynchronize());
    for (int i = 0; i < MASR_PROF_N; ++i) {
        float tot = 0.f;
        for (int k = 0; k < m->prof_used[i]; ++k) {
            float t = 0.f;
            hipEventElapsedTime(&t, m->prof_ev[i][k].first, m->prof_ev[i][k].second);
            tot += t;
        }
        ms[i] = tot; launches[i] = m->prof_used[i]; m->prof_used[i] = 0;
    }
    return 0;
}

// ---------------------------------------------------------------- standalone kernel entry points (parity tests)
int masr_test_gemm(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, int M, int N, int K, int reduction_major, const float* bias,
                   int relu, float* C32, int64_t ldc, void* stream) {
    GemmArgs g = gemm_args();
    g.A = (const bf16*)A; g.lda = lda; g.B = (const bf16*)B; g.ldb = ldb; g.M = M; g.N = N; g.K = K; g.reduction_major = reduction_major;
    g.bias = bias; g.relu = relu; g.C32 = C32; g.ldc = ldc;
    return mk_gemm(g, (hipStream_t)stream);
}
int masr_test_dropout_mask(uint32_t seed, uint32_t site, int64_t n, float p, float* out, void* stream) {
    return mk_dropout_mask(out, n, p, seed, site, (hipStream_t)stream);
}
int masr_test_gemm_dropout(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, int M, int N, int K, float drop_p, uint32_t seed,
                           uint32_t site, float* C32, int64_t ldc, void* stream) {
    GemmArgs g = gemm_args();
    g.A = (const bf16*)A; g.lda = lda; g.B = (const bf16*)B; g.ldb = ldb; g.M = M; g.N = N; g.K = K;
    g.drop_p = drop_p; g.seed = seed; g.site = site; g.C32 = C32; g.ldc = ldc;
    return mk_gemm(g, (hipStream_t)stream);
}
int masr_test_attention_dropout(const uint16_t* q, const uint16_t* k, const uint16_t* v, uint16_t* o, float* lse, int B, int H, int Tq, int Tk,
                                int hd, float drop_p, uint32_t seed, uint32_t site, void* stream) {
    const long E = (long)H * hd;
    AttnArgs a{};
    a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.ldq = a.ldk = a.ldv = E; a.o = (bf16*)o; a.ldo = E; a.lse = lse;
    a.B = B; a.H = H; a.Tq = Tq; a.Tk = Tk; a.hd = hd; a.drop_p = drop_p; a.seed = seed; a.site = site;
    return mk_attn_fwd(a, (hipStream_t)stream);
}
int masr_test_gemm_epi(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, int M, int N, int K, const float* bias, int relu,
                       float drop_p, const float* residual, const uint16_t* mask, float* C32, uint16_t* C16, void* stream) {
    GemmArgs g = gemm_args();
    g.A = (const bf16*)A; g.lda = lda; g.B = (const bf16*)B; g.ldb = ldb; g.M = M; g.N = N; g.K = K; g.bias = bias; g.relu = relu;
    g.drop_p = drop_p; g.seed = 1; g.site = 2; g.residual = residual; g.ldres = N; g.mask = (const bf16*)mask; g.ldmask = N;
    g.C32 = C32; g.ldc = N; g.C16 = (bf16*)C16; g.ldc16 = N;
    return mk_gemm(g, (hipStream_t)stream);
}
int masr_test_linear_shadows(const float* P, int64_t src, int N, int K, int ldt, uint16_t* k16, uint16_t* t16, void* stream) {
    if (N <= 0 || K <= 0 || ldt < N || src < 4) { mk_set_error("masr_test_linear_shadows", "N, K > 0, ldt >= N, src >= 4 (the tile pass reads up to three floats in front of a row)"); return -1; }
    ShadowJobs jobs{};
    jobs.n = 1;
    jobs.d[0] = ShadowDesc{src, SH_LINEAR, N, K, ldt, 0, 0, 0};
    jobs.blocks = mk_shadow_blocks(jobs.d[0]);
    jobs.p[0] = (bf16*)k16; jobs.p[1] = (bf16*)t16;
    return mk_all_shadows(P, jobs, (hipStream_t)stream);
}
int masr_test_conv1_fwd(const float* x, const float* w, const float* bias, uint16_t* out, uint64_t* relu_bits, int B, int H, int W, void* stream) {
    return mk_conv1_fwd(x, w, bias, (bf16*)out, B, H, W, (hipStream_t)stream, reinterpret_cast<unsigned long long*>(relu_bits));
}
int masr_test_conv3x3(const uint16_t* in, const uint16_t* wk, const float* bias, int relu, uint16_t* out, int B, int H, int W, int CIN,
                      int COUT, void* stream) {
    ConvArgs a{}; a.in = (const bf16*)in; a.wk = (const bf16*)wk; a.bias = bias; a.relu = relu; a.out = (bf16*)out;
    a.B = B; a.H = H; a.W = W; a.CIN = CIN; a.COUT = COUT;
    return mk_conv3x3(a, (hipStream_t)stream);
}
int masr_test_conv3x3_ex(const uint16_t* in, const uint16_t* wk, const float* bias, int relu, const uint16_t* mask, uint16_t* out,
                         uint16_t* pool_out, int B, int H, int W, int CIN, int COUT, void* stream) {
    ConvArgs a{}; a.in = (const bf16*)in; a.wk = (const bf16*)wk; a.bias = bias; a.relu = relu; a.mask = (const bf16*)mask; a.out = (bf16*)out;
    a.pool_out = (bf16*)pool_out; a.B = B; a.H = H; a.W = W; a.CIN = CIN; a.COUT = COUT;
    return mk_conv3x3(a, (hipStream_t)stream);
}
int masr_test_conv3x3_sign_bits(const uint16_t* in, const uint16_t* wk, const float* bias, int relu, const uint16_t* mask, const uint32_t* mask_bits,
                                uint16_t* out, uint32_t* out_sign_bits, int B, int H, int W, int CIN, int COUT, void* stream) {
    ConvArgs a{}; a.in = (const bf16*)in; a.wk = (const bf16*)wk; a.bias = bias; a.relu = relu; a.mask = (const bf16*)mask;
    a.mask_bits = (const unsigned long long*)mask_bits; a.out = (bf16*)out; a.out_sign_bits = (unsigned long long*)out_sign_bits;
    a.B = B; a.H = H; a.W = W; a.CIN = CIN; a.COUT = COUT;
    return mk_conv3x3(a, (hipStream_t)stream);
}
int masr_test_conv3x3_pool_idx(const uint16_t* in, const uint16_t* wk, const float* bias, uint16_t* out, uint16_t* pool_out, uint8_t* pool_idx,
                               int drop_out, int B, int H, int W, int CIN, int COUT, void* stream) {
    ConvArgs a{}; a.in = (const bf16*)in; a.wk = (const bf16*)wk; a.bias = bias; a.relu = 1; a.out = (bf16*)out;
    a.pool_out = (bf16*)pool_out; a.pool_idx = pool_idx; a.out_optional = drop_out; a.B = B; a.H = H; a.W = W; a.CIN = CIN; a.COUT = COUT;
    return mk_conv3x3(a, (hipStream_t)stream);
}
int masr_test_conv3x3_dgrad_pooled(const uint16_t* dy, const uint16_t* dy_pooled, const uint8_t* pool_idx, const uint16_t* wk, const uint32_t* mask_bits,
                                   uint16_t* out, int B, int H, int W, void* stream) {
    ConvArgs a{}; a.in = (const bf16*)dy; a.in_pooled = (const bf16*)dy_pooled; a.in_idx = pool_idx; a.wk = (const bf16*)wk;
    a.mask = (const bf16*)out; a.mask_bits = (const unsigned long long*)mask_bits; a.out = (bf16*)out;      // (mask: any non-null pointer -- the sign words are what is read)
    a.B = B; a.H = H; a.W = W; a.CIN = 128; a.COUT = 128;
    return mk_conv3x3(a, (hipStream_t)stream);
}
int64_t masr_test_conv1_wgrad_fused_slab_floats(int B, int H, int W) { return mk_conv1_wgrad_fused_slab_floats(B, H, W); }
int masr_test_conv1_wgrad_fused(const uint16_t* dy, const uint16_t* dy_pooled, const uint8_t* pool_idx, const uint16_t* wk, const uint64_t* mask_bits,
                                const float* x1, float* slab, int64_t slab_floats, float* dw1, float* db1, int B, int H, int W, void* stream) {
    if (slab_floats < mk_conv1_wgrad_fused_slab_floats(B, H, W)) { mk_set_error("masr_test_conv1_wgrad_fused", "slab too small"); return -1; }
    ConvArgs a{}; a.in = (const bf16*)dy; a.in_pooled = (const bf16*)dy_pooled; a.in_idx = pool_idx; a.wk = (const bf16*)wk;
    a.mask = (const bf16*)wk; a.mask_bits = (const unsigned long long*)mask_bits; a.x1 = x1; a.w1_slab = slab;
    a.B = B; a.H = H; a.W = W; a.CIN = 64; a.COUT = 64;
    CK(mk_conv3x3(a, (hipStream_t)stream));
    return mk_conv1_wgrad_fused_reduce(slab, B, H, W, dw1, db1, (hipStream_t)stream);
}
int64_t masr_test_conv3x3_wgrad_slab_floats(int B, int H, int W, int CIN, int COUT) { return mk_conv3x3_wgrad_slab_floats(B, H, W, CIN, COUT); }
int masr_test_conv3x3_wgrad(const uint16_t* in, const uint16_t* dy, float* dw, float* slab, int64_t slab_floats, int B, int H, int W, int CIN,
                            int COUT, void* stream) {
    if (slab_floats < mk_conv3x3_wgrad_slab_floats(B, H, W, CIN, COUT)) { mk_set_error("masr_test_conv3x3_wgrad", "slab too small"); return -1; }
    ConvWgradArgs a{}; a.in = (const bf16*)in; a.dy = (const bf16*)dy; a.dw = dw; a.slab = slab; a.B = B; a.H = H; a.W = W; a.CIN = CIN; a.COUT = COUT;
    return mk_conv3x3_wgrad(a, (hipStream_t)stream);
}
int64_t masr_test_layernorm_slab_floats(int rows, int E) { return mk_layernorm_bwd_slab_floats(rows, E); }
int masr_test_layernorm(const float* x, const float* gamma, const float* beta, const float* dy, float* y, uint16_t* y16, float* mean,
                        float* rstd, float* dx, uint16_t* dx16, float* dgamma, float* dbeta, float* slab, int rows, int E, float drop_p,
                        uint32_t seed, uint32_t site, void* stream) {
    if (mk_layernorm_fwd(x, gamma, beta, y, (bf16*)y16, mean, rstd, rows, E, (hipStream_t)stream)) return -1;
    return mk_layernorm_bwd(dy, x, gamma, mean, rstd, dx, (bf16*)dx16, drop_p, seed, site, dgamma, dbeta, slab, rows, E, (hipStream_t)stream, nullptr);
}
int masr_test_ksplit_ln(const uint16_t* A, const uint16_t* B, int rows, int E, int K, int split, const float* bias, const float* residual, float drop_p,
                        uint32_t seed, uint32_t site, float* part, const float* gamma, const float* beta, float* sum_out, float* y32, uint16_t* y16,
                        float* mean, float* rstd, const float* x, float* dx32, uint16_t* dx16, float* slab, void* stream) {
    // the engine's k-split pair (ffn_fwd + ln_fwd / ffn_bwd + ln_bwd): C = A [rows, K] . B [E, K]^T as `split` fp32 partial products in `part`,
    // then the LayerNorm that sums them.  x == null: forward (row = sum + bias, dropout, + residual -> sum_out, y32 / y16, mean, rstd);
    // x given: backward (dy = sum + residual; mean / rstd are inputs; dx32 / dx16 and the [ceil(rows / 4)][2][E] partials of dgamma / dbeta in slab)
    GemmArgs g = gemm_args();
    g.A = (const bf16*)A; g.lda = K; g.B = (const bf16*)B; g.ldb = K; g.M = rows; g.N = E; g.K = K;
    g.C32 = part; g.ldc = E; g.split_k = split; g.split_stride = (long)rows * E;
    CK(mk_gemm(g, (hipStream_t)stream));
    if (!x) {
        const LnSumArgs sm{part, (long)rows * E, split, bias, residual, drop_p, seed, site, nullptr, sum_out};
        return mk_layernorm_fwd_sum(sm, gamma, beta, y32, (bf16*)y16, mean, rstd, rows, E, (hipStream_t)stream);
    }
    const LnSumArgs sm{part, (long)rows * E, split, nullptr, residual, 0.f, 0u, 0u, nullptr, nullptr};
    return mk_layernorm_bwd_sum(sm, x, gamma, mean, rstd, dx32, (bf16*)dx16, drop_p, seed, site, slab, rows, E, (hipStream_t)stream, nullptr);
}
int masr_test_wgrad_grouped(const uint16_t* dy, int64_t lddy, const uint16_t* x, int64_t ldx, float* dW, float* db, float* dW2, float* db2,
                            int rows, int N, int K, void* stream) {
    // two members over the same operands (the second one optional): exercises the descriptor walk of the grouped grid
    WgradGroup grp{};
    grp.n = dW2 ? 2 : 1;
    for (int i = 0; i < grp.n; ++i) {
        WgradDesc& d = grp.p[i];
        d.dy = (const bf16*)dy; d.x = (const bf16*)x; d.dW = i ? dW2 : dW; d.db = i ? db2 : db; d.lddy = (int)lddy; d.ldx = (int)ldx; d.rows = rows; d.N = N; d.K = K;
    }
    return mk_gemm_wgrad_grouped(grp, (hipStream_t)stream);
}
int masr_test_wgrad_grouped_n(const uint16_t* dy, int64_t lddy, const uint16_t* x, int64_t ldx, float* dW, int64_t member_stride, int members,
                              int first_members, int rows, int rows_rest, int N, int K, void* stream) {
    // `members` group members over the SAME operands (member i writes dW + i * member_stride; 0 = all into one buffer): what the grouped
    // launch costs when every panel is resident in L2 / the Infinity Cache (tools/wgrad_probe.py).  first_members > 0: the two-segment
    // tile list of the engine's merged launch -- members [0, first_members) reduce over `rows` rows and are dispatched first, the rest
    // over the first `rows_rest` rows
    WgradGroup grp{};
    grp.n = members < WGRAD_GROUP_MAX ? members : WGRAD_GROUP_MAX;
    for (int i = 0; i < grp.n; ++i) {
        WgradDesc& d = grp.p[i];
        d.dy = (const bf16*)dy; d.x = (const bf16*)x; d.dW = dW + (int64_t)i * member_stride; d.db = nullptr; d.lddy = (int)lddy; d.ldx = (int)ldx;
        d.rows = (first_members > 0 && i >= first_members) ? rows_rest : rows; d.N = N; d.K = K;
    }
    return mk_gemm_wgrad_grouped(grp, (hipStream_t)stream, first_members);
}
int masr_test_conv3x3_wgrad_pooled(const uint16_t* in, const uint16_t* dy_pooled, const uint8_t* pool_idx, float* dw, float* db, float* slab,
                                   int64_t slab_floats, int B, int H, int W, int CIN, int COUT, void* stream) {
    if (slab_floats < mk_conv3x3_wgrad_slab_floats(B, H, W, CIN, COUT)) { mk_set_error("masr_test_conv3x3_wgrad_pooled", "slab too small"); return -1; }
    ConvWgradArgs a{}; a.in = (const bf16*)in; a.dy_pooled = (const bf16*)dy_pooled; a.pool_idx = pool_idx; a.dw = dw; a.db = db; a.slab = slab;
    a.B = B; a.H = H; a.W = W; a.CIN = CIN; a.COUT = COUT;
    return mk_conv3x3_wgrad(a, (hipStream_t)stream);
}
int masr_test_attention_dropout_bwd(const uint16_t* q, const uint16_t* k, const uint16_t* v, const uint16_t* dout, uint16_t* o, uint16_t* dq, uint16_t* dk,
                                    uint16_t* dv, float* lse, const int32_t* klens, int B, int H, int Tq, int Tk, int hd, int causal, float drop_p,
                                    uint32_t seed, uint32_t site, void* stream) {
    const long E = (long)H * hd;
    AttnArgs a{};
    a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.ldq = a.ldk = a.ldv = E; a.o = (bf16*)o; a.ldo = E; a.lse = lse;
    a.klens = klens; a.B = B; a.H = H; a.Tq = Tq; a.Tk = Tk; a.hd = hd; a.causal = causal; a.drop_p = drop_p; a.seed = seed; a.site = site;
    CK(mk_attn_fwd(a, (hipStream_t)stream));
    a.dout = (const bf16*)dout; a.lddo = E; a.dq = (bf16*)dq; a.dk = (bf16*)dk; a.dv = (bf16*)dv; a.lddq = a.lddk = a.lddv = E;
    return mk_attn_bwd(a, (hipStream_t)stream);
}
int masr_test_attention(const uint16_t* q, const uint16_t* k, const uint16_t* v, const uint16_t* dout, uint16_t* o, uint16_t* dq, uint16_t* dk,
                        uint16_t* dv, float* lse, float* delta, const int32_t* klens, int B, int H, int Tq, int Tk, int hd, int causal,
                        void* stream) {
    const long E = (long)H * hd;
    AttnArgs a{};
    a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.ldq = a.ldk = a.ldv = E; a.o = (bf16*)o; a.ldo = E; a.lse = lse;
    a.klens = klens; a.B = B; a.H = H; a.Tq = Tq; a.Tk = Tk; a.hd = hd; a.causal = causal;
    CK(mk_attn_fwd(a, (hipStream_t)stream));
    if (dout) {
        a.dout = (const bf16*)dout; a.lddo = E; a.dq = (bf16*)dq; a.dk = (bf16*)dk; a.dv = (bf16*)dv; a.lddq = a.lddk = a.lddv = E; a.delta = delta;
        CK(mk_attn_bwd(a, (hipStream_t)stream));
    }
    return 0;
}

}  // extern "C"
