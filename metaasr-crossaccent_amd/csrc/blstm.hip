// libmasr BLSTM-CTC engine: the reference's second model family (config/blstm):
//   MonoBLSTM.forward (src/model/blstm/mono_blstm.py:77-92) = BlstmEncoder (src/modules/encoder.py:215-298: VGG
//   1->128->128 pool(ceil) ->256->256 pool(ceil), then RNNP = L x {packed BLSTM(enc_dim), Linear(2 enc_dim -> proj), tanh},
//   pad frames zeroed) + Linear head, and BLSTMTrainer.run_batch (src/blstm_trainer.py:55-85): targets [sos]+y+[eos] with
//   sos = eos = odim-1, log_softmax + nn.CTCLoss(blank 0, mean, zero_infinity), backward.
// Same memory model as engine.hip: one flat fp32 parameter / gradient buffer in the reference's state_dict order, bf16
// operand shadows, a bump-allocated activation arena, every kernel on the caller's stream.  Time sub-sampling between layers
// (encoder.sample_rate, RNNP.forward encoder.py:118-121): layer i's LSTM runs on Ts[i] frames per utterance, its output keeps
// every sub[i]-th frame before the projection, enc_lens -> (enc_lens + 1) / sub[i].  Dropout 0 per layer (a no-op in the
// reference too: nn.LSTM(dropout=..., num_layers=1)).
#include <cstring>
#include <string>
#include <vector>

#include "../../include/masr.h"
#include "../../include/masr_test.h"
#include "kernels.h"

namespace {

struct PInfo { std::string name; int64_t shape[4]; int ndim; int64_t off; int64_t numel; };
struct Arena {
    char* base; int64_t cap, off;
    template <class T> T* get(int64_t n) {
        const int64_t bytes = (n * (int64_t)sizeof(T) + 255) & ~(int64_t)255;
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += bytes;
        return p;
    }
};
struct ConvP { int64_t w, b; int CO, CI; bf16 *k16, *d16; };
struct LstmDir { int64_t wih, whh, bih, bhh; bf16 *wih16, *wihT16, *whh16, *whhT16; float* bias; };
struct Layer { LstmDir d[2]; int64_t btw, btb; int K, N; bf16 *bt16, *btT16; };       // K = LSTM input width, N = projection width
struct LayerAct {
    float *gx[2], *act[2], *c[2]; bf16* y16; float* z32; float* x32; bf16* x16;      // x = tanh(projection) = next layer's input
    bf16* ys16;                                                                       // y16 sub-sampled in time (== y16 where sub == 1)
    bf16 *dzg[2], *hp[2];
};

}  // namespace

struct masr_blstm {
    masr_blstm_config cfg;
    int D, C, H, KP, L, Cp8;
    std::vector<PInfo> params; int64_t nparams = 0;
    ConvP conv[4]; std::vector<Layer> layers; int64_t headw, headb; bf16 *head16 = nullptr, *headT16 = nullptr;
    float *P = nullptr, *G = nullptr; char* ws = nullptr; int64_t ws_bytes = 0, persist_bytes = 0;
    unsigned* conv_sched = nullptr;
    float* stats = nullptr; float* h_stats = nullptr; int* h_stage = nullptr; hipEvent_t stage_ev = nullptr;
    // activations of the last batch
    int B = 0, T = 0, H2 = 0, W2 = 0, Tp = 0, Dp = 0, F = 0; int64_t rows = 0;
    int sub[8] = {1, 1, 1, 1, 1, 1, 1, 1}; int Ts[9] = {0};                         // Ts[i] = frames per utterance entering layer i, Ts[L] = leaving the encoder
    int* lens_l[9] = {nullptr};                                                      // enc_lens entering layer i (device [B]); lens_l[L]: the encoder's output lengths
    bf16 *c1, *c2, *p1, *c3, *c4, *p2;
    std::vector<LayerAct> act;
    float* logits; float* dlogits; bf16* dl16; int *lens, *tgt, *tgt_off, *tgt_len; float *nll, *ctc_work; int maxS = 0;
    bf16 *h16[2][2]; float* cstate[2];
    unsigned long long* rec_words = nullptr;                                         // granule exchange of the resident recurrence (lstm_rec.hip)
    bool resident = true;                                                            // masr_blstm_set_resident_recurrence
    int test_stall = 0;                                                              // masr_test_blstm_stall (include/masr_test.h)
    float *dx32, *dy32, *dys32, *wtmp, *slab; int64_t slab_floats = 0;
    bf16 *dp2, *dc4, *dc3, *dp1, *dc2, *dc1;
    bool have = false;
};

namespace {

#define CK(expr) do { if ((expr) != 0) return -1; } while (0)

int64_t add_param(masr_blstm* m, const std::string& name, std::initializer_list<int64_t> shape) {
    PInfo p; p.name = name; p.ndim = (int)shape.size(); p.numel = 1;
    int i = 0; for (auto s : shape) { p.shape[i++] = s; p.numel *= s; }
    for (; i < 4; ++i) p.shape[i] = 1;
    // every tensor starts on a 4-float boundary of the flat buffer (16-byte rows for the vectorised kernels)
    m->nparams = (m->nparams + 3) / 4 * 4;
    p.off = m->nparams; m->nparams += p.numel;
    m->params.push_back(p);
    return p.off;
}

void plan_persistent(masr_blstm* m, Arena& ar) {
    for (int i = 1; i < 4; ++i) {
        m->conv[i].k16 = ar.get<bf16>((int64_t)m->conv[i].CO * 9 * m->conv[i].CI);
        m->conv[i].d16 = ar.get<bf16>((int64_t)m->conv[i].CO * 9 * m->conv[i].CI);
    }
    const int H = m->H, G = 4 * H;
    for (auto& l : m->layers) {
        for (int d = 0; d < 2; ++d) {
            l.d[d].wih16 = ar.get<bf16>((int64_t)G * l.K); l.d[d].wihT16 = ar.get<bf16>((int64_t)l.K * G);
            l.d[d].whh16 = ar.get<bf16>((int64_t)G * m->KP); l.d[d].whhT16 = ar.get<bf16>((int64_t)H * G);
            l.d[d].bias = ar.get<float>(G);
        }
        l.bt16 = ar.get<bf16>((int64_t)l.N * 2 * H); l.btT16 = ar.get<bf16>((int64_t)2 * H * ((l.N + 7) / 8 * 8));
    }
    m->head16 = ar.get<bf16>((int64_t)m->Cp8 * m->layers.back().N);
    m->headT16 = ar.get<bf16>((int64_t)m->layers.back().N * m->Cp8);
    m->stats = ar.get<float>(64);
    m->conv_sched = ar.get<unsigned>(64);
}

void plan_acts(masr_blstm* m, Arena& ar, int B, int T, int maxS) {
    m->B = B; m->T = T; m->H2 = (T + 1) / 2; m->W2 = (m->D + 1) / 2; m->Tp = (m->H2 + 1) / 2; m->Dp = (m->W2 + 1) / 2;
    m->F = 256 * m->Dp; m->rows = (int64_t)B * m->Tp; m->maxS = maxS;
    m->Ts[0] = m->Tp;
    for (int i = 0; i < m->L; ++i) m->Ts[i + 1] = (m->Ts[i] + m->sub[i] - 1) / m->sub[i];
    const int64_t P1 = (int64_t)B * T * m->D, P2 = (int64_t)B * m->H2 * m->W2, R = m->rows;
    const int H = m->H, G = 4 * H;
    m->c1 = ar.get<bf16>(P1 * 128); m->c2 = ar.get<bf16>(P1 * 128); m->p1 = ar.get<bf16>(P2 * 128);
    m->c3 = ar.get<bf16>(P2 * 256); m->c4 = ar.get<bf16>(P2 * 256); m->p2 = ar.get<bf16>(R * m->F);
    m->act.resize(m->L);
    for (int i = 0; i < m->L; ++i) {
        LayerAct& a = m->act[i]; const Layer& l = m->layers[i];
        for (int d = 0; d < 2; ++d) {
            a.gx[d] = ar.get<float>(R * G); a.act[d] = ar.get<float>(R * G); a.c[d] = ar.get<float>(R * H);
            a.dzg[d] = ar.get<bf16>(R * G); a.hp[d] = ar.get<bf16>(R * m->KP);
        }
        a.y16 = ar.get<bf16>(R * 2 * H); a.ys16 = m->sub[i] > 1 ? ar.get<bf16>(R * 2 * H) : a.y16; a.z32 = ar.get<float>(R * l.N); a.x32 = ar.get<float>(R * l.N); a.x16 = ar.get<bf16>(R * l.N);
    }
    m->logits = ar.get<float>(R * m->C); m->dlogits = ar.get<float>(R * m->C); m->dl16 = ar.get<bf16>(R * m->Cp8);
    m->lens = ar.get<int>((int64_t)B * (m->L + 1)); m->tgt_off = ar.get<int>(B); m->tgt_len = ar.get<int>(B); m->tgt = ar.get<int>((int64_t)B * (maxS / 2 + 2));
    for (int i = 0; i <= m->L; ++i) m->lens_l[i] = m->lens ? m->lens + (int64_t)i * B : nullptr;
    m->nll = ar.get<float>(B); m->ctc_work = ar.get<float>(mk_ctc_work_floats(m->Tp, B, maxS));
    for (int d = 0; d < 2; ++d) { m->h16[d][0] = ar.get<bf16>((int64_t)B * m->KP); m->h16[d][1] = ar.get<bf16>((int64_t)B * m->KP); m->cstate[d] = ar.get<float>((int64_t)B * H); }
    m->rec_words = ar.get<unsigned long long>(mk_lstm_rec_words(B > 32 ? 32 : B, H));
    int64_t maxK = m->F > 2 * H ? m->F : 2 * H;
    m->dx32 = ar.get<float>(R * maxK); m->dy32 = ar.get<float>(R * 2 * H); m->dys32 = ar.get<float>(R * 2 * H); m->wtmp = ar.get<float>((int64_t)G * maxK + G);
    m->dp2 = ar.get<bf16>(R * m->F); m->dc4 = ar.get<bf16>(P2 * 256); m->dc3 = ar.get<bf16>(P2 * 256); m->dp1 = ar.get<bf16>(P2 * 128);
    m->dc2 = ar.get<bf16>(P1 * 128); m->dc1 = ar.get<bf16>(P1 * 128);
    int64_t sl = mk_sumsq_slab_floats(m->nparams);
    auto mx = [&](int64_t v) { if (v > sl) sl = v; };
    mx(mk_conv1_wgrad_slab_floats(B, T, m->D));
    mx(mk_conv3x3_wgrad_slab_floats(B, T, m->D, 128, 128));
    mx(mk_conv3x3_wgrad_slab_floats(B, m->H2, m->W2, 128, 256));
    mx(mk_conv3x3_wgrad_slab_floats(B, m->H2, m->W2, 256, 256));
    m->slab_floats = sl; m->slab = ar.get<float>(sl);
}

GemmArgs nt(const bf16* x, long ldx, const bf16* w16, long ldw, int M, int N, int K, const float* bias) {
    GemmArgs g = gemm_args();
    g.A = x; g.lda = ldx; g.B = w16; g.ldb = ldw; g.M = M; g.N = N; g.K = K; g.bias = bias;
    return g;
}
GemmArgs rm(const bf16* dy, long lddy, const bf16* x, long ldx, int rows, int N, int K, float* dW, long ldc, float* db) {
    GemmArgs g = gemm_args();
    g.reduction_major = 1; g.A = dy; g.lda = lddy; g.B = x; g.ldb = ldx; g.M = N; g.N = K; g.K = rows; g.C32 = dW; g.ldc = ldc; g.colsum = db;
    return g;
}

int forward(masr_blstm* m, const float* xs, hipStream_t s) {
    const float* P = m->P; const int B = m->B, T = m->T, D = m->D, H = m->H, G = 4 * H;
    CK(mk_conv1_fwd_n(xs, P + m->conv[0].w, P + m->conv[0].b, m->c1, B, T, D, 128, s));
    auto conv = [&](const bf16* in, const ConvP& cv, bf16* out, int Hh, int Ww) -> int {
        ConvArgs ca{}; ca.sched = m->conv_sched; ca.in = in; ca.wk = cv.k16; ca.bias = P + cv.b; ca.relu = 1; ca.out = out; ca.B = B; ca.H = Hh; ca.W = Ww; ca.CIN = cv.CI; ca.COUT = cv.CO;
        return mk_conv3x3(ca, s);
    };
    CK(conv(m->c1, m->conv[1], m->c2, T, D));
    CK(mk_maxpool_fwd(m->c2, m->p1, B, T, D, 128, s, 1));
    CK(conv(m->p1, m->conv[2], m->c3, m->H2, m->W2));
    CK(conv(m->c3, m->conv[3], m->c4, m->H2, m->W2));
    CK(mk_maxpool_fwd(m->c4, m->p2, B, m->H2, m->W2, 256, s, 1));
    const bf16* x16 = m->p2; int K = m->F;
    for (int i = 0; i < m->L; ++i) {
        LayerAct& a = m->act[i]; const Layer& l = m->layers[i];
        const int Tin = m->Ts[i], Tout = m->Ts[i + 1], R = B * Tin, Ro = B * Tout;
        for (int d = 0; d < 2; ++d) {
            GemmArgs g = nt(x16, K, l.d[d].wih16, K, R, G, K, l.d[d].bias); g.C32 = a.gx[d]; g.ldc = G;
            CK(mk_gemm(g, s));
        }
        LstmStepArgs st{}; st.B = B; st.T = Tin; st.H = H; st.KP = m->KP; st.lens = m->lens_l[i]; st.y16 = a.y16;
        for (int d = 0; d < 2; ++d) {
            st.h16[d][0] = m->h16[d][0]; st.h16[d][1] = m->h16[d][1]; st.whh16[d] = l.d[d].whh16; st.whhT16[d] = l.d[d].whhT16;
            st.gx[d] = a.gx[d]; st.act[d] = a.act[d]; st.c[d] = a.c[d]; st.cstate[d] = m->cstate[d]; st.dz16[d] = a.dzg[d];
        }
        // one launch for the whole sequence where the shape allows (lstm_rec.hip), else one per timestep
        if (m->resident && mk_lstm_rec_ok(B, H, m->KP)) CK(mk_lstm_fwd_rec(st, m->rec_words, reinterpret_cast<int*>(m->stats + 8), s, m->test_stall));
        else CK(mk_lstm_fwd_steps(st, s));
        if (m->sub[i] > 1) CK(mk_subsample_rows(a.y16, a.ys16, B, Tin, Tout, m->sub[i], 2 * H, s));      // ys_pad[:, ::sub] (encoder.py:118-121)
        GemmArgs g = nt(a.ys16, 2 * H, l.bt16, 2 * H, Ro, l.N, 2 * H, P + l.btb); g.C32 = a.z32; g.ldc = l.N;
        CK(mk_gemm(g, s));
        CK(mk_tanh_fwd(a.z32, a.x32, a.x16, (long)Ro * l.N, s));
        x16 = a.x16; K = l.N;
    }
    LayerAct& la = m->act[m->L - 1];
    const int Tl = m->Ts[m->L], R = B * Tl;
    CK(mk_mask_rows(la.x32, la.x16, m->lens_l[m->L], B, Tl, K, s));            // out.masked_fill(pad, 0) (encoder.py:297-298)
    GemmArgs g = nt(la.x16, K, m->head16, K, R, m->C, K, P + m->headb); g.C32 = m->logits; g.ldc = m->C;
    CK(mk_gemm(g, s));
    return 0;
}

int backward(masr_blstm* m, const float* xs, hipStream_t s) {
    float* Gr = m->G; const int B = m->B, T = m->T, D = m->D, H = m->H, G = 4 * H;
    int K = m->layers.back().N;
    LayerAct& la = m->act[m->L - 1];
    // head
    {
        const int Tl = m->Ts[m->L], R = B * Tl;
        CK(mk_cast_rows_pad(m->dlogits, m->dl16, R, m->C, m->Cp8, s));
        CK(mk_gemm(rm(m->dl16, m->Cp8, la.x16, K, R, m->C, K, Gr + m->headw, K, Gr + m->headb), s));
        { GemmArgs g = nt(m->dl16, m->Cp8, m->headT16, m->Cp8, R, K, m->Cp8, nullptr); g.C32 = m->dx32; g.ldc = K; CK(mk_gemm(g, s)); }
        CK(mk_mask_rows(m->dx32, nullptr, m->lens_l[m->L], B, Tl, K, s));
    }
    for (int i = m->L - 1; i >= 0; --i) {
        LayerAct& a = m->act[i]; const Layer& l = m->layers[i];
        const bf16* xin = i > 0 ? m->act[i - 1].x16 : m->p2;
        const int Kin = l.K;
        const int Tp = m->Ts[i], To = m->Ts[i + 1], R = B * Tp, Ro = B * To;       // frames entering / leaving this layer
        // tanh + projection
        bf16* dz16 = a.x16;                                   // the bf16 copy of this layer's output is dead now: reuse it for d(projection)
        CK(mk_tanh_bwd(m->dx32, a.x32, dz16, (long)Ro * l.N, s));
        CK(mk_gemm(rm(dz16, l.N, a.ys16, 2 * H, Ro, l.N, 2 * H, Gr + l.btw, 2 * H, Gr + l.btb), s));
        { GemmArgs g = nt(dz16, l.N, l.btT16, (l.N + 7) / 8 * 8, Ro, 2 * H, l.N, nullptr); g.C32 = m->sub[i] > 1 ? m->dys32 : m->dy32; g.ldc = 2 * H; CK(mk_gemm(g, s)); }
        if (m->sub[i] > 1) CK(mk_subsample_rows_bwd(m->dys32, m->dy32, B, Tp, To, m->sub[i], 2 * H, s));     // the dropped frames carry no gradient
        // recurrence
        LstmStepArgs st{}; st.B = B; st.T = Tp; st.H = H; st.KP = m->KP; st.lens = m->lens_l[i]; st.y16 = a.y16; st.dy = m->dy32;
        for (int d = 0; d < 2; ++d) {
            st.h16[d][0] = m->h16[d][0]; st.h16[d][1] = m->h16[d][1]; st.whh16[d] = l.d[d].whh16; st.whhT16[d] = l.d[d].whhT16;
            st.gx[d] = a.gx[d]; st.act[d] = a.act[d]; st.c[d] = a.c[d]; st.cstate[d] = m->cstate[d]; st.dz16[d] = a.dzg[d];
        }
        if (m->resident && mk_lstm_rec_ok(B, H, m->KP)) CK(mk_lstm_bwd_rec(st, m->rec_words, reinterpret_cast<int*>(m->stats + 8), s));
        else CK(mk_lstm_bwd_steps(st, s));
        CK(mk_lstm_hprev(a.y16, a.hp[0], a.hp[1], B, Tp, H, m->KP, s));
        const int pc = i == 0 ? 256 : 0, pd = i == 0 ? m->Dp : 0;
        for (int d = 0; d < 2; ++d) {
            // dW_ih, db (unit-major rows in wtmp, then back to torch order); column sums land behind the matrix
            float* dbt = m->wtmp + (int64_t)G * Kin;
            CK(mk_gemm(rm(a.dzg[d], G, xin, Kin, R, G, Kin, m->wtmp, Kin, dbt), s));
            CK(mk_lstm_unperm(m->wtmp, Gr + l.d[d].wih, nullptr, H, Kin, pc, pd, s));
            CK(mk_lstm_unperm(dbt, Gr + l.d[d].bih, Gr + l.d[d].bhh, H, 1, 0, 0, s));
            CK(mk_gemm(rm(a.dzg[d], G, a.hp[d], m->KP, R, G, H, m->wtmp, H, nullptr), s));
            CK(mk_lstm_unperm(m->wtmp, Gr + l.d[d].whh, nullptr, H, H, 0, 0, s));
            GemmArgs g = nt(a.dzg[d], G, l.d[d].wihT16, G, R, Kin, G, nullptr); g.C32 = m->dx32; g.ldc = Kin; g.accumulate = d;
            CK(mk_gemm(g, s));
        }
    }
    // VGG front-end
    CK(mk_cast_bf16(m->dx32, m->dp2, (long)m->rows * m->F, s));
    auto wgrad = [&](const bf16* in, const bf16* dy, const ConvP& cv, int Hh, int Ww) -> int {
        ConvWgradArgs wa{}; wa.in = in; wa.dy = dy; wa.dw = Gr + cv.w; wa.db = Gr + cv.b; wa.slab = m->slab; wa.B = B; wa.H = Hh; wa.W = Ww; wa.CIN = cv.CI; wa.COUT = cv.CO;
        return mk_conv3x3_wgrad(wa, s);
    };
    auto dgrad = [&](const bf16* dy, const ConvP& cv, const bf16* mask, bf16* out, int Hh, int Ww) -> int {
        ConvArgs ca{}; ca.sched = m->conv_sched; ca.in = dy; ca.wk = cv.d16; ca.mask = mask; ca.out = out; ca.B = B; ca.H = Hh; ca.W = Ww; ca.CIN = cv.CO; ca.COUT = cv.CI;
        return mk_conv3x3(ca, s);
    };
    CK(mk_maxpool_relu_bwd(m->c4, m->dp2, m->dc4, B, m->H2, m->W2, 256, s, 1));
    CK(wgrad(m->c3, m->dc4, m->conv[3], m->H2, m->W2));
    CK(dgrad(m->dc4, m->conv[3], m->c3, m->dc3, m->H2, m->W2));
    CK(wgrad(m->p1, m->dc3, m->conv[2], m->H2, m->W2));
    CK(dgrad(m->dc3, m->conv[2], nullptr, m->dp1, m->H2, m->W2));
    CK(mk_maxpool_relu_bwd(m->c2, m->dp1, m->dc2, B, T, D, 128, s, 1));
    CK(wgrad(m->c1, m->dc2, m->conv[1], T, D));
    CK(dgrad(m->dc2, m->conv[1], m->c1, m->dc1, T, D));
    CK(mk_conv1_wgrad_n(xs, m->dc1, Gr + m->conv[0].w, Gr + m->conv[0].b, m->slab, B, T, D, 128, s));
    return 0;
}

}  // namespace

extern "C" {

masr_blstm* masr_blstm_create(const masr_blstm_config* cfg) {
    if (!cfg || cfg->nlayers < 1 || cfg->nlayers > 8 || cfg->enc_dim < 8 || cfg->enc_dim % 8 || cfg->proj_dim % 8 || cfg->enc_odim % 8 || cfg->idim < 4 || cfg->odim < 2) {
        mk_set_error("masr_blstm_create", "bad config (enc_dim / proj_dim / odim of the encoder must be multiples of 8)"); return nullptr;
    }
    masr_blstm* m = new masr_blstm();
    m->cfg = *cfg; m->D = cfg->idim; m->C = cfg->odim; m->H = cfg->enc_dim; m->KP = (cfg->enc_dim + 31) / 32 * 32; m->L = cfg->nlayers;
    m->Cp8 = (cfg->odim + 7) / 8 * 8;
    for (int i = 0; i < m->L; ++i) {
        m->sub[i] = cfg->sample_rate[i] > 0 ? cfg->sample_rate[i] : 1;
        if (m->sub[i] > 8) { mk_set_error("masr_blstm_create", "sample_rate: 1 .. 8 per layer"); delete m; return nullptr; }
    }
    const int idx[4] = {0, 2, 5, 7}; const int co[4] = {128, 128, 256, 256}, ci[4] = {1, 128, 128, 256};
    for (int i = 0; i < 4; ++i) {
        ConvP& c = m->conv[i]; c.CO = co[i]; c.CI = ci[i]; c.k16 = c.d16 = nullptr;
        const std::string pre = "encoder.vgg." + std::to_string(idx[i]);
        c.w = add_param(m, pre + ".weight", {co[i], ci[i], 3, 3});
        c.b = add_param(m, pre + ".bias", {co[i]});
    }
    const int Dp = ((cfg->idim + 1) / 2 + 1) / 2, F = 256 * Dp, H = m->H;
    m->layers.resize(m->L);
    for (int i = 0; i < m->L; ++i) {
        Layer& l = m->layers[i];
        l.K = i == 0 ? F : cfg->proj_dim; l.N = i == m->L - 1 ? cfg->enc_odim : cfg->proj_dim;
        const std::string pre = "encoder.blstm.rnn" + std::to_string(i);
        const char* suf[2] = {"", "_reverse"};
        for (int d = 0; d < 2; ++d) {
            l.d[d].wih = add_param(m, pre + ".weight_ih_l0" + suf[d], {4 * H, l.K});
            l.d[d].whh = add_param(m, pre + ".weight_hh_l0" + suf[d], {4 * H, H});
            l.d[d].bih = add_param(m, pre + ".bias_ih_l0" + suf[d], {4 * H});
            l.d[d].bhh = add_param(m, pre + ".bias_hh_l0" + suf[d], {4 * H});
        }
        const std::string bt = "encoder.blstm.bt" + std::to_string(i);
        l.btw = add_param(m, bt + ".weight", {l.N, 2 * H}); l.btb = add_param(m, bt + ".bias", {l.N});
    }
    m->headw = add_param(m, "head.weight", {m->C, cfg->enc_odim}); m->headb = add_param(m, "head.bias", {m->C});
    m->nparams = (m->nparams + 3) / 4 * 4;
    Arena ar{nullptr, 0, 0};
    plan_persistent(m, ar);
    m->persist_bytes = ar.off;
    return m;
}
void masr_blstm_destroy(masr_blstm* m) {
    if (!m) return;
    if (m->h_stage) hipHostFree(m->h_stage);
    if (m->h_stats) hipHostFree(m->h_stats);
    if (m->stage_ev) hipEventDestroy(m->stage_ev);
    delete m;
}
int64_t masr_blstm_param_numel(const masr_blstm* m) { return m->nparams; }
int masr_blstm_param_count(const masr_blstm* m) { return (int)m->params.size(); }
int masr_blstm_param_info(const masr_blstm* m, int idx, char* name, int cap, int64_t shape[4], int* ndim, int64_t* offset) {
    if (idx < 0 || idx >= (int)m->params.size()) { mk_set_error("masr_blstm_param_info", "index out of range"); return -1; }
    const PInfo& p = m->params[idx];
    if (name && cap > 0) { std::strncpy(name, p.name.c_str(), cap - 1); name[cap - 1] = 0; }
    for (int i = 0; i < 4; ++i) shape[i] = p.shape[i];
    *ndim = p.ndim; *offset = p.off;
    return 0;
}
int64_t masr_blstm_workspace_bytes(const masr_blstm* mc, int B, int T, int max_target_len) {
    masr_blstm tmp = *mc;                                   // plan on a copy: the planner only fills pointers and sizes
    Arena ar{nullptr, 0, 0};
    plan_acts(&tmp, ar, B, T, 2 * (max_target_len + 2) + 1);
    return mc->persist_bytes + ar.off + 4096;
}
int masr_blstm_bind(masr_blstm* m, float* params, float* grads, void* workspace, int64_t ws_bytes) {
    if (!params || !grads || !workspace || ws_bytes < m->persist_bytes) { mk_set_error("masr_blstm_bind", "null pointer or workspace too small"); return -1; }
    if (((uintptr_t)workspace & 255) || ((uintptr_t)params & 15) || ((uintptr_t)grads & 15)) { mk_set_error("masr_blstm_bind", "misaligned buffers"); return -1; }
    m->P = params; m->G = grads; m->ws = (char*)workspace; m->ws_bytes = ws_bytes;
    Arena ar{m->ws, ws_bytes, 0};
    plan_persistent(m, ar);
    if (!m->h_stage) {
        HIP_CHECK_RET(hipHostMalloc((void**)&m->h_stage, sizeof(int) * (1 << 16), hipHostMallocDefault));
        HIP_CHECK_RET(hipHostMalloc((void**)&m->h_stats, sizeof(float) * 64, hipHostMallocDefault));
        HIP_CHECK_RET(hipEventCreateWithFlags(&m->stage_ev, hipEventDisableTiming));
    }
    HIP_CHECK_RET(hipMemset(m->conv_sched, 0, sizeof(unsigned) * 64));
    HIP_CHECK_RET(hipMemset(m->stats, 0, sizeof(float) * 64));
    HIP_CHECK_RET(hipMemset(m->head16, 0, sizeof(bf16) * (size_t)m->Cp8 * m->layers.back().N));
    HIP_CHECK_RET(hipMemset(m->headT16, 0, sizeof(bf16) * (size_t)m->layers.back().N * m->Cp8));
    m->have = false;
    return 0;
}
int masr_blstm_refresh(masr_blstm* m, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!m->P) { mk_set_error("masr_blstm_refresh", "not bound"); return -1; }
    const float* P = m->P; const int H = m->H;
    const int Dp = ((m->D + 1) / 2 + 1) / 2;
    for (int i = 1; i < 4; ++i) CK(mk_conv_weight_shadows(P + m->conv[i].w, m->conv[i].k16, m->conv[i].d16, m->conv[i].CO, m->conv[i].CI, s));
    for (int i = 0; i < m->L; ++i) {
        Layer& l = m->layers[i];
        for (int d = 0; d < 2; ++d)
            CK(mk_lstm_shadows(P + l.d[d].wih, P + l.d[d].whh, P + l.d[d].bih, P + l.d[d].bhh, H, l.K, l.K, m->KP, l.d[d].wih16, l.d[d].wihT16,
                               l.d[d].whh16, l.d[d].whhT16, l.d[d].bias, i == 0 ? 256 : 0, i == 0 ? Dp : 0, s));
        CK(mk_cast_bf16(P + l.btw, l.bt16, (long)l.N * 2 * H, s));
        CK(mk_transpose_cast_bf16(P + l.btw, l.btT16, l.N, 2 * H, (l.N + 7) / 8 * 8, s));
    }
    const int E = m->layers.back().N;
    CK(mk_cast_bf16(P + m->headw, m->head16, (long)m->C * E, s));
    CK(mk_transpose_cast_bf16(P + m->headw, m->headT16, m->C, E, m->Cp8, s));
    return 0;
}
int masr_blstm_run_batch(masr_blstm* m, const float* xs, const int64_t* ilens, const int64_t* ys_flat, const int64_t* olens, int B, int T,
                         int flags, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!m->P) { mk_set_error("masr_blstm_run_batch", "not bound"); return -1; }
    if (B <= 0 || T <= 0) { mk_set_error("masr_blstm_run_batch", "empty batch"); return -1; }
    int maxL = 0;
    for (int b = 0; b < B; ++b) if ((int)olens[b] > maxL) maxL = (int)olens[b];
    const int maxS = 2 * (maxL + 2) + 1;
    Arena ar{m->ws, m->ws_bytes, m->persist_bytes};
    plan_acts(m, ar, B, T, maxS);
    if (ar.off > m->ws_bytes) { mk_set_error("masr_blstm_run_batch", "workspace too small (masr_blstm_workspace_bytes)"); return -2; }
    if ((int64_t)B * (maxL + 5) + 3 * B + (int64_t)B * (m->L + 1) > (1 << 16)) { mk_set_error("masr_blstm_run_batch", "staging buffer too small"); return -1; }
    HIP_CHECK_RET(hipEventSynchronize(m->stage_ev));
    // targets [sos] + y + [eos] with sos = eos = odim - 1 (blstm_trainer.py:56-59); enc_lens = ceil(ceil(ilens/2)/2)
    int* h = m->h_stage; int* h_len = h; int* h_off = h + B; int* h_tl = h + 2 * B; int* h_t = h + 3 * B;
    const int eos = m->C - 1;
    int64_t src = 0; int dst = 0;
    for (int b = 0; b < B; ++b) {
        if (ilens[b] < 1 || ilens[b] > T) { mk_set_error("masr_blstm_run_batch", "ilens must be in [1, T]"); return -1; }
        h_len[b] = (int)(((ilens[b] + 1) / 2 + 1) / 2);
        h_off[b] = dst; h_tl[b] = (int)olens[b] + 2;
        h_t[dst++] = eos;
        for (int l = 0; l < (int)olens[b]; ++l) {
            const int tok = (int)ys_flat[src + l];
            if (tok < 0 || tok >= m->C) { mk_set_error("masr_blstm_run_batch", "label out of range"); return -1; }
            h_t[dst++] = tok;
        }
        h_t[dst++] = eos;
        src += olens[b];
    }
    int* h_ll = h_t + dst;                                      // enc_lens entering every layer + leaving the last: (len + 1) / sub (encoder.py:121)
    for (int b = 0; b < B; ++b) h_ll[b] = h_len[b];
    for (int i = 0; i < m->L; ++i) for (int b = 0; b < B; ++b) h_ll[(i + 1) * B + b] = m->sub[i] > 1 ? (h_ll[i * B + b] + 1) / m->sub[i] : h_ll[i * B + b];
    HIP_CHECK_RET(hipMemcpyAsync(m->lens, h_ll, sizeof(int) * B * (m->L + 1), hipMemcpyHostToDevice, s));
    HIP_CHECK_RET(hipMemcpyAsync(m->tgt_off, h_off, sizeof(int) * B, hipMemcpyHostToDevice, s));
    HIP_CHECK_RET(hipMemcpyAsync(m->tgt_len, h_tl, sizeof(int) * B, hipMemcpyHostToDevice, s));
    HIP_CHECK_RET(hipMemcpyAsync(m->tgt, h_t, sizeof(int) * dst, hipMemcpyHostToDevice, s));
    HIP_CHECK_RET(hipEventRecord(m->stage_ev, s));
    CK(forward(m, xs, s));
    CK(mk_ctc_loss(m->logits, m->tgt, m->tgt_off, m->lens_l[m->L], m->tgt_len, m->Ts[m->L], B, m->C, 0, m->nll, m->stats, m->dlogits, m->ctc_work, maxS, s, 1));
    m->have = true;
    if (flags & MASR_TRAIN) CK(backward(m, xs, s));
    return 0;
}
int masr_blstm_forward(masr_blstm* m, const float* xs, const int64_t* ilens, int B, int T, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!m->P) { mk_set_error("masr_blstm_forward", "not bound"); return -1; }
    if (B <= 0 || T <= 0 || B > 4096) { mk_set_error("masr_blstm_forward", "bad batch"); return -1; }
    Arena ar{m->ws, m->ws_bytes, m->persist_bytes};
    plan_acts(m, ar, B, T, 2 * (1 + 2) + 1);
    if (ar.off > m->ws_bytes) { mk_set_error("masr_blstm_forward", "workspace too small (masr_blstm_workspace_bytes)"); return -2; }
    HIP_CHECK_RET(hipEventSynchronize(m->stage_ev));
    for (int b = 0; b < B; ++b) {
        if (ilens[b] < 1 || ilens[b] > T) { mk_set_error("masr_blstm_forward", "ilens must be in [1, T]"); return -1; }
        m->h_stage[b] = (int)(((ilens[b] + 1) / 2 + 1) / 2);
    }
    for (int i = 0; i < m->L; ++i) for (int b = 0; b < B; ++b) m->h_stage[(i + 1) * B + b] = m->sub[i] > 1 ? (m->h_stage[i * B + b] + 1) / m->sub[i] : m->h_stage[i * B + b];
    HIP_CHECK_RET(hipMemcpyAsync(m->lens, m->h_stage, sizeof(int) * B * (m->L + 1), hipMemcpyHostToDevice, s));
    HIP_CHECK_RET(hipEventRecord(m->stage_ev, s));
    CK(forward(m, xs, s));
    m->have = true;
    return 0;
}
int masr_blstm_read_stats(masr_blstm* m, float out[4], void* stream) {
    hipStream_t s = (hipStream_t)stream;
    HIP_CHECK_RET(hipMemcpyAsync(m->h_stats, m->stats, sizeof(float) * 12, hipMemcpyDeviceToHost, s));
    HIP_CHECK_RET(hipStreamSynchronize(s));
    for (int i = 0; i < 4; ++i) out[i] = m->h_stats[i];
    if (reinterpret_cast<const int*>(m->h_stats)[8] != 0) {       // a workgroup of the resident recurrence gave up waiting for its peers (lstm_rec.hip)
        HIP_CHECK_RET(hipMemsetAsync(m->stats + 8, 0, sizeof(int), s));
        mk_set_error("masr_blstm_read_stats", "the resident LSTM recurrence timed out waiting for a peer workgroup: the step's results are invalid");
        return -1;
    }
    return 0;
}
void masr_blstm_set_resident_recurrence(masr_blstm* m, int on) { m->resident = on != 0; }
void masr_test_blstm_stall(masr_blstm* m, int on) { m->test_stall = on; }
// forward-only callers (masr_blstm_forward + masr_blstm_last_logits: the Tester) never read the stats block: this is their check.  Synchronises
// the stream; -1 (and the mark cleared) when the resident recurrence of a launch since the last check timed out -- the logits are then invalid.
int masr_blstm_check(masr_blstm* m, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    HIP_CHECK_RET(hipMemcpyAsync(m->h_stats + 8, m->stats + 8, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_CHECK_RET(hipStreamSynchronize(s));
    if (reinterpret_cast<const int*>(m->h_stats)[8] != 0) {
        HIP_CHECK_RET(hipMemsetAsync(m->stats + 8, 0, sizeof(int), s));
        mk_set_error("masr_blstm_check", "the resident LSTM recurrence timed out waiting for a peer workgroup: the forward's results are invalid");
        return -1;
    }
    return 0;
}
int masr_blstm_last_logits(masr_blstm* m, float** logits, int32_t** enc_lens, int* B, int* Tp, int* C) {
    if (!m->have) { mk_set_error("masr_blstm_last_logits", "run a batch first"); return -1; }
    *logits = m->logits; *enc_lens = m->lens_l[m->L]; *B = m->B; *Tp = m->Ts[m->L]; *C = m->C;      // (frames / lengths LEAVING the encoder)
    return 0;
}
int masr_blstm_clip_grads(masr_blstm* m, float max_norm, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!m->have) { mk_set_error("masr_blstm_clip_grads", "run a batch first"); return -1; }
    CK(mk_sumsq(m->G, m->nparams, m->slab, m->stats + 3, s));
    return mk_clip_scale(m->G, m->nparams, m->stats + 3, max_norm, s);
}
int masr_blstm_clip_sgd_step(masr_blstm* m, float* mom, float max_norm, float lr, float momentum, int nesterov, int first_step, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!m->have) { mk_set_error("masr_blstm_clip_sgd_step", "run a batch first"); return -1; }
    CK(mk_sumsq(m->G, m->nparams, m->slab, m->stats + 3, s));
    CK(mk_clip_sgd(m->P, m->G, mom, m->nparams, m->stats + 3, max_norm, lr, momentum, nesterov, first_step, s));
    return masr_blstm_refresh(m, stream);
}

}  // extern "C"
