// CTC forward-backward lattice (config 1 / BLSTM path only; reference call site src/blstm_trainer.py:22,62-70:
// nn.CTCLoss(blank=0, reduction='mean', zero_infinity=True) on log_softmax(pred)).
// One workgroup per utterance; the alpha/beta rows of the S = 2L+1 lattice live in LDS (double buffered)
// and are advanced one frame per barrier ("wavefront scan"); alpha is parked in HBM scratch so the beta sweep can
// form the posteriors and write d loss / d logits = (softmax - posterior) / (L * B) in the same pass.
#include "common.h"
#include "kernels.h"

namespace {
constexpr int MAXS = 2048;
constexpr float NINF = -INFINITY;
// lengths arrive in DEVICE arrays, so the launch cannot vet them on the host: an utterance whose lengths do not fit the call
// (in_len < 0 or > T, tgt_len < 0, 2 tgt_len + 1 > the lattice width the work buffer was sized for) is not run -- its nll becomes
// NaN (so does the mean: loud), its gradient rows stay zero, and its index + 1 is left here for mk_ctc_status
__device__ int g_ctc_bad = 0;

__device__ __forceinline__ float lae(float a, float b) {          // log(exp a + exp b)
    if (a == NINF) return b;
    if (b == NINF) return a;
    const float m = fmaxf(a, b);
    return m + log1pf(__expf(fminf(a, b) - m));
}

__global__ __launch_bounds__(256) void ctc_kernel(const float* __restrict__ logits, const int* __restrict__ targets,
                                                  const int* __restrict__ tgt_off, const int* __restrict__ in_len,
                                                  const int* __restrict__ tgt_len, int T, int B, int C, int blank,
                                                  float* __restrict__ nll, float* __restrict__ grad,
                                                  float* __restrict__ work, int Spad, long st_t, long st_b) {
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Tb = in_len[b], Lb = tgt_len[b], S = 2 * Lb + 1;
    const bool bad = Tb < 0 || Tb > T || Lb < 0 || S > Spad;
    const int* tg = targets + tgt_off[b];
    float* walpha = work + (long)b * T * Spad;                    // [T][Spad]
    float* wlse = work + (long)B * T * Spad + (long)b * T;        // [T]
    __shared__ float row[2][MAXS];
    __shared__ float acc[4096];
    __shared__ float post[MAXS];                                  // state posteriors of the current frame
    __shared__ float red[256];
    __shared__ short nxt_same[MAXS / 2];                          // label i -> next position with the same label (or -1)
    __shared__ unsigned char is_first[MAXS / 2];                  // label i is the first occurrence of its class
    __shared__ float s_ll;

    // zero the gradient of this utterance (frames >= Tb stay zero)
    // element (t, b, c) of logits / grad sits at ((t * st_t + b * st_b) * C + c): time-major [T][B][C] or batch-first [B][T][C]
    for (long i = tid; i < (long)T * C; i += 256) grad[((i / C) * st_t + b * st_b) * C + (i % C)] = 0.f;
    if (bad || Tb == 0) {
        // no frames: torch's lattice has no path unless the target is empty too (nll 0); with a target the likelihood is 0,
        // i.e. nll = inf, which zero_infinity turns into 0 with a zero gradient -- 0 either way
        if (tid == 0) { nll[b] = bad ? __int_as_float(0x7fc00000) : 0.f; if (bad) atomicMax(&g_ctc_bad, b + 1); }
        return;
    }
    // log-softmax normalisers
    for (int t = wave; t < Tb; t += 4) {
        const float* z = logits + ((long)t * st_t + b * st_b) * C;
        float mx = -3.4e38f;
        for (int c = lane; c < C; c += 64) mx = fmaxf(mx, z[c]);
        mx = wave_max(mx);
        float se = 0.f;
        for (int c = lane; c < C; c += 64) se += __expf(z[c] - mx);
        se = wave_sum(se);
        if (lane == 0) wlse[t] = mx + __logf(se);
    }
    // class ownership for the posterior sums: every class is summed by ONE thread in a fixed order (the first occurrence of
    // a label walks the chain of its repeats; blanks are folded by a fixed tree), so the gradient is bit-reproducible --
    // LDS float atomics made it depend on the order in which the lanes arrived
    for (int i = tid; i < Lb; i += 256) {
        const int v = tg[i];
        int nx = -1;
        for (int j = i + 1; j < Lb; ++j) if (tg[j] == v) { nx = j; break; }
        bool first = true;
        for (int j = 0; j < i; ++j) if (tg[j] == v) { first = false; break; }
        nxt_same[i] = (short)nx; is_first[i] = first ? 1 : 0;
    }
    __syncthreads();
    auto ext = [&](int s) { return (s & 1) ? tg[s >> 1] : blank; };
    auto lp = [&](int t, int s) { return logits[((long)t * st_t + b * st_b) * C + ext(s)] - wlse[t]; };

    // alpha sweep
    for (int s = tid; s < S; s += 256) {
        const float a = (s < 2) ? lp(0, s) : NINF;
        row[0][s] = a;
        walpha[s] = a;
    }
    __syncthreads();
    for (int t = 1; t < Tb; ++t) {
        const float* prev = row[(t - 1) & 1];
        float* cur = row[t & 1];
        for (int s = tid; s < S; s += 256) {
            float a = prev[s];
            if (s > 0) a = lae(a, prev[s - 1]);
            if (s > 1 && (s & 1) && ext(s) != ext(s - 2)) a = lae(a, prev[s - 2]);
            a = (a == NINF) ? NINF : a + lp(t, s);
            cur[s] = a;
            walpha[(long)t * Spad + s] = a;
        }
        __syncthreads();
    }
    if (tid == 0) {
        const float* last = row[(Tb - 1) & 1];
        float ll = last[S - 1];
        if (S > 1) ll = lae(ll, last[S - 2]);
        s_ll = ll;
    }
    __syncthreads();
    const float ll = s_ll;
    if (ll == NINF || ll != ll) {                                   // zero_infinity
        if (tid == 0) nll[b] = 0.f;
        return;
    }
    if (tid == 0) nll[b] = -ll;
    const float gscale = 1.f / ((float)(Lb > 0 ? Lb : 1) * (float)B);

    // beta sweep + gradient
    for (int t = Tb - 1; t >= 0; --t) {
        float* cur = row[t & 1];
        const float* nxt = row[(t + 1) & 1];
        for (int c = tid; c < C; c += 256) acc[c] = 0.f;
        __syncthreads();
        for (int s = tid; s < S; s += 256) {
            float bta;
            if (t == Tb - 1) bta = (s >= S - 2) ? lp(t, s) : NINF;
            else {
                bta = nxt[s];
                if (s + 1 < S) bta = lae(bta, nxt[s + 1]);
                if (s + 2 < S && (s & 1) && ext(s + 2) != ext(s)) bta = lae(bta, nxt[s + 2]);
                bta = (bta == NINF) ? NINF : bta + lp(t, s);
            }
            cur[s] = bta;
            const float al = walpha[(long)t * Spad + s];
            post[s] = (al != NINF && bta != NINF) ? __expf(al + bta - lp(t, s) - ll) : 0.f;
        }
        __syncthreads();
        float bsum = 0.f;                                         // blank states: even s, strided partials + fixed tree
        for (int s = 2 * tid; s < S; s += 512) bsum += post[s];
        red[tid] = bsum;
        for (int i = tid; i < Lb; i += 256) {                     // label classes: the first occurrence sums its chain
            if (!is_first[i]) continue;
            float v = post[2 * i + 1];
            for (int j = nxt_same[i]; j >= 0; j = nxt_same[j]) v += post[2 * j + 1];
            acc[tg[i]] = v;
        }
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) red[tid] += red[tid + o];
            __syncthreads();
        }
        if (tid == 0) acc[blank] = red[0];
        __syncthreads();
        const float* z = logits + ((long)t * st_t + b * st_b) * C;
        float* g = grad + ((long)t * st_t + b * st_b) * C;
        const float lse = wlse[t];
        for (int c = tid; c < C; c += 256) g[c] = (__expf(z[c] - lse) - acc[c]) * gscale;
        __syncthreads();
    }
}
__global__ void ctc_mean_kernel(const float* __restrict__ nll, const int* __restrict__ tgt_len, int B, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += nll[b] / (float)(tgt_len[b] > 0 ? tgt_len[b] : 1);
        out[0] = s / B;
    }
}
}  // namespace

int mk_ctc_status(hipStream_t s) {
    // > 0: (index + 1) of the last utterance a CTC launch on this stream refused (see g_ctc_bad); reads and clears, synchronises `s`
    int h = 0;
    const int zero = 0;
    if (hipMemcpyFromSymbolAsync(&h, HIP_SYMBOL(g_ctc_bad), sizeof(int), 0, hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
    if (hipStreamSynchronize(s) != hipSuccess) return -1;
    if (h > 0) {
        hipMemcpyToSymbolAsync(HIP_SYMBOL(g_ctc_bad), &zero, sizeof(int), 0, hipMemcpyHostToDevice, s);
        hipStreamSynchronize(s);
        char msg[160];
        snprintf(msg, sizeof msg, "utterance %d: in_len outside [0, T] or target longer than the lattice the work buffer holds (its nll is NaN)", h - 1);
        mk_set_error("mk_ctc_loss", msg);
    }
    return h;
}
long mk_ctc_work_floats(int T, int B, int maxS) { return (long)B * T * ((maxS + 3) / 4 * 4) + (long)B * T; }
int mk_ctc_loss(const float* logits, const int* targets, const int* tgt_off, const int* in_len, const int* tgt_len, int T,
                  int B, int C, int blank, float* nll, float* loss_out, float* grad, float* work, int maxS, hipStream_t s, int batch_first) {
    if (maxS > MAXS || C > 4096) { mk_set_error("mk_ctc_loss", "lattice wider than 2048 states or > 4096 classes"); return -1; }
    if (T <= 0 || B <= 0 || C <= 0 || maxS < 1 || blank < 0 || blank >= C) { mk_set_error("mk_ctc_loss", "T, B, C, maxS must be positive and 0 <= blank < C"); return -1; }
    hipLaunchKernelGGL(ctc_kernel, dim3(B), dim3(256), 0, s, logits, targets, tgt_off, in_len, tgt_len, T, B, C, blank, nll, grad,
                       work, (maxS + 3) / 4 * 4, batch_first ? 1L : (long)B, batch_first ? (long)T : 1L);
    hipLaunchKernelGGL(ctc_mean_kernel, dim3(1), dim3(64), 0, s, nll, tgt_len, B, loss_out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
