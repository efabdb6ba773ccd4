// CTC forward-backward lattice (config 1 / BLSTM path only; reference call site src/blstm_trainer.py:22,62-70:
// nn.CTCLoss(blank=0, reduction='mean', zero_infinity=True) on log_softmax(pred)).
// The alpha and the beta recursion of an utterance are wavefront scans over the S = 2L+1 lattice: rows double-buffered in LDS, one
// barrier per frame, run side by side by two workgroups; their rows are parked in HBM scratch and a chip-filling grid (one workgroup
// per frame) forms the posteriors and d loss / d logits = (softmax - posterior) / (L * B).
#include "common.h"
#include "kernels.h"

namespace {
constexpr int MAXS = 2048;
constexpr float NINF = -INFINITY;
// lengths arrive in DEVICE arrays, so the launch cannot vet them on the host: an utterance whose lengths do not fit the call
// (in_len < 0 or > T, tgt_len < 0, 2 tgt_len + 1 > the lattice width the work buffer was sized for) is not run -- its nll becomes
// NaN (so does the mean: loud), its gradient rows stay zero, and its index + 1 is left in the CALL's own work buffer (ctc_mark: the last
// word of mk_ctc_work_floats, zeroed by the launch itself) for mk_ctc_status -- no process-wide state, concurrent calls do not mix
__device__ __forceinline__ int* ctc_mark(float* work, int T, int B, int Spad) { return reinterpret_cast<int*>(work + 2L * B * T * Spad + 2L * B * T + B + 3); }

__device__ __forceinline__ float lae(float a, float b) {          // log(exp a + exp b)
    if (a == NINF) return b;
    if (b == NINF) return a;
    // (hardware exp / log: the argument of the log lies in (1, 2], where its absolute error is ~1e-7 -- below one ulp of the O(10 .. 100)
    // log-probabilities it is added to; log1pf + a libm exp made this function two thirds of a frame's dependent chain)
    const float m = fmaxf(a, b);
    return m + __logf(1.f + __expf(fminf(a, b) - m));
}

// Round 4: the lattice as THREE grids instead of one workgroup per utterance doing everything (468 us for B = 8, T' = 100 on 8 of the
// 256 CUs, 13 barriers per frame in the beta sweep):
//   ctc_lse_kernel     grid (T / 4, B): the log-softmax normaliser of every frame, one wave each
//   ctc_sweep_kernel   grid (B, 2): the alpha sweep and the beta sweep of an utterance run CONCURRENTLY in two workgroups (they are
//                      independent recursions); rows double-buffered in LDS, ONE barrier per frame, every row parked in HBM scratch;
//                      the alpha workgroup ends with the log-likelihood -> nll[b] (zero_infinity: 0) and ll[b]
//   ctc_grad_kernel    grid (T, B): posteriors and d loss / d logits of ONE frame per workgroup from the parked rows -- embarrassingly
//                      parallel, so this part fills the chip; class posteriors summed by owner threads in a fixed order as before
//                      (bit-reproducible), the blank states by wave sums folded in wave order
// Lengths are vetted by both (see ctc_mark).
struct CtcGeo { int T, B, C, blank, Spad; long st_t, st_b; };
__device__ __forceinline__ long ctc_at(const CtcGeo& g, int t, int b) { return ((long)t * g.st_t + (long)b * g.st_b) * g.C; }

// log-softmax normaliser of every frame: one wave per frame, grid (ceil(T / 4), B)
__global__ __launch_bounds__(256) void ctc_lse_kernel(const float* __restrict__ logits, const int* __restrict__ in_len, CtcGeo g, float* __restrict__ work) {
    const int b = blockIdx.y, lane = threadIdx.x & 63, t = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int Tb = in_len[b];
    if (blockIdx.x == 0 && b == 0 && threadIdx.x == 0) *ctc_mark(work, g.T, g.B, g.Spad) = 0;      // (the sweeps run behind this launch)
    if (t >= g.T || t >= Tb) return;                       // (also bad lengths: Tb < 0 or > T are handled by the sweeps; rows past T do not exist)
    const float* z = logits + ctc_at(g, t, b);
    float v[8];                                            // C <= 512 in registers (larger vocabularies re-read)
    float mx = -3.4e38f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int c = lane + 64 * k; v[k] = c < g.C ? z[c] : -3.4e38f; mx = fmaxf(mx, v[k]); }
    for (int c = lane + 512; c < g.C; c += 64) mx = fmaxf(mx, z[c]);
    mx = wave_max(mx);
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) se += (lane + 64 * k < g.C) ? __expf(v[k] - mx) : 0.f;
    for (int c = lane + 512; c < g.C; c += 64) se += __expf(z[c] - mx);
    se = wave_sum(se);
    if (lane == 0) work[2L * g.B * g.T * g.Spad + (long)b * g.T + t] = mx + __logf(se);
}

__global__ __launch_bounds__(256) void ctc_sweep_kernel(const float* __restrict__ logits, const int* __restrict__ targets,
                                                        const int* __restrict__ tgt_off, const int* __restrict__ in_len,
                                                        const int* __restrict__ tgt_len, CtcGeo g, float* __restrict__ nll, float* __restrict__ work) {
    const int b = blockIdx.x, beta = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = g.T, B = g.B, C = g.C, blank = g.blank, Spad = g.Spad;
    const int Tb = in_len[b], Lb = tgt_len[b], S = 2 * Lb + 1;
    const bool bad = Tb < 0 || Tb > T || Lb < 0 || S > Spad;
    float* rows = work + ((long)beta * B + b) * T * Spad;                       // alpha | beta rows [T][Spad] of this utterance
    const float* wlse = work + 2L * B * T * Spad + (long)b * T;                 // log-softmax normalisers (ctc_lse_kernel)
    float* ll_out = work + 2L * B * T * Spad + 2L * B * T;                      // [B]
    if (bad || Tb == 0) {
        // no frames: torch's lattice has no path unless the target is empty too (nll 0); with a target the likelihood is 0, i.e. nll =
        // inf, which zero_infinity turns into 0 with a zero gradient -- 0 either way.  Bad lengths: NaN, marked.
        if (!beta && tid == 0) { nll[b] = bad ? __int_as_float(0x7fc00000) : 0.f; ll_out[b] = NINF; if (bad) atomicMax(ctc_mark(work, T, B, Spad), b + 1); }
        return;
    }
    const int* tg = targets + tgt_off[b];
    __shared__ float row[2][MAXS];
    (void)lane; (void)wave; (void)C;
    auto ext = [&](int s) { return (s & 1) ? tg[s >> 1] : blank; };
    auto lp = [&](int t, int s) { return logits[ctc_at(g, t, b) + ext(s)] - wlse[t]; };
    if (S <= 256) {
        // the speech-sized lattice (L <= 127): ONE state per thread.  What a frame costs is a chain of dependent operations -- neighbour
        // rows from LDS, two log-add-exps, the emission, the barrier -- and the emission log-prob is a GLOBAL load (~1 us when it sits in
        // that chain: 162 us for 100 frames): it does not depend on the recursion, so the next PF frames' values are requested ahead and
        // ride in registers
        constexpr int PF = 6;
        const int s = tid < S ? tid : S - 1;                                      // (threads past S repeat the last state: uniform control flow)
        const int e = ext(s);
        const bool skip = (s & 1) && (beta ? (s + 2 < S && ext(s + 2) != e) : (s > 1 && ext(s - 2) != e));
        const int dir = beta ? -1 : 1, t0 = beta ? Tb - 1 : 0;
        float ring[PF];
#pragma unroll
        for (int k = 0; k < PF; ++k) { const int t = t0 + dir * k; ring[k] = (t >= 0 && t < Tb) ? lp(t, s) : 0.f; }
        {
            const float v = beta ? ((s >= S - 2) ? ring[0] : NINF) : ((s < 2) ? ring[0] : NINF);
            if (tid < S) { row[t0 & 1][s] = v; rows[(long)t0 * Spad + s] = v; }
        }
        __syncthreads();
        for (int i = 1; i < Tb; i += PF) {
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                const int step = i + k;
                if (step >= Tb) break;                                              // (uniform)
                const int t = t0 + dir * step, tn = t0 + dir * (step + PF - 1);
                const float emis = ring[(k + 1) % PF];                            // requested PF - 1 frames ago
                ring[k] = (tn >= 0 && tn < Tb) ? lp(tn, s) : 0.f;                 // slot k held frame step - 1's value: free now
                const float* prv = row[(t - dir) & 1];
                float a = prv[s];
                if (beta) { if (s + 1 < S) a = lae(a, prv[s + 1]); if (skip) a = lae(a, prv[s + 2]); }
                else { if (s > 0) a = lae(a, prv[s - 1]); if (skip) a = lae(a, prv[s - 2]); }
                a = (a == NINF) ? NINF : a + emis;
                if (tid < S) { row[t & 1][s] = a; rows[(long)t * Spad + s] = a; }
                __syncthreads();
            }
        }
        if (!beta && tid == 0) {
            const float* last = row[(Tb - 1) & 1];
            float ll = last[S - 1];
            if (S > 1) ll = lae(ll, last[S - 2]);
            const bool inf = ll == NINF || ll != ll;                               // zero_infinity
            nll[b] = inf ? 0.f : -ll;
            ll_out[b] = inf ? NINF : ll;
        }
        return;
    }
    if (!beta) {
        for (int s = tid; s < S; s += 256) { const float a = (s < 2) ? lp(0, s) : NINF; row[0][s] = a; rows[s] = a; }
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
                rows[(long)t * Spad + s] = a;
            }
            __syncthreads();
        }
        if (tid == 0) {
            const float* last = row[(Tb - 1) & 1];
            float ll = last[S - 1];
            if (S > 1) ll = lae(ll, last[S - 2]);
            const bool inf = ll == NINF || ll != ll;                           // zero_infinity
            nll[b] = inf ? 0.f : -ll;
            ll_out[b] = inf ? NINF : ll;
        }
    } else {
        for (int s = tid; s < S; s += 256) { const float v = (s >= S - 2) ? lp(Tb - 1, s) : NINF; row[(Tb - 1) & 1][s] = v; rows[(long)(Tb - 1) * Spad + s] = v; }
        __syncthreads();
        for (int t = Tb - 2; t >= 0; --t) {
            const float* nxt = row[(t + 1) & 1];
            float* cur = row[t & 1];
            for (int s = tid; s < S; s += 256) {
                float bta = nxt[s];
                if (s + 1 < S) bta = lae(bta, nxt[s + 1]);
                if (s + 2 < S && (s & 1) && ext(s + 2) != ext(s)) bta = lae(bta, nxt[s + 2]);
                bta = (bta == NINF) ? NINF : bta + lp(t, s);
                cur[s] = bta;
                rows[(long)t * Spad + s] = bta;
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(256) void ctc_grad_kernel(const float* __restrict__ logits, const int* __restrict__ targets,
                                                       const int* __restrict__ tgt_off, const int* __restrict__ in_len,
                                                       const int* __restrict__ tgt_len, CtcGeo g, float* __restrict__ grad,
                                                       const float* __restrict__ work) {
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = g.T, B = g.B, C = g.C, blank = g.blank, Spad = g.Spad;
    const int Tb = in_len[b], Lb = tgt_len[b], S = 2 * Lb + 1;
    float* gr = grad + ctc_at(g, t, b);
    const float ll = work[2L * B * T * Spad + 2L * B * T + b];
    const bool bad = Tb < 0 || Tb > T || Lb < 0 || S > Spad;
    if (bad || t >= Tb || ll == NINF || ll != ll) {                            // padded frame, refused or infeasible utterance: zero row
        for (int c = tid; c < C; c += 256) gr[c] = 0.f;
        return;
    }
    const int* tg = targets + tgt_off[b];
    const float* al = work + (long)b * T * Spad + (long)t * Spad;
    const float* be = work + ((long)B + b) * T * Spad + (long)t * Spad;
    const float lse = work[2L * B * T * Spad + (long)b * T + t];
    const float* z = logits + ctc_at(g, t, b);
    __shared__ float post[MAXS];
    __shared__ float acc[4096];
    __shared__ float red[4];
    for (int c = tid; c < C; c += 256) acc[c] = 0.f;
    float bsum = 0.f;
    for (int s = tid; s < S; s += 256) {
        const float a = al[s], bt = be[s];
        const float lps = z[(s & 1) ? tg[s >> 1] : blank] - lse;
        const float p = (a != NINF && bt != NINF) ? __expf(a + bt - lps - ll) : 0.f;
        post[s] = p;
    }
    __syncthreads();
    // blank states (even s): per-thread strided partials in a fixed order, wave sums, waves folded in order
    for (int s = 2 * tid; s < S; s += 512) bsum += post[s];
    bsum = wave_sum(bsum);
    if (lane == 0) red[wave] = bsum;
    // label classes: the first occurrence of a label sums the chain of its repeats, in target order (one owner per class: no atomics)
    for (int i = tid; i < Lb; i += 256) {
        const int v = tg[i];
        bool first = true;
        for (int j = 0; j < i; ++j) if (tg[j] == v) { first = false; break; }
        if (!first) continue;
        float sum = post[2 * i + 1];
        for (int j = i + 1; j < Lb; ++j) if (tg[j] == v) sum += post[2 * j + 1];
        acc[v] = sum;
    }
    __syncthreads();
    if (tid == 0) acc[blank] = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    const float gscale = 1.f / ((float)(Lb > 0 ? Lb : 1) * (float)B);
    for (int c = tid; c < C; c += 256) gr[c] = (__expf(z[c] - lse) - acc[c]) * gscale;
}
__global__ void ctc_mean_kernel(const float* __restrict__ nll, const int* __restrict__ tgt_len, int B, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += nll[b] / (float)(tgt_len[b] > 0 ? tgt_len[b] : 1);
        out[0] = s / B;
    }
}
}  // namespace

int mk_ctc_status(const float* work, int T, int B, int maxS, hipStream_t s) {
    // > 0: (index + 1) of the last utterance the CTC launch that used `work` refused (see ctc_mark); synchronises `s`
    if (!work || T <= 0 || B <= 0 || maxS < 1) { mk_set_error("mk_ctc_status", "bad arguments"); return -1; }
    int h = 0;
    const float* mark = work + 2L * B * T * ((maxS + 3) / 4 * 4) + 2L * B * T + B + 3;
    if (hipMemcpyAsync(&h, mark, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
    if (hipStreamSynchronize(s) != hipSuccess) return -1;
    if (h > 0) {
        char msg[160];
        snprintf(msg, sizeof msg, "utterance %d: in_len outside [0, T] or target longer than the lattice the work buffer holds (its nll is NaN)", h - 1);
        mk_set_error("mk_ctc_loss", msg);
    }
    return h;
}
// alpha rows + beta rows [2][B][T][Spad], the log-softmax normalisers [B][T] (+ spare), the log-likelihoods [B]
long mk_ctc_work_floats(int T, int B, int maxS) { return 2L * B * T * ((maxS + 3) / 4 * 4) + 2L * B * T + B + 4; }
int mk_ctc_loss(const float* logits, const int* targets, const int* tgt_off, const int* in_len, const int* tgt_len, int T,
                  int B, int C, int blank, float* nll, float* loss_out, float* grad, float* work, int maxS, hipStream_t s, int batch_first) {
    if (maxS > MAXS || C > 4096) { mk_set_error("mk_ctc_loss", "lattice wider than 2048 states or > 4096 classes"); return -1; }
    if (T <= 0 || B <= 0 || C <= 0 || maxS < 1 || blank < 0 || blank >= C) { mk_set_error("mk_ctc_loss", "T, B, C, maxS must be positive and 0 <= blank < C"); return -1; }
    const CtcGeo g{T, B, C, blank, (maxS + 3) / 4 * 4, batch_first ? 1L : (long)B, batch_first ? (long)T : 1L};
    hipLaunchKernelGGL(ctc_lse_kernel, dim3((T + 3) / 4, B), dim3(256), 0, s, logits, in_len, g, work);
    hipLaunchKernelGGL(ctc_sweep_kernel, dim3(B, 2), dim3(256), 0, s, logits, targets, tgt_off, in_len, tgt_len, g, nll, work);
    hipLaunchKernelGGL(ctc_grad_kernel, dim3(T, B), dim3(256), 0, s, logits, targets, tgt_off, in_len, tgt_len, g, grad, work);
    hipLaunchKernelGGL(ctc_mean_kernel, dim3(1), dim3(64), 0, s, nll, tgt_len, B, loss_out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
