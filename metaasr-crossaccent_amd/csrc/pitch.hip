// Kaldi pitch features on the GPU: the last 3 of the 83 dims the shipped configs train on (config/transformer/pretrain/
// fometa-hkust.yaml:13 `idim: 83`, README.md:24 = 80 fbank + 3 pitch).  The reference has no extraction code;
// the rows come from ESPnet's make_fbank_pitch.sh = compute-kaldi-pitch-feats | process-kaldi-pitch-feats pasted behind the fbank
// rows.  The algorithm (Ghahremani et al. 2014; Kaldi feat/pitch-functions.cc, feat/resample.cc, default options) is restated on
// the CPU in oracle/pitch_np.py -- the checker of these kernels, **parity unpinned** against Kaldi itself (absent here):
//   1. 16 kHz -> 4 kHz, Hanning-windowed sinc (cutoff 1 kHz, one zero crossing): 15 taps per output sample        [pitch_downsample]
//   2. mean square of the 4 kHz signal per utterance -> the NCCF ballast (fixed-order tree sum in double)            [pitch_stats]
//   3. per frame (100 samples every 40, needs 182): window mean removed, <w0, w_lag> and |w_lag|^2 at the integer lags 8..82
//      (double), NCCF with and without ballast, both re-sampled to the 417 lags 1/400 s * 1.005^i by an 11-tap windowed sinc
//      (cutoff 2 kHz, five zero crossings; taps tabulated once per call)                                             [pitch_nccf]
//   4. Viterbi over the 417 lags, one workgroup per utterance: local cost 1 - nccf (1 - 10 lag), transition cost
//      0.1 ln(1.005)^2 (i - j)^2, the exact minimum over all j (first one on ties), forward costs kept as remainders, back-pointers
//      in HBM, back-trace by one lane                                                                                [pitch_viterbi]
//   5. pov feature, log pitch minus its POV-weighted mean over +-75 frames, delta over +-2 frames, written behind the n_mel fbank
//      values of the same rows                                                                                       [pitch_post]
#include "kernels.h"

namespace {

constexpr int NLAGS = 417, NIN = 75, MINLAG = 8, MAXLAG = 82, WIN = 100, SHIFT = 40, FULL = WIN + MAXLAG, UPTAPS = 12;
constexpr double SR4 = 4000.0, SR16 = 16000.0;

struct Tables { float lag[NLAGS]; int first[NLAGS]; float w[NLAGS][UPTAPS]; };

__device__ __forceinline__ double filt(double t, double cutoff, double zeros) {
    const double width = zeros / (2.0 * cutoff);
    if (fabs(t) >= width) return 0.0;
    const double win = 0.5 * (1.0 + cos(2.0 * M_PI * cutoff / zeros * t));
    return win * (t != 0.0 ? sin(2.0 * M_PI * cutoff * t) / (M_PI * t) : 2.0 * cutoff);
}

__global__ void pitch_tables_kernel(Tables* tb) {
    const int i = threadIdx.x;
    if (i >= NLAGS) return;
    double lag = 1.0 / 400.0;
    for (int k = 0; k < i; ++k) lag *= 1.005;                        // (SelectLags multiplies step by step)
    tb->lag[i] = (float)lag;
    const double cutoff = 0.5 * SR4, width = 5.0 / (2.0 * cutoff), t = lag - MINLAG / SR4;
    int lo = (int)ceil(SR4 * (t - width)), hi = (int)floor(SR4 * (t + width));
    lo = lo < 0 ? 0 : lo; hi = hi > NIN - 1 ? NIN - 1 : hi;
    tb->first[i] = lo;
    for (int k = 0; k < UPTAPS; ++k) tb->w[i][k] = lo + k <= hi ? (float)(filt((lo + k) / SR4 - t, cutoff, 5.0) / SR4) : 0.f;
}

__device__ __forceinline__ long x4_start(const long* wav_off, int b) { return wav_off[b] / 4 + b; }
__device__ __forceinline__ int pitch_frames(long n16) { const long n4 = (n16 + 3) / 4; return n4 < FULL ? 0 : (int)((n4 - FULL) / SHIFT) + 1; }
__device__ __forceinline__ int fbank_frames(long n16) { return n16 < 400 ? 0 : 1 + (int)((n16 - 400) / 160); }

__global__ __launch_bounds__(256) void pitch_downsample_kernel(const float* __restrict__ wav, const long* __restrict__ wav_off, float* __restrict__ x4) {
    const int b = blockIdx.y;
    const long w0 = wav_off[b], n = wav_off[b + 1] - w0, n4 = (n + 3) / 4;
    const long o = (long)blockIdx.x * 256 + threadIdx.x;
    if (o >= n4) return;
    const double t = o / SR4, width = 1.0 / 2000.0;
    long lo = (long)ceil(SR16 * (t - width)), hi = (long)floor(SR16 * (t + width));
    lo = lo < 0 ? 0 : lo; hi = hi > n - 1 ? n - 1 : hi;
    double acc = 0.0;
    for (long j = lo; j <= hi; ++j) acc += filt(j / SR16 - t, 1000.0, 1.0) / SR16 * (double)wav[w0 + j];
    x4[x4_start(wav_off, b) + o] = (float)acc;
}

__global__ __launch_bounds__(256) void pitch_stats_kernel(const float* __restrict__ x4, const long* __restrict__ wav_off, float* __restrict__ ballast) {
    __shared__ double s1[256], s2[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const long n4 = (wav_off[b + 1] - wav_off[b] + 3) / 4;
    const float* x = x4 + x4_start(wav_off, b);
    double a = 0.0, q = 0.0;
    for (long i = tid; i < n4; i += 256) { const double v = x[i]; a += v; q += v * v; }
    s1[tid] = a; s2[tid] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) { s1[tid] += s1[tid + o]; s2[tid] += s2[tid + o]; } __syncthreads(); }
    if (tid == 0) {
        const double ms = n4 > 0 ? s2[0] / n4 - (s1[0] / n4) * (s1[0] / n4) : 0.0;
        ballast[b] = (float)((ms * WIN) * (ms * WIN) * 7000.0);
    }
}

__global__ __launch_bounds__(448) void pitch_nccf_kernel(const float* __restrict__ x4, const long* __restrict__ wav_off, const float* __restrict__ ballast,
                                                         const Tables* __restrict__ tb, float* __restrict__ nccf_p, float* __restrict__ nccf_v, int maxT) {
    __shared__ double w[FULL];
    __shared__ float sp[NIN], sv[NIN];
    __shared__ double red[2];
    const int b = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
    const int T = pitch_frames(wav_off[b + 1] - wav_off[b]);
    if (t >= T) return;
    const float* x = x4 + x4_start(wav_off, b) + (long)t * SHIFT;
    if (tid < FULL) w[tid] = x[tid];
    __syncthreads();
    if (tid == 0) { double m = 0.0; for (int i = 0; i < WIN; ++i) m += w[i]; red[0] = m / WIN; }      // (mean of the FIRST window only, as Kaldi)
    __syncthreads();
    if (tid < FULL) w[tid] -= red[0];
    __syncthreads();
    if (tid == 0) { double e = 0.0; for (int i = 0; i < WIN; ++i) e += w[i] * w[i]; red[1] = e; }
    __syncthreads();
    if (tid < NIN) {
        const int lag = MINLAG + tid;
        double ip = 0.0, e2 = 0.0;
        for (int i = 0; i < WIN; ++i) { const double v = w[lag + i]; ip += w[i] * v; e2 += v * v; }
        const double nrm = red[1] * e2, bl = ballast[b];
        sp[tid] = nrm + bl > 0.0 ? (float)(ip / sqrt(nrm + bl)) : 0.f;
        sv[tid] = nrm > 0.0 ? (float)(ip / sqrt(nrm)) : 0.f;
    }
    __syncthreads();
    if (tid < NLAGS) {
        const int f = tb->first[tid];
        float a = 0.f, c = 0.f;
#pragma unroll
        for (int k = 0; k < UPTAPS; ++k) {
            const int j = f + k < NIN ? f + k : NIN - 1;               // (weights beyond the support are 0)
            a = fmaf(tb->w[tid][k], sp[j], a);
            c = fmaf(tb->w[tid][k], sv[j], c);
        }
        const long o = ((long)b * maxT + t) * NLAGS + tid;
        nccf_p[o] = a; nccf_v[o] = c;
    }
}

__global__ __launch_bounds__(448) void pitch_viterbi_kernel(const long* __restrict__ wav_off, const Tables* __restrict__ tb, const float* __restrict__ nccf_p,
                                                            const float* __restrict__ nccf_v, unsigned short* __restrict__ back, float* __restrict__ raw, int maxT) {
    __shared__ float fwd[2][448];
    __shared__ float smin[7];
    __shared__ int sbest;
    const int b = blockIdx.x, i = threadIdx.x, lane = i & 63, wave = i >> 6;
    const int T = pitch_frames(wav_off[b + 1] - wav_off[b]);
    if (T == 0) return;
    const float lag = i < NLAGS ? tb->lag[i] : 0.f;
    const float factor = (float)(0.1 * log(1.005) * log(1.005));
    const float* np_b = nccf_p + (long)b * maxT * NLAGS;
    unsigned short* back_b = back + (long)b * maxT * NLAGS;
    auto local = [&](int t) { const float n = np_b[(long)t * NLAGS + i]; return 1.f - n + 10.f * lag * n; };
    fwd[0][i] = i < NLAGS ? local(0) : 3.0e38f;
    __syncthreads();
    for (int t = 1; t < T; ++t) {
        const float* prev = fwd[(t - 1) & 1];
        float best = 3.0e38f; int bj = 0;
        if (i < NLAGS) {
            const float loc = local(t);                                // (requested before the scan, used after it)
            for (int j = 0; j < NLAGS; ++j) {
                const float d = (float)(i - j);
                const float c = fmaf(d * d, factor, prev[j]);
                if (c < best) { best = c; bj = j; }                    // first minimum on ties
            }
            best += loc;
            back_b[(long)t * NLAGS + i] = (unsigned short)bj;
        }
        // forward costs are kept as remainders: subtract the frame's minimum
        float m = best;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o, 64));
        if (lane == 0) smin[wave] = m;
        __syncthreads();
        float mm = smin[0];
#pragma unroll
        for (int k = 1; k < 7; ++k) mm = fminf(mm, smin[k]);
        fwd[t & 1][i] = i < NLAGS ? best - mm : 3.0e38f;
        __syncthreads();
    }
    if (i == 0) {
        const float* last = fwd[(T - 1) & 1];
        int bi = 0; float bv = last[0];
        for (int j = 1; j < NLAGS; ++j) if (last[j] < bv) { bv = last[j]; bi = j; }
        const float* nv_b = nccf_v + (long)b * maxT * NLAGS;
        float* raw_b = raw + (long)b * maxT * 2;
        for (int t = T - 1; t >= 0; --t) {
            raw_b[2 * t] = nv_b[(long)t * NLAGS + bi];
            raw_b[2 * t + 1] = 1.0f / tb->lag[bi];
            if (t > 0) bi = back_b[(long)t * NLAGS + bi];
        }
    }
}

__device__ __forceinline__ float nccf_to_pov(float n) {
    const float nd = fminf(fabsf(n), 1.f);
    const float r = -5.2f + 5.4f * expf(7.5f * (nd - 1.f)) + 4.8f * nd - 2.f * expf(-10.f * nd) + 4.2f * expf(20.f * (nd - 1.f));
    return 1.f / (1.f + expf(-r));
}

__global__ __launch_bounds__(256) void pitch_post_kernel(const long* __restrict__ wav_off, const long* __restrict__ row_off, const float* __restrict__ raw,
                                                         float* __restrict__ feat, int n_mel, int maxT) {
    const int b = blockIdx.x;
    const long n16 = wav_off[b + 1] - wav_off[b];
    const int Tp = pitch_frames(n16), Tf = fbank_frames(n16), T = Tp < Tf ? Tp : Tf;      // paste-feats --length-tolerance=2: the shorter one
    const float* r = raw + (long)b * maxT * 2;
    const int ld = n_mel + 3;
    for (int t = threadIdx.x; t < T; t += 256) {
        const float nccf = fminf(fmaxf(r[2 * t], -1.f), 1.f);
        float* o = feat + (row_off[b] + t) * ld + n_mel;
        o[0] = 2.f * (powf(1.0001f - nccf, 0.15f) - 1.f);
        // (the normalisation window and the deltas run over the pitch tracker's Tp frames, as process-kaldi-pitch-feats sees them)
        const int lo = t - 75 < 0 ? 0 : t - 75, hi = t + 76 > Tp ? Tp : t + 76;
        double num = 0.0, den = 0.0;
        for (int k = lo; k < hi; ++k) { const double p = nccf_to_pov(r[2 * k]); num += p * log((double)r[2 * k + 1]); den += p; }
        o[1] = 2.f * (float)(log((double)r[2 * t + 1]) - num / den);
        auto lp = [&](int k) { k = k < 0 ? 0 : (k > Tp - 1 ? Tp - 1 : k); return logf(r[2 * k + 1]); };
        o[2] = 10.f * (-2.f * lp(t - 2) - lp(t - 1) + lp(t + 1) + 2.f * lp(t + 2)) / 10.f;
    }
}

}  // namespace

static long align256(long x) { return (x + 255) / 256 * 256; }
long mk_pitch_work_bytes(long total_samples, int B, int max_frames) {
    return align256(sizeof(Tables)) + align256((total_samples / 4 + B + 4) * 4) + align256(B * 4) + 2 * align256((long)B * max_frames * NLAGS * 4)
           + align256((long)B * max_frames * NLAGS * 2) + align256((long)B * max_frames * 2 * 4);
}
// the 3 pitch dims of every utterance's rows (n_mel .. n_mel+2 of rows with n_mel + 3 columns); the fbank columns are mk_fbank's
int mk_pitch(const float* wav, const long* wav_off, const long* row_off, long total_samples, long max_samples, int B, int max_frames, int n_mel, float* feat,
             void* work, long work_bytes, hipStream_t s) {
    if (B <= 0 || max_frames <= 0) return 0;
    if (!work || work_bytes < mk_pitch_work_bytes(total_samples, B, max_frames)) { mk_set_error("mk_pitch", "work buffer too small"); return -1; }
    char* p = (char*)work;
    Tables* tb = (Tables*)p; p += align256(sizeof(Tables));
    float* x4 = (float*)p; p += align256((total_samples / 4 + B + 4) * 4);
    float* ballast = (float*)p; p += align256(B * 4);
    float* nccf_p = (float*)p; p += align256((long)B * max_frames * NLAGS * 4);
    float* nccf_v = (float*)p; p += align256((long)B * max_frames * NLAGS * 4);
    unsigned short* back = (unsigned short*)p; p += align256((long)B * max_frames * NLAGS * 2);
    float* raw = (float*)p;
    const long max4 = (max_samples + 3) / 4;
    hipLaunchKernelGGL(pitch_tables_kernel, dim3(1), dim3(448), 0, s, tb);
    hipLaunchKernelGGL(pitch_downsample_kernel, dim3((unsigned)((max4 + 255) / 256), B), dim3(256), 0, s, wav, wav_off, x4);
    hipLaunchKernelGGL(pitch_stats_kernel, dim3(B), dim3(256), 0, s, x4, wav_off, ballast);
    hipLaunchKernelGGL(pitch_nccf_kernel, dim3(max_frames, B), dim3(448), 0, s, x4, wav_off, ballast, tb, nccf_p, nccf_v, max_frames);
    hipLaunchKernelGGL(pitch_viterbi_kernel, dim3(B), dim3(448), 0, s, wav_off, tb, nccf_p, nccf_v, back, raw, max_frames);
    hipLaunchKernelGGL(pitch_post_kernel, dim3(B), dim3(256), 0, s, wav_off, row_off, raw, feat, n_mel, max_frames);
    if (hipGetLastError() != hipSuccess) { mk_set_error("mk_pitch", "launch failed"); return -1; }
    return 0;
}
