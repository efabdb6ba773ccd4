// Log-mel filterbank extraction on the GPU (north-star "log-mel fbank framing").
// The reference has no feature extraction of its own (its feat.dat shards come from an offline Kaldi/ESPnet recipe), so the
// algorithm is Kaldi's published compute-fbank-feats (feature-window.cc ProcessWindow, mel-computations.cc MelBanks) with
// the options the recipe uses -- see oracle/fbank_np.py, the CPU restatement these kernels are tested against:
//   16 kHz, 400-sample frames every 160 samples, snip_edges, no dither, DC removal, pre-emphasis 0.97, povey window,
//   512-point FFT, power spectrum, triangular mel filters on [20 Hz, 8 kHz], log(max(e, FLT_EPSILON)).
// One workgroup per frame: the frame lives in LDS from the first load to the log; HBM sees the samples once (the 60 %
// overlap of neighbouring frames is served by L2) and the [T][n_mel] rows once, written straight into the ragged
// [sum T_b][n_mel] layout of the reference's feat.dat (src/io/dataset.py:123-139).
#include "kernels.h"

namespace {

constexpr int FLEN = 400, FSHIFT = 160, NFFT = 512, NBIN = NFFT / 2;
constexpr float SR = 16000.f, LOWF = 20.f, PREEMPH = 0.97f;

__device__ __forceinline__ float melf(float f) { return 1127.0f * logf(1.0f + f / 700.0f); }
__device__ __forceinline__ int bitrev9(int x) { return (int)(__brev((unsigned)x) >> 23); }

__global__ __launch_bounds__(256) void fbank_kernel(const float* __restrict__ wav, const long* __restrict__ wav_off,
                                                    const long* __restrict__ row_off, float* __restrict__ feat, int n_mel, int ld, int cap_to_pitch) {
    __shared__ float s_re[NFFT], s_im[NFFT];
    __shared__ float s_twr[NBIN], s_twi[NBIN];
    __shared__ float s_red[4];
    const int b = blockIdx.y, t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long w0 = wav_off[b], n = wav_off[b + 1] - w0;
    int T = n < FLEN ? 0 : 1 + (int)((n - FLEN) / FSHIFT);
    if (cap_to_pitch) {                                    // rows shared with the pitch dims: the shorter of the two streams (paste-feats --length-tolerance=2)
        const long n4 = (n + 3) / 4;
        const int Tp = n4 < 182 ? 0 : (int)((n4 - 182) / 40) + 1;
        T = Tp < T ? Tp : T;
    }
    if (t >= T) return;
    const float* x = wav + w0 + (long)t * FSHIFT;

    // twiddles W512^k = exp(-2 pi i k / 512)
    { float sn, cs; sincospif((float)tid / 256.0f, &sn, &cs); s_twr[tid] = cs; s_twi[tid] = -sn; }
    // frame -> DC removal
    const float a0 = tid < FLEN ? x[tid] : 0.f, a1 = tid + 256 < FLEN ? x[tid + 256] : 0.f;
    float sum = wave_sum(a0 + a1);
    if (lane == 0) s_red[wave] = sum;
    __syncthreads();
    const float mean = ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) / FLEN;
    // pre-emphasis needs the (DC-removed) left neighbour: stage the frame in s_im first
    s_im[tid] = a0 - mean;
    s_im[tid + 256] = tid + 256 < FLEN ? a1 - mean : 0.f;
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int i = tid + h * 256;
        float v = 0.f;
        if (i < FLEN) {
            const float cur = s_im[i], prev = s_im[i > 0 ? i - 1 : 0];
            v = cur - PREEMPH * prev;
            const float wn = 0.5f - 0.5f * cospif(2.0f * (float)i / (float)(FLEN - 1));
            v *= powf(wn, 0.85f);                                      // povey window
        }
        s_re[bitrev9(i)] = v;                                           // bit-reversed order for the in-place DIT FFT
    }
    __syncthreads();
    s_im[tid] = 0.f; s_im[tid + 256] = 0.f;
    __syncthreads();
    // 9 radix-2 stages, one butterfly per thread
#pragma unroll
    for (int s = 1; s <= 9; ++s) {
        const int half = 1 << (s - 1);
        const int j = tid & (half - 1), i0 = ((tid >> (s - 1)) << s) + j, i1 = i0 + half;
        const int k = j << (9 - s);
        const float wr = s_twr[k], wi = s_twi[k];
        const float br = s_re[i1] * wr - s_im[i1] * wi, bi = s_re[i1] * wi + s_im[i1] * wr;
        const float ar = s_re[i0], ai = s_im[i0];
        s_re[i0] = ar + br; s_im[i0] = ai + bi;
        s_re[i1] = ar - br; s_im[i1] = ai - bi;
        __syncthreads();
    }
    // power spectrum of bins 0 .. 255 (Kaldi's MelBanks never looks at the Nyquist bin) -> reuse s_twr
    const float pw = s_re[tid] * s_re[tid] + s_im[tid] * s_im[tid];
    __syncthreads();
    s_twr[tid] = pw;
    __syncthreads();
    // triangular filters: thread m owns mel bin m
    if (tid < n_mel) {
        const float lo = melf(LOWF), hi = melf(SR * 0.5f), delta = (hi - lo) / (float)(n_mel + 1);
        const float left = lo + tid * delta, center = left + delta, right = center + delta;
        // first FFT bin whose mel exceeds `left`: invert the mel scale, then walk (robust to rounding at the edge)
        int i = (int)(700.0f * (expf(left / 1127.0f) - 1.0f) / (SR / NFFT));
        if (i < 0) i = 0;
        while (i > 0 && melf(i * (SR / NFFT)) > left) --i;
        float e = 0.f;
        for (; i < NBIN; ++i) {
            const float fm = melf(i * (SR / NFFT));
            if (fm <= left) continue;
            if (fm >= right) break;
            const float wgt = fm <= center ? (fm - left) / (center - left) : (right - fm) / (right - center);
            e = fmaf(wgt, s_twr[i], e);
        }
        feat[(row_off[b] + t) * ld + tid] = logf(fmaxf(e, 1.1920929e-07f));
    }
}

}  // namespace

int mk_fbank(const float* wav, const long* wav_off, const long* row_off, int B, int max_frames, int n_mel, float* feat, hipStream_t s, int with_pitch) {
    if (n_mel < 1 || n_mel > 256) { mk_set_error("mk_fbank", "1 <= n_mel <= 256"); return -1; }
    if (B <= 0 || max_frames <= 0) return 0;
    hipLaunchKernelGGL(fbank_kernel, dim3(max_frames, B), dim3(256), 0, s, wav, wav_off, row_off, feat, n_mel, with_pitch ? n_mel + 3 : n_mel, with_pitch);
    if (hipGetLastError() != hipSuccess) { mk_set_error("mk_fbank", "launch failed"); return -1; }
    return 0;
}
