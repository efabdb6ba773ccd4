"""CPU restatement (numpy, float64) of Kaldi's pitch features -- TEST INFRASTRUCTURE ONLY.

The shipped configs train on 83-dim features = 80 log-mel bins + 3 pitch dims (config/transformer/pretrain/fometa-hkust.yaml:13,
README.md:24); the reference itself ships NO extraction code.  The 3 dims come from ESPnet's
`make_fbank_pitch.sh`: `compute-kaldi-pitch-feats | process-kaldi-pitch-feats`, pasted behind the fbank rows with
`paste-feats --length-tolerance=2`.  This file restates that PUBLISHED algorithm (Ghahremani et al., "A pitch extraction
algorithm tuned for automatic speech recognition", ICASSP 2014; Kaldi feat/pitch-functions.cc, feat/resample.cc) with Kaldi's
default options and is **parity unpinned**: neither Kaldi nor any other implementation of it exists in the build container
(probed: torchaudio, kaldi_native_fbank, librosa, kaldiio are all absent), so it is checked through the algorithm's invariants only
(tests/test_oracle_golden.py).

compute-kaldi-pitch-feats, offline (the whole utterance is seen at once; defaults of PitchExtractionOptions):
  16 kHz in; resampled to 4 kHz (windowed-sinc low-pass at 1 kHz, 1 zero crossing: LinearResample); frames of 25 ms every 10 ms on
  the 4 kHz signal (100 samples, shift 40, snip_edges: a frame needs its window + the largest lag); per frame the normalised
  cross-correlation NCCF(lag) = <w0, w_lag> / sqrt(|w0|^2 |w_lag|^2 + ballast) at the integer lags 8..82 (window mean removed),
  with ballast = (mean square of the 4 kHz signal * 100)^2 * 7000 for the tracker and 0 for the voicing output; both re-sampled
  to 1.005-ratio spaced lags between 1/400 s and 1/50 s (ArbitraryResample, 5 zero crossings, cutoff 2 kHz); Viterbi over those
  lags with local cost 1 - nccf (1 - 10 lag) and transition cost 0.1 ln(1.005)^2 (delta index)^2; output per frame
  (NCCF without ballast at the chosen lag, 1 / lag in Hz).
process-kaldi-pitch-feats (defaults of ProcessPitchOptions): [pov feature, normalised log pitch, delta log pitch]
  pov feature = 2 ((1.0001 - nccf)^0.15 - 1); log pitch minus its POV-weighted mean over +-75 frames, times 2; delta of the log
  pitch over +-2 frames (weights k / 10, edges replicated), times 10.  Kaldi adds Gaussian noise of sigma 0.005 to the log pitch
  before differentiating (dithering; a random stream that no other implementation can reproduce): `delta_noise` passes such a
  vector in, default none.
"""
import math

import numpy as np

SR_IN, SR = 16000, 4000
WIN, SHIFT = 100, 40                       # 25 ms / 10 ms at 4 kHz
MIN_F0, MAX_F0, SOFT_MIN_F0 = 50.0, 400.0, 10.0
PENALTY, DELTA_PITCH, BALLAST = 0.1, 0.005, 7000.0
LP_CUTOFF, LP_WIDTH = 1000.0, 1            # resampling to 4 kHz
UP_WIDTH = 5                               # NCCF upsampling
OUTER_MIN_LAG = int(math.floor(SR / MAX_F0)) - UP_WIDTH // 2        # 8
OUTER_MAX_LAG = int(math.ceil(SR / MIN_F0)) + UP_WIDTH // 2         # 82
NLAG_IN = OUTER_MAX_LAG - OUTER_MIN_LAG + 1                           # 75 measured lags
FULL = WIN + OUTER_MAX_LAG                                            # 182 samples a frame touches


def num_frames(n_samples_16k: int) -> int:
    n4 = (n_samples_16k + 3) // 4
    return 0 if n4 < FULL else (n4 - FULL) // SHIFT + 1


def _filter(t, cutoff, zeros):
    """Hanning-windowed sinc of LinearResample / ArbitraryResample::FilterFunc"""
    t = np.asarray(t, dtype=np.float64)
    width = zeros / (2.0 * cutoff)
    win = np.where(np.abs(t) < width, 0.5 * (1.0 + np.cos(2.0 * np.pi * cutoff / zeros * t)), 0.0)
    with np.errstate(divide="ignore", invalid="ignore"):
        sinc = np.where(t != 0.0, np.sin(2.0 * np.pi * cutoff * t) / (np.pi * t), 2.0 * cutoff)
    return win * sinc


def downsample(wav: np.ndarray) -> np.ndarray:
    """16 kHz -> 4 kHz (LinearResample with flush): output n at time n / 4000, taps at the input samples within +-0.5 ms"""
    wav = np.asarray(wav, dtype=np.float64)
    n_out = (len(wav) + 3) // 4
    width = LP_WIDTH / (2.0 * LP_CUTOFF)
    out = np.zeros(n_out)
    for n in range(n_out):
        t = n / SR
        lo = int(math.ceil(SR_IN * (t - width))); hi = int(math.floor(SR_IN * (t + width)))
        lo = max(lo, 0); hi = min(hi, len(wav) - 1)
        j = np.arange(lo, hi + 1)
        out[n] = np.dot(_filter(j / SR_IN - t, LP_CUTOFF, LP_WIDTH) / SR_IN, wav[j])
    return out


def lags() -> np.ndarray:
    """SelectLags: 1 / max_f0 * 1.005^i up to 1 / min_f0 (seconds)"""
    out, lag = [], 1.0 / MAX_F0
    while lag <= 1.0 / MIN_F0:
        out.append(lag)
        lag *= 1.0 + DELTA_PITCH
    return np.asarray(out)


def upsample_matrix(lg: np.ndarray) -> np.ndarray:
    """ArbitraryResample weights [n_lags, NLAG_IN]: NCCF measured at lag (OUTER_MIN_LAG + j) / SR -> NCCF at lg[i]"""
    cutoff = 0.5 * SR
    width = UP_WIDTH / (2.0 * cutoff)
    W = np.zeros((len(lg), NLAG_IN))
    for i, l in enumerate(lg):
        t = l - OUTER_MIN_LAG / SR
        lo = max(int(math.ceil(SR * (t - width))), 0); hi = min(int(math.floor(SR * (t + width))), NLAG_IN - 1)
        j = np.arange(lo, hi + 1)
        W[i, j] = _filter(j / SR - t, cutoff, UP_WIDTH) / SR
    return W


def nccf_frames(x4: np.ndarray):
    """-> (nccf_pitch [T, NLAG_IN], nccf_pov [T, NLAG_IN]) at the integer lags"""
    T = 0 if len(x4) < FULL else (len(x4) - FULL) // SHIFT + 1
    mean_square = float(np.mean(x4 ** 2) - np.mean(x4) ** 2) if len(x4) else 0.0
    ballast = (mean_square * WIN) ** 2 * BALLAST
    a, b = np.zeros((T, NLAG_IN)), np.zeros((T, NLAG_IN))
    for t in range(T):
        w = x4[t * SHIFT: t * SHIFT + FULL].copy()
        w -= w[:WIN].mean()                                        # (the mean of the first window only, as Kaldi does)
        w0 = w[:WIN]
        e1 = float(w0 @ w0)
        for k, lag in enumerate(range(OUTER_MIN_LAG, OUTER_MAX_LAG + 1)):
            wl = w[lag: lag + WIN]
            ip, nrm = float(w0 @ wl), e1 * float(wl @ wl)
            a[t, k] = ip / math.sqrt(nrm + ballast) if nrm + ballast > 0.0 else 0.0
            b[t, k] = ip / math.sqrt(nrm) if nrm > 0.0 else 0.0
    return a, b


def viterbi(nccf_pitch: np.ndarray, lg: np.ndarray) -> np.ndarray:
    """best lag index per frame: local cost 1 - nccf (1 - soft_min_f0 lag), transition PENALTY ln(1.005)^2 (i - j)^2"""
    T, N = nccf_pitch.shape
    if T == 0:
        return np.zeros(0, dtype=np.int64)
    factor = PENALTY * math.log(1.0 + DELTA_PITCH) ** 2
    idx = np.arange(N)
    trans = factor * (idx[:, None] - idx[None, :]) ** 2            # [to i, from j]
    local = 1.0 - nccf_pitch + SOFT_MIN_F0 * lg[None, :] * nccf_pitch
    fwd = local[0].copy()
    back = np.zeros((T, N), dtype=np.int64)
    for t in range(1, T):
        tot = fwd[None, :] + trans
        back[t] = np.argmin(tot, axis=1)                           # (first minimum on ties)
        fwd = tot[idx, back[t]] + local[t]
        fwd -= fwd.min()                                           # Kaldi keeps the remainder only
    best = np.zeros(T, dtype=np.int64)
    best[-1] = int(np.argmin(fwd))
    for t in range(T - 1, 0, -1):
        best[t - 1] = back[t, best[t]]
    return best


def viterbi_f32(nccf_pitch: np.ndarray, lg: np.ndarray) -> np.ndarray:
    """viterbi() in the arithmetic of the GPU kernel (csrc/pitch.hip pitch_viterbi_kernel): NCCF and lags rounded to fp32, local
    cost fma(10 lag, n, 1 - n), candidate fma((i - j)^2, factor, prev[j]) scanned over j with a strict <, forward costs kept as fp32
    remainders.  Neighbouring lags of the 1.005-ratio grid differ by ~1e-7 in path cost (viterbi_margins), i.e. by about one fp32
    ulp of a forward cost on a noise frame: THIS, not the algorithm, is what a float64 and an fp32 tracker disagree about."""
    f32, f64 = np.float32, np.float64
    n32 = nccf_pitch.astype(f32)
    T, N = n32.shape
    if T == 0:
        return np.zeros(0, dtype=np.int64)
    lag = lg.astype(f32)
    factor = f32(0.1 * math.log(1.005) * math.log(1.005))
    d2 = ((np.arange(N)[:, None] - np.arange(N)[None, :]) ** 2).astype(f64)          # [to i, from j], exact
    tl = (f32(10.0) * lag).astype(f32)
    local = (tl[None, :].astype(f64) * n32.astype(f64) + (f32(1.0) - n32).astype(f64)).astype(f32)    # one rounding: fma
    fwd = local[0].copy()
    back = np.zeros((T, N), dtype=np.int64)
    idx = np.arange(N)
    for t in range(1, T):
        cand = (d2 * f64(factor) + fwd[None, :].astype(f64)).astype(f32)              # fma(d2, factor, prev[j])
        back[t] = np.argmin(cand, axis=1)                                              # first minimum == strict < scan
        best = (cand[idx, back[t]] + local[t]).astype(f32)
        fwd = (best - best.min()).astype(f32)
    out = np.zeros(T, dtype=np.int64)
    out[-1] = int(np.argmin(fwd))
    for t in range(T - 1, 0, -1):
        out[t - 1] = back[t, out[t]]
    return out


def viterbi_margins(nccf_pitch: np.ndarray, lg: np.ndarray) -> np.ndarray:
    """how well defined the tracker's decision is, per frame: the cost of the best lag path that passes through a DIFFERENT lag at
    frame t, minus the cost of the optimal path (forward + backward min-sum over the same local / transition costs as viterbi()).
    A margin of 1e-7 means two lags are tied to within float rounding at that frame -- an implementation in another precision may
    legitimately take the other one there; frames with a clear margin admit one answer."""
    T, N = nccf_pitch.shape
    if T == 0:
        return np.zeros(0)
    factor = PENALTY * math.log(1.0 + DELTA_PITCH) ** 2
    idx = np.arange(N)
    trans = factor * (idx[:, None] - idx[None, :]) ** 2
    local = 1.0 - nccf_pitch + SOFT_MIN_F0 * lg[None, :] * nccf_pitch
    F = np.zeros((T, N)); B = np.zeros((T, N))
    F[0] = local[0]
    for t in range(1, T):
        F[t] = (F[t - 1][None, :] + trans).min(axis=1) + local[t]
    for t in range(T - 2, -1, -1):
        B[t] = (B[t + 1][None, :] + local[t + 1][None, :] + trans.T).min(axis=1)
    tot = F + B                                                       # cost of the best path through (t, i)
    srt = np.sort(tot, axis=1)
    return srt[:, 1] - srt[:, 0]


def pitch_decision_margins(wav: np.ndarray) -> np.ndarray:
    """viterbi_margins of one utterance's tracker problem, [T]"""
    lg = lags()
    a, _ = nccf_frames(downsample(wav))
    return viterbi_margins(a @ upsample_matrix(lg).T, lg)


def compute_kaldi_pitch(wav: np.ndarray, fp32_tracker: bool = False) -> np.ndarray:
    """-> [T, 2]: (NCCF at the chosen lag, computed without ballast; pitch in Hz).  fp32_tracker: the Viterbi recursion in the GPU
    kernel's fp32 arithmetic (viterbi_f32) instead of float64."""
    x4 = downsample(wav)
    lg = lags()
    W = upsample_matrix(lg)
    a, b = nccf_frames(x4)
    ap, bp = a @ W.T, b @ W.T
    best = viterbi_f32(ap, lg) if fp32_tracker else viterbi(ap, lg)
    T = len(best)
    return np.stack([bp[np.arange(T), best], 1.0 / lg[best]], axis=1) if T else np.zeros((0, 2))


def nccf_to_pov_feature(n):
    n = np.clip(n, -1.0, 1.0)
    return np.power(1.0001 - n, 0.15) - 1.0


def nccf_to_pov(n):
    nd = np.minimum(np.abs(n), 1.0)
    r = -5.2 + 5.4 * np.exp(7.5 * (nd - 1.0)) + 4.8 * nd - 2.0 * np.exp(-10.0 * nd) + 4.2 * np.exp(20.0 * (nd - 1.0))
    return 1.0 / (1.0 + np.exp(-r))


def process_kaldi_pitch(raw: np.ndarray, delta_noise=None) -> np.ndarray:
    """[T, 2] (nccf, pitch) -> [T, 3] (pov feature * 2, normalised log pitch * 2, delta log pitch * 10)"""
    T = len(raw)
    if T == 0:
        return np.zeros((0, 3))
    nccf, logp = raw[:, 0], np.log(raw[:, 1])
    pov = nccf_to_pov(nccf)
    out = np.zeros((T, 3))
    out[:, 0] = 2.0 * nccf_to_pov_feature(nccf)
    for t in range(T):
        lo, hi = max(0, t - 75), min(T, t + 76)
        out[t, 1] = 2.0 * (logp[t] - float(pov[lo:hi] @ logp[lo:hi]) / float(pov[lo:hi].sum()))
    lp = logp + (np.asarray(delta_noise, dtype=np.float64) if delta_noise is not None else 0.0)
    pad = np.concatenate([[lp[0]] * 2, lp, [lp[-1]] * 2])
    out[:, 2] = 10.0 * sum(k * pad[2 + k: 2 + k + T] for k in (-2, -1, 1, 2)) / 10.0
    return out


def pitch_feats(wav: np.ndarray, delta_noise=None, fp32_tracker: bool = False) -> np.ndarray:
    return process_kaldi_pitch(compute_kaldi_pitch(wav, fp32_tracker), delta_noise)


def fbank_pitch(wav: np.ndarray, n_mel: int = 80) -> np.ndarray:
    """the recipe's 83-dim rows: fbank | pitch, truncated to the shorter of the two (paste-feats --length-tolerance=2)"""
    from . import fbank_np
    f, p = fbank_np.fbank(wav, n_mel), pitch_feats(wav)
    assert abs(len(f) - len(p)) <= 2, (len(f), len(p))
    T = min(len(f), len(p))
    return np.concatenate([f[:T], p[:T]], axis=1)
