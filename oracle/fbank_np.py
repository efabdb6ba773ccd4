"""CPU restatement (numpy, float64) of Kaldi-style log-mel filterbank extraction -- TEST INFRASTRUCTURE ONLY.

The reference repository ships NO feature extraction (its data/ shards are produced offline by a Kaldi/ESPnet recipe that is
not part of it), so this restates the PUBLISHED algorithm of Kaldi's `compute-fbank-feats`
(feat/feature-window.cc ProcessWindow + feat/mel-computations.cc MelBanks, Kaldi 5.5 defaults as used by ESPnet's
`make_fbank_pitch.sh`) and is "parity unpinned" with respect to the reference.  Fixed options:
  sample rate 16 kHz, 25 ms frames (400 samples) every 10 ms (160), snip_edges=true, dither=0, remove_dc_offset=true,
  preemphasis 0.97, povey window, FFT 512, power spectrum, mel bins over [20 Hz, Nyquist], log with floor FLT_EPSILON.
Samples are on the 16-bit PCM scale (Kaldi does not normalise).
"""
import numpy as np

SR, FLEN, FSHIFT, NFFT, LOW, PREEMPH = 16000, 400, 160, 512, 20.0, 0.97
EPS = np.finfo(np.float32).eps


def num_frames(n_samples: int) -> int:
    return 0 if n_samples < FLEN else 1 + (n_samples - FLEN) // FSHIFT


def mel(f):
    return 1127.0 * np.log(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def mel_banks(n_mel: int) -> np.ndarray:
    """[n_mel, NFFT/2] triangular weights (MelBanks::MelBanks: bins 0 .. NFFT/2-1, strict inequalities at the edges)."""
    nbins = NFFT // 2
    lo, hi = mel(LOW), mel(SR / 2.0)
    delta = (hi - lo) / (n_mel + 1)
    fm = mel(np.arange(nbins) * (SR / NFFT))
    w = np.zeros((n_mel, nbins))
    for m in range(n_mel):
        left, center, right = lo + m * delta, lo + (m + 1) * delta, lo + (m + 2) * delta
        up = (fm - left) / (center - left)
        down = (right - fm) / (right - center)
        inside = (fm > left) & (fm < right)
        w[m] = np.where(inside, np.where(fm <= center, up, down), 0.0)
    return w


def povey_window() -> np.ndarray:
    i = np.arange(FLEN)
    return np.power(0.5 - 0.5 * np.cos(2.0 * np.pi * i / (FLEN - 1)), 0.85)


def fbank(wav: np.ndarray, n_mel: int = 80) -> np.ndarray:
    """wav: 1-D samples (PCM scale) -> [T, n_mel] float64 log-mel energies."""
    wav = np.asarray(wav, dtype=np.float64)
    T = num_frames(len(wav))
    out = np.zeros((T, n_mel))
    win, banks = povey_window(), mel_banks(n_mel)
    for t in range(T):
        fr = wav[t * FSHIFT: t * FSHIFT + FLEN].copy()
        fr -= fr.mean()                                           # remove_dc_offset
        fr[1:] -= PREEMPH * fr[:-1]                               # preemphasis (uses the un-emphasised neighbour)
        fr[0] -= PREEMPH * fr[0]
        fr *= win
        spec = np.fft.rfft(fr, NFFT)
        power = (spec.real ** 2 + spec.imag ** 2)[:NFFT // 2]
        out[t] = np.log(np.maximum(banks @ power, EPS))
    return out
