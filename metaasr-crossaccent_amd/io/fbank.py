"""GPU log-mel filterbank front-end feeding the reference's shard layout.

The reference trains from pre-extracted `<accent>/<split>/feat.dat` matrices (src/io/dataset.py:123-139) and ships no
extraction code; `extract` computes Kaldi-style fbank rows with libmasr's HIP kernel (masr_fbank, one workgroup per frame)
and `write_feat_shard` stores them in exactly that layout (NPY-format feat.dat opened with np.load(mmap_mode='r'),
ilens.npy).  `extract(..., pitch=True)` appends the 3 Kaldi pitch dims of the shipped 83-dim rows (masr_fbank_pitch:
compute-kaldi-pitch-feats | process-kaldi-pitch-feats with Kaldi's defaults, as ESPnet's make_fbank_pitch.sh runs them).
"""
import ctypes as C
from pathlib import Path

import numpy as np
import torch

from .. import _cabi

FRAME_LEN, FRAME_SHIFT = 400, 160


def num_frames(n_samples: int) -> int:
    return 0 if n_samples < FRAME_LEN else 1 + (n_samples - FRAME_LEN) // FRAME_SHIFT


def num_pitch_frames(n_samples: int) -> int:
    n4 = (n_samples + 3) // 4                                  # the tracker works on the signal resampled to 4 kHz
    return 0 if n4 < 182 else (n4 - 182) // 40 + 1             # 100-sample window + the largest lag (82), every 40 samples


def extract(wavs, n_mel: int = 80, device="cuda:0", pitch: bool = False):
    """wavs: list of 1-D float tensors / arrays on the 16-bit PCM scale (16 kHz).
    Returns (feat [sum T_b, n_mel (+ 3 with pitch)] fp32 on `device`, ilens int64 [B]); with pitch an utterance has
    min(fbank frames, pitch frames) rows, as `paste-feats --length-tolerance=2` leaves them."""
    if not torch.cuda.is_available():
        raise RuntimeError("fbank.extract needs a HIP device (MI355X); there is no CPU path")
    dev = torch.device(device)
    ws = [torch.as_tensor(np.asarray(w, dtype=np.float32) if not torch.is_tensor(w) else w, dtype=torch.float32).reshape(-1) for w in wavs]
    lens = [int(w.numel()) for w in ws]
    ilens = torch.tensor([min(num_frames(n), num_pitch_frames(n)) if pitch else num_frames(n) for n in lens], dtype=torch.int64)
    wav_off = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int64, device=dev)
    row_off = torch.tensor(np.concatenate([[0], np.cumsum(ilens.numpy())[:-1]]), dtype=torch.int64, device=dev)
    wav = torch.cat(ws).to(dev) if ws else torch.zeros(0, device=dev)
    feat = torch.empty(int(ilens.sum()), n_mel + (3 if pitch else 0), dtype=torch.float32, device=dev)
    if len(ws) and int(ilens.max()) > 0 and pitch:
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        L = _cabi.lib()
        nbytes = int(L.masr_fbank_pitch_work_bytes(sum(lens), len(ws), int(ilens.max())))
        work = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _cabi.check(L.masr_fbank_pitch(C.c_void_p(wav.data_ptr()), C.c_void_p(wav_off.data_ptr()), C.c_void_p(row_off.data_ptr()), sum(lens), max(lens),
                                       len(ws), int(ilens.max()), n_mel, C.c_void_p(feat.data_ptr()), C.c_void_p(work.data_ptr()), nbytes, stream), "masr_fbank_pitch")
    elif len(ws) and int(ilens.max()) > 0:
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _cabi.check(_cabi.lib().masr_fbank(C.c_void_p(wav.data_ptr()), C.c_void_p(wav_off.data_ptr()), C.c_void_p(row_off.data_ptr()),
                                           len(ws), int(ilens.max()), n_mel, C.c_void_p(feat.data_ptr()), stream), "masr_fbank")
    return feat, ilens


def write_feat_shard(dirpath, feat: torch.Tensor, ilens: torch.Tensor):
    """<dirpath>/feat.dat (NPY header + [sum T, idim] float32, as the reference memory-maps it) and ilens.npy"""
    d = Path(dirpath)
    d.mkdir(parents=True, exist_ok=True)
    arr = feat.detach().cpu().numpy().astype(np.float32)
    mm = np.lib.format.open_memmap(d / "feat.dat", mode="w+", dtype=np.float32, shape=arr.shape)
    mm[:] = arr
    mm.flush()
    del mm
    np.save(d / "ilens.npy", ilens.cpu().numpy().astype(np.int64))
