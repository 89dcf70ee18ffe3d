"""Prompt-feature front half on MI355X: drop-in for the reference's ``mel_spectrogram``
(/root/reference/cosyvoice/dataset/processor_kaldidata.py:37-74, configured by cosyvoice2/conf/cosyvoice.yaml:120-128:
n_fft = win 1920, hop 480, 80 mels, fmin 0, fmax 8000, center=False) and the 24 kHz ``feat = 2 x token`` trimming of
``frontend_zero_shot`` (cli/frontend.py:141-159).

The reference builds its mel basis with ``librosa.filters.mel`` (a third-party dependency that is not in the container and
not pinned by the reference): ``slaney_mel_basis`` restates librosa's published default algorithm (Slaney mel scale,
htk=False, norm='slaney') — **parity unpinned at that boundary**; everything after the basis is pinned by running the
reference function itself with this basis substituted (tests/golden/make_golden.py).

Device path: reflect pad (data movement, torch) -> frames as a strided view of the padded signal (row stride = hop) times a
hann-windowed cos/-sin DFT basis = one fp32 cv_gemm -> |.| (cv_stft_magnitude) -> mel projection (cv_gemm) -> log-clamp +
channels-first store (cv_log_clamp_channels_first)."""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib as L
from . import ops


def _hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def slaney_mel_basis(sr: int, n_fft: int, n_mels: int, fmin: float, fmax: float) -> np.ndarray:
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) with its defaults (htk=False, norm='slaney', float32):
    triangular filters on the Slaney mel scale, each normalised by 2 / (f_hi - f_lo).  Shape (n_mels, n_fft // 2 + 1)."""
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, None]
    return weights.astype(np.float32)


class MelSpectrogram:
    """Callable with the reference signature: ``y (B, S) float in [-1, 1] -> (B, num_mels, T) float32`` with
    T = (S + 2 * pad - n_fft) // hop + 1, pad = (n_fft - hop) // 2."""

    def __init__(self, n_fft=1920, num_mels=80, sampling_rate=24000, hop_size=480, win_size=1920, fmin=0, fmax=8000, center=False,
                 device="cuda"):
        if center or win_size != n_fft:
            raise ValueError("built for the reference's configuration: center=False, win_size == n_fft")
        if not torch.cuda.is_available():
            raise RuntimeError("cosyvoice_amd needs an MI355X (no CPU fallback)")
        self.n_fft, self.hop, self.n_mels = n_fft, hop_size, num_mels
        self.nbins = n_fft // 2 + 1
        self.device = torch.device(device)
        assert n_fft % 4 == 0 and hop_size % 4 == 0, "16-byte rows of the frame view"
        # hann-windowed DFT basis, rows = [cos k (nbins) | -sin k (nbins) | zero pad], fp64 -> fp32
        n = torch.arange(n_fft, dtype=torch.float64)
        k = torch.arange(self.nbins, dtype=torch.float64)
        ang = 2.0 * math.pi * torch.outer(k, n) / n_fft
        win = torch.hann_window(win_size, periodic=True, dtype=torch.float64)
        self.ld_spec = (2 * self.nbins + 3) // 4 * 4
        basis = torch.zeros(self.ld_spec, n_fft, dtype=torch.float64)
        basis[:self.nbins] = torch.cos(ang) * win
        basis[self.nbins:2 * self.nbins] = -torch.sin(ang) * win
        self.dft = basis.to(self.device, torch.float32).contiguous()
        self.ld_mag = (self.nbins + 3) // 4 * 4
        mel = torch.zeros(num_mels, self.ld_mag)
        mel[:, :self.nbins] = torch.from_numpy(slaney_mel_basis(sampling_rate, n_fft, num_mels, fmin, fmax))
        self.mel = mel.to(self.device).contiguous()

    @torch.no_grad()
    def __call__(self, y: torch.Tensor) -> torch.Tensor:
        if y.dim() == 1:
            y = y.unsqueeze(0)
        y = y.to(self.device, torch.float32)
        B, S = y.shape
        pad = (self.n_fft - self.hop) // 2
        yp = torch.nn.functional.pad(y.unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
        Sp = yp.shape[1]
        T = (Sp - self.n_fft) // self.hop + 1
        ldp = (Sp + 3) // 4 * 4
        buf = torch.zeros(B, ldp, device=self.device)
        buf[:, :Sp] = yp
        spec = torch.empty(B, T, self.ld_spec, device=self.device)
        # frame t of sequence b = buf[b, t*hop : t*hop + n_fft]: A rows overlap (lda = hop), K = n_fft contiguous
        ops.gemm(buf, self.dft, T, self.ld_spec, self.n_fft, batch=B, a_bs=(ldp, 0), lda=self.hop, a_rows=T,
                 out_f32=spec, o32_bs=(spec.stride(0), 0), ldo32=self.ld_spec)
        mag = torch.empty(B * T, self.ld_mag, device=self.device)
        L.check(L.lib().cv_stft_magnitude(C.c_void_p(spec.data_ptr()), self.ld_spec, C.c_void_p(mag.data_ptr()), self.ld_mag, B * T,
                                          self.nbins, C.c_float(1e-9), L.stream_ptr()), "cv_stft_magnitude")
        melo = torch.empty(B * T, self.n_mels, device=self.device)
        ops.gemm(mag, self.mel, B * T, self.n_mels, self.ld_mag, lda=self.ld_mag, out_f32=melo, ldo32=self.n_mels)
        out = torch.empty(B, self.n_mels, T, device=self.device)
        L.check(L.lib().cv_log_clamp_channels_first(C.c_void_p(melo.data_ptr()), self.n_mels, C.c_void_p(out.data_ptr()), B, T,
                                                    self.n_mels, C.c_float(1e-5), L.stream_ptr()), "cv_log_clamp_channels_first")
        return out


def align_prompt_24k(speech_feat: torch.Tensor, speech_token: torch.Tensor):
    """cli/frontend.py:148-152 (cosyvoice2): force feat_len == 2 * token_len.  speech_feat (1, T, 80), speech_token (1, N)
    -> (speech_feat[:, :2n], speech_feat_len, speech_token[:, :n], speech_token_len) with n = min(T // 2, N)."""
    n = min(int(speech_feat.shape[1] / 2), speech_token.shape[1])
    return (speech_feat[:, :2 * n], torch.tensor([2 * n], dtype=torch.int32), speech_token[:, :n], torch.tensor([n], dtype=torch.int32))


def extract_speech_feat(feat_extractor, speech: torch.Tensor):
    """cli/frontend.py:102-106: (1, S) waveform -> (speech_feat (1, T, 80), speech_feat_len (1,) int32)."""
    feat = feat_extractor(speech).squeeze(dim=0).transpose(0, 1).unsqueeze(dim=0)
    return feat, torch.tensor([feat.shape[1]], dtype=torch.int32)
