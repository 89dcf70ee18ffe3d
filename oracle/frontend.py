"""Oracle for the prompt-feature front half (TEST INFRASTRUCTURE — see oracle/__init__.py).

Restates mel_spectrogram of /root/reference/cosyvoice/dataset/processor_kaldidata.py:37-74 with torch.stft, line by line.
The mel basis comes from librosa there (absent here, unpinned by the reference): this oracle takes the basis as an argument
(cosyvoice_amd.frontend.slaney_mel_basis in the tests), so that boundary stays **parity unpinned**."""
import torch


def mel_spectrogram(y: torch.Tensor, mel_basis: torch.Tensor, n_fft=1920, hop_size=480, win_size=1920, center=False) -> torch.Tensor:
    window = torch.hann_window(win_size)                                                     # :48
    pad = int((n_fft - hop_size) / 2)
    y = torch.nn.functional.pad(y.unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)      # :50-53
    spec = torch.view_as_real(torch.stft(y, n_fft, hop_length=hop_size, win_length=win_size, window=window, center=center,
                                         pad_mode="reflect", normalized=False, onesided=True, return_complex=True))  # :55-67
    spec = torch.sqrt(spec.pow(2).sum(-1) + 1e-9)                                            # :69
    spec = torch.matmul(mel_basis, spec)                                                     # :71
    return torch.log(torch.clamp(spec, min=1e-5))                                            # :27-28,72
