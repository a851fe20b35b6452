"""Audio front-end of the `a` modality on the GPU (utils/audio_utils.py of the reference).

``wav2filterbanks`` keeps the reference's signature and return convention; the STFT + mel + log run in
``logmel_kernel``.  The mel filter bank is a restatement of the published algorithm of
``librosa.filters.mel`` (librosa==0.10.2.post1: Slaney mel scale, ``htk=False``, ``norm='slaney'``);
librosa is not in this image, so that part is **parity unpinned** (DESIGN.md section 2).
"""
import numpy as np
import torch

audio_opts = {"sample_rate": 16000, "n_fft": 512, "win_length": 320, "hop_length": 160, "n_mel": 80}


def _hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def mel_filterbank(sr=16000, n_fft=512, n_mels=80, fmin=0.0, fmax=8000.0):
    """(n_mels, 1+n_fft/2) float32 triangular filters, Slaney-normalised (area 1 per band)."""
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    weights = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return (weights * enorm[:, None]).astype(np.float32)


def load_wav(path, fr=0, to=100000000, sample_rate=16000):
    """audio_utils.py:20-25: raw int16 samples, not normalised."""
    from scipy.io import wavfile
    _, wav = wavfile.read(path)
    return wav


def wav2filterbanks(wav, mel_basis=None, engine=None):
    """wav: tensor b x T (float, int16 scale) -> (features b x T//160 x 80, None, None, mel_basis)."""
    from ._lib import Engine
    assert len(wav.shape) == 2, "Need batch of wavs as input"
    eng = engine or Engine.get()
    if mel_basis is None:
        mel_basis = torch.from_numpy(mel_filterbank(audio_opts["sample_rate"], audio_opts["n_fft"], audio_opts["n_mel"], 0,
                                                    audio_opts["sample_rate"] / 2)).to(eng.device)
    feats = eng.logmel(wav, mel_basis)
    return feats, None, None, mel_basis
