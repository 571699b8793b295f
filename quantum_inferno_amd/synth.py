"""Seeded synthetic records for benchmarks and demos: per channel a logarithmic
chirp from fs * 2^-14 to 0.4 fs under a 5 % Tukey taper, plus white noise 8 bits below the signal's
standard deviation, numpy.random.default_rng(20250213 + channel).  This build's own generator."""
import numpy as np

SEED = 20250213


def log_chirp(n, fs, channel=0, n_channels=1, dtype=np.float32, seed=SEED):
    k = np.arange(n, dtype=np.float64)
    f0, f1 = fs * 2.0 ** -14, 0.4 * fs
    rate = np.log(f1 / f0) / (n / fs)
    phase = 2 * np.pi * f0 * (np.exp(rate * k / fs) - 1.0) / rate + 2 * np.pi * channel / n_channels
    edge = int(np.floor(0.05 * (n - 1) / 2.0))
    ramp = 0.5 * (1 + np.cos(np.pi * (-1 + 2.0 * np.arange(edge + 1) / 0.05 / (n - 1))))
    taper = np.ones(n)
    taper[: edge + 1] = ramp
    taper[n - edge - 1 :] = ramp[::-1]
    x = np.sin(phase) * taper
    x = x + (2.0 ** -8) * np.std(x) * np.random.default_rng(seed + channel).standard_normal(n)
    return x.astype(dtype)


def channels(n, fs, first, count, total, dtype=np.float32):
    """[count, n] block of channels first .. first+count-1 out of `total`."""
    return np.stack([log_chirp(n, fs, c, total, dtype) for c in range(first, first + count)])
