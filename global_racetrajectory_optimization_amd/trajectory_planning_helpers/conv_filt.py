"""Host-side shim of tph.conv_filt: moving-average filter (closed signals wrap around)."""
import numpy as np


def conv_filt(signal: np.ndarray, filt_window: int, closed: bool) -> np.ndarray:
    if filt_window % 2 == 0:
        raise RuntimeError("Window width of moving average filter must be odd!")
    w = int((filt_window - 1) / 2)
    ker = np.ones(filt_window) / float(filt_window)
    if closed:
        tmp = np.concatenate((signal[-w:], signal, signal[:w]), axis=0)
        return np.convolve(tmp, ker, mode="same")[w:-w]
    out = np.copy(signal)
    out[w:-w] = np.convolve(signal, ker, mode="same")[w:-w]
    return out
