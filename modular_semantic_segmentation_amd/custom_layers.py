"""Host-side counterparts of xview/models/custom_layers.py that the MI355X path needs:
the bilinear transposed-conv constant and the check that an imported deconv kernel IS that
constant (the HIP path applies it as a depthwise 2x2-tap interpolation)."""
import numpy as np


def bilinear_filter(filter_shape):
    """Bilinear transposed-conv kernel, layout [kh, kw, out, in], zero off the channel diagonal
    (reference: custom_layers.py:8-25 bilinear_filter_initializer)."""
    kh, kw, cout, cin = filter_shape
    factor = np.ceil(kh / 2.0)
    center = (2 * factor - 1 - factor % 2) / (2.0 * factor)
    w1 = 1 - np.abs(np.arange(kh) / factor - center)
    w2 = 1 - np.abs(np.arange(kw) / factor - center)
    weights = np.zeros(filter_shape)
    idx = np.arange(min(cout, cin))
    weights[:, :, idx, idx] = np.outer(w1, w2)[:, :, None]
    return weights.astype(np.float32)


def is_bilinear_filter(kernel, atol=1e-6):
    """True if `kernel` ([k,k,U,U]) equals the constant the reference initialises and never trains
    (simple_fcn.py:80-83,117-130: deconvs are always `trainable=False`)."""
    kernel = np.asarray(kernel)
    if kernel.ndim != 4 or kernel.shape[0] != kernel.shape[1] or kernel.shape[2] != kernel.shape[3]:
        return False
    return bool(np.allclose(kernel, bilinear_filter(kernel.shape), rtol=0, atol=atol))
