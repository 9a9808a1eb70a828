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


def dense_deconv_as_conv3x3(kernel, stride):
    """Re-arrange a dense transposed-conv kernel [k, k, out, in] (tf.layers.conv2d_transpose layout,
    custom_layers.py:71-121) with k = 2 * stride and 'same' padding into the HWIO kernel [3, 3, in, s*s*out] of a
    stride-1 3x3 'same' convolution that computes all s*s output phases at the input resolution:

        [TF1] out[s*q + p] = sum_i in[i] * W[p + pad - (i - q)*s],  pad = (k - s) // 2 = s / 2
        =>  phase p sees in[q + d] for d in {-1, 0, 1} through tap ky = p + pad - d*s (when 0 <= ky < k)

    so K[d_y + 1, d_x + 1, ci, (p_y*s + p_x)*out + co] = W[ky, kx, co, ci].  xv_deconv_dense_fwd runs it on the MFMA
    conv and un-shuffles the phases (depth-to-space)."""
    kernel = np.asarray(kernel, np.float32)
    k, k2, cout, cin = kernel.shape
    s = int(stride)
    if k != k2 or k != 2 * s:
        raise NotImplementedError('dense transposed conv: kernel %dx%d with stride %d (only k = 2*stride)' % (k, k2, s))
    pad = (k - s) // 2
    out = np.zeros((3, 3, cin, s * s * cout), np.float32)
    for py in range(s):
        for dy in (-1, 0, 1):
            ky = py + pad - dy * s
            if not 0 <= ky < k:
                continue
            for px in range(s):
                for dx in (-1, 0, 1):
                    kx = px + pad - dx * s
                    if 0 <= kx < k:
                        out[dy + 1, dx + 1, :, (py * s + px) * cout:(py * s + px + 1) * cout] = kernel[ky, kx].T
    return out
