#!/usr/bin/env python3
"""fp8 accuracy of the trained experts under different calibrations of the e4m3 activation exponents (VERDICT r4 x1 /
weak #3: the depth expert loses 0.6 .. 3.7 points of mIoU on the default plan, run by run).  Both experts are trained
through the HIP fit() on the procedural RGB-D task; the reference labels come from conv_dtype='fp32' (label-exact against
the fp32 oracle); every variant is scored on the same held-out images.  GPU only, no oracle."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import accuracy_evidence as ae  # noqa: E402
from modular_semantic_segmentation_amd import fcn, get_model  # noqa: E402
from modular_semantic_segmentation_amd.datasets.synthetic import data_description, make_rgbd_shapes  # noqa: E402


def predict(variables, m, dtype, heldout, calib, deep=False, method='max', margin=1, start=None):
    net = get_model('fcn')(m, data_description(), m, num_units=ae.U, batch_normalization=False, batchsize=4, conv_dtype=dtype,
                           fp8_deep=deep, fp8_start=start)
    net.variables.update({k: v for k, v in variables.items() if k.startswith(m + '/')})
    net._variables_changed()
    if dtype == 'fp8':
        x = net._to_device(calib[m], torch.float32)
        net.engine.calibrate(x, margin_bits=margin, method=method)
    return net.predict(heldout)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    h, w = 384, 768
    variables, train = ae.train_experts(h, w, steps, batch=8)
    calib = {k: v[:16] for k, v in train.items()}
    heldout = make_rgbd_shapes(48, h, w, seed=1001)
    out = {}
    for m, _ in ae.MODS:
        ref = predict(variables, m, 'fp32', heldout, calib)
        miou_ref, _ = ae._miou(heldout['labels'], ref)
        rec = {'miou_fp32': round(miou_ref, 5)}
        for tag, kw in (('bf16', dict(dtype='bf16')), ('fp8 max+1', dict(dtype='fp8')), ('fp8 max+0', dict(dtype='fp8', margin=0)),
                        ('fp8 mse', dict(dtype='fp8', method='mse')), ('fp8 deep max+1', dict(dtype='fp8', deep=True)),
                        ('fp8 from conv2_2', dict(dtype='fp8', start='conv2_2')), ('fp8 from conv3_1', dict(dtype='fp8', start='conv3_1')),
                        ('fp8 from conv3_2', dict(dtype='fp8', start='conv3_2')), ('fp8 from conv4_1', dict(dtype='fp8', start='conv4_1')),
                        ('fp8 from conv5_1', dict(dtype='fp8', start='conv5_1'))):
            pred = predict(variables, m, heldout=heldout, calib=calib, **kw)
            a, _ = ae._miou(heldout['labels'], pred)
            rec[tag] = {'delta_miou_pp': round(100 * (a - miou_ref), 3), 'agreement': round(float((pred == ref).mean()), 5)}
        out[m] = rec
        print(m, json.dumps(rec), flush=True)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
