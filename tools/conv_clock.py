#!/usr/bin/env python3
"""In-kernel clock of the conv kernels' item loops (GPU box only; diagnostic build `make -C csrc stamp`).

The question: are the bf16 MFMA convs held at 0.50-0.56 of peak by the clock the chip holds under load, or
by their schedule?  /opt/skills/guides/MI355X_MICROARCH.md, 'DVFS give-back' item 6, prescribes the direct test: stamp
s_memtime (shader cycles) and s_memrealtime (constant 100 MHz) once around the loop in a separate diagnostic build, after
>= 2 s of back-to-back launches, clock = delta cycles / delta realtime x 100 MHz, median over workgroups.

For every (layer shape, tile configuration, operand data) this prints / records: the median in-kernel clock, the shader
cycles of the loop (median per workgroup), the wall time per launch and the TFLOP/s.  Cycles x clock separates the two
explanations: equal cycles at a lower clock = power; more cycles at the same clock = schedule.

The stamps live in tools/build/libxview_hip_stamp.so only (never the shipped library); this script points the loader at it.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from modular_semantic_segmentation_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, 'tools', 'build', 'libxview_hip_stamp.so')
from modular_semantic_segmentation_amd import ops  # noqa: E402

# name, stride of the map against the input, cin, cout, fused pool
LAYERS = {'conv1_2': (1, 64, 64, True), 'conv2_1': (2, 64, 128, False), 'conv2_2': (2, 128, 128, True),
          'conv3_1': (4, 128, 256, False), 'conv3_2': (4, 256, 256, False), 'conv4_2': (8, 512, 512, False),
          'conv5_1': (16, 512, 512, False)}
SLOTS = 2048


def read_clock(gen):
    buf = np.zeros(4 * SLOTS, dtype=np.uint64)
    fn = getattr(_lib.lib(), 'xv_debug_read_clock_g%d' % gen)
    fn.restype = ctypes.c_int
    rc = fn(ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(buf.nbytes))
    assert rc == 0, rc
    b = buf.reshape(SLOTS, 4)
    b = b[b[:, 3] > b[:, 1]]
    cyc = (b[:, 2] - b[:, 0]).astype(np.float64)
    rt = (b[:, 3] - b[:, 1]).astype(np.float64)
    return cyc, rt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--height', type=int, default=384)
    ap.add_argument('--width', type=int, default=768)
    ap.add_argument('--seconds', type=float, default=2.0, help='back-to-back launches before the stamps are read')
    ap.add_argument('--cases', default='conv3_2:26,conv3_2:17,conv4_2:26,conv4_2:17,conv3_1:26,conv1_2:17,conv1_2:26,conv2_1:17,conv2_1:26,conv2_2:17,'
                                       'conv2_2:26,conv5_1:22,conv5_1:27',
                    help='layer:configuration pairs (configuration 25, the 32x32x16 form in profiles/r4_conv_inkernel_clock.json, was '
                         'retired after that measurement)')
    ap.add_argument('--data', default='normal,relu,zero')
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'r4_conv_inkernel_clock.json'))
    args = ap.parse_args()
    handle = _lib.lib()
    assert hasattr(handle, 'xv_debug_read_clock_g4'), 'not the stamp build'
    rows = []
    print('%-9s cfg data    clock GHz (p10 / median / p90)   loop kcyc   us/launch  TFLOP/s' % 'layer')
    for case in args.cases.split(','):
        name, cfg = case.split(':')
        cfg = int(cfg)
        s, cin, cout, pool = LAYERS[name]
        h, w = args.height // s, args.width // s
        gen = 5 if cfg in (27, 28) else (4 if cfg in (24, 25, 26) else 2)
        for data in args.data.split(','):
            torch.manual_seed(0)
            x = ops.Act(args.batch, h, w, cin)
            wt = torch.randn(3, 3, cin, cout, device='cuda') * (1.0 / (9 * cin) ** 0.5)
            if data != 'zero':
                x.interior().normal_()
                if data == 'relu':
                    x.interior().clamp_(min=0)
            else:
                wt.zero_()
            wp = ops.pack_conv_weights(wt)
            b = torch.zeros(cout, device='cuda')
            y = ops.Act(args.batch, h, w, cout)
            q = ops.Act(args.batch, h // 2, w // 2, cout) if (pool and cfg not in (22, 27)) else None
            try:
                ops.conv2d_fwd(x, wp, b, 3, y=y, pooled=q, cfg=cfg)
            except _lib.XvError:
                print('%-9s %3d %-6s  (shape refused)' % (name, cfg, data))
                continue
            torch.cuda.synchronize()
            getattr(handle, 'xv_debug_reset_clock_g%d' % gen)()
            # >= `seconds` of back-to-back launches, then one timed block whose last launch leaves the stamps
            t0 = time.time()
            while time.time() - t0 < args.seconds:
                for _ in range(50):
                    ops.conv2d_fwd(x, wp, b, 3, y=y, pooled=q, cfg=cfg)
                torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            iters = 200
            e0.record()
            for _ in range(iters):
                ops.conv2d_fwd(x, wp, b, 3, y=y, pooled=q, cfg=cfg)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / iters * 1e3
            cyc, rt = read_clock(gen)
            ghz = cyc / rt * 0.1
            flops = 2.0 * args.batch * h * w * cin * cout * 9
            row = {'layer': name, 'cfg': cfg, 'data': data, 'shape': [args.batch, h, w, cin, cout],
                   'clock_ghz_median': float(np.median(ghz)), 'clock_ghz_p10': float(np.percentile(ghz, 10)),
                   'clock_ghz_p90': float(np.percentile(ghz, 90)), 'loop_cycles_median': float(np.median(cyc)),
                   'loop_us_median': float(np.median(rt) / 100.0), 'workgroups': int(len(cyc)), 'us_per_launch': us,
                   'tflops': flops / us / 1e6}
            rows.append(row)
            print('%-9s %3d %-6s  %5.3f / %5.3f / %5.3f            %8.1f   %8.1f   %6.0f' % (
                name, cfg, data, row['clock_ghz_p10'], row['clock_ghz_median'], row['clock_ghz_p90'],
                row['loop_cycles_median'] / 1e3, us, row['tflops']), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, 'w') as f:
        json.dump({'method': 'delta s_memtime / delta s_memrealtime x 100 MHz around the item loop, one stamp pair per workgroup, '
                             'read after >= %.1f s of back-to-back launches (MI355X_MICROARCH.md DVFS give-back item 6); '
                             'diagnostic build -DXV_CLOCK_STAMP' % args.seconds,
                   'device': torch.cuda.get_device_name(0), 'rows': rows}, f, indent=1)
    print('wrote', args.out)


if __name__ == '__main__':
    main()
