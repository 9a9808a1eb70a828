#!/usr/bin/env python3
"""Headline benchmark: images/sec of the two-stream (RGB + depth) SimpleFCN experts + Bayes
fusion + argmax at 768x384 on MI355X (BASELINE.json configs[1]); protocol of the reference's
experiments/timing.py (inputs resident on the device, wall clock around the whole pipeline).

  python bench.py [--gpus N --steps K --warmup W] [--batch B] [--height 384 --width 768]
                  [--fusion bayes|dirichlet|joint] [--dtype bf16|fp8] [--mode infer|train]
                  [--no-cpu-baseline] [--no-accuracy] [--no-extra]
N > 1 is launched by the driver through torch.distributed.run (one rank per GPU); images are
independent, so ranks shard the batch with no data-path collective ("weak" scaling) and only the
timing is reduced (MAX over ranks).  Rank 0 prints ONE JSON line:

  metric / value / roofline / cpu_baseline   the headline workload (contract of the driver)
  accuracy     mean IoU of the HIP bf16 and fp8 paths against the fp32 oracle on TRAINED experts (N = 1)
  extra        the other BASELINE.json configurations, each with its own roofline fraction: the reference's
               timing.py protocol (batch 1, tf.ones input, label map fetched to the host), 1024x512 Bayes,
               2048x1024 bf16 and fp8, Dirichlet fusion, the training step
  train_dp     N > 1: the data-parallel training step (bucketed RCCL all-reduce overlapped with backward) with the
               exposed all-reduce time -- the collective path an inference scaling curve never touches
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

_JSON_OUT = sys.stdout
PEAK_TFLOPS = {'bf16': 2500.0, 'fp8': 5000.0, 'fp32': 157.3}     # dense MFMA, MI355X_MICROARCH.md chip table (fp32: the
#                                                                  f32-input matrix instruction = the fp32 vector rate)
C, U = 12, 64


def conv_flops_per_image(h, w, cin):
    """Credited conv FLOPs of one FCN stream (181.23 GFLOP at 384x768 RGB)."""
    layers = [(cin, 64, 1), (64, 64, 1), (64, 128, 2), (128, 128, 2), (128, 256, 4), (256, 256, 4), (256, 256, 4),
              (256, 512, 8), (512, 512, 8), (512, 512, 8), (512, 512, 16), (512, 512, 16), (512, 512, 16)]
    f = sum(2.0 * (h // s) * (w // s) * ci * co * 9 for ci, co, s in layers)
    f += 2.0 * (h // 8) * (w // 8) * 512 * U + 2.0 * (h // 16) * (w // 16) * 512 * U + 2.0 * h * w * U * C
    return f


def adapnet_flops_per_image(h, w, cin):
    """Algorithmic conv FLOPs of one AdapNet stream (adapnet.py:103-173): every conv at its own kernel size, stride and
    resolution (7x7 stride 2 = 49 taps at H/2, atrous 3x3 = 9 taps), first_deconvolution_conv counted for the
    num_units channels the x2 deconv reads."""
    from modular_semantic_segmentation_amd.adapnet import BLOCKS
    f = 2.0 * h * w * cin * 64 * 9 + 2.0 * (h // 2) * (w // 2) * 64 * 64 * 49
    c, s = 64, 4
    for name, kind, a in BLOCKS:
        if kind == 'a':
            mid, cout, stride, shortcut = a
            s *= stride
            px = (h // s) * (w // s)
            f += 2.0 * px * (c * mid + 9 * mid * mid + mid * cout + (c * cout if shortcut else 0))
        else:
            f1, f2, cout, _, _, shortcut = a
            px = (h // s) * (w // s)
            f += 2.0 * px * (c * f1 + 9 * f1 * f2 + f2 * cout + (c * cout if shortcut else 0))
        if name == 'block_layer_7':
            f += 2.0 * px * cout * U
        c = cout
    return f + 2.0 * (h // 16) * (w // 16) * 2048 * U


def build_model(device, fusion='bayes', expert='fcn', batch=16, dtype='bf16', streamk=False, fp8_deep=False, fp8_start=None):
    from modular_semantic_segmentation_amd import get_model
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'notebook_868.npz'))
    desc = ({'rgb': 'float32', 'depth': 'float32', 'labels': 'int32'},
            {'rgb': (None, None, 3), 'depth': (None, None, 1), 'labels': (None, None)}, C)
    common = dict(data_description=desc, num_units=U, num_channels={'rgb': 3, 'depth': 1}, expert_model=expert,
                  class_prior='data', batchsize=batch, seed=1, device=str(device), conv_dtype=dtype, streamk=streamk,
                  fp8_deep=fp8_deep, fp8_start=fp8_start,
                  fp8_agreement=0)     # (speed records of FIXED e4m3 plans: random-init logits are nearly degenerate, the
    #                                     accuracy guard of calibrate() would send every expert back to bf16; the guarded plan is
    #                                     chosen on the TRAINED experts of the accuracy leg and passed in as fp8_start)
    if fusion == 'joint':
        # the reference's joint baseline fusion_fcn (experiments/timing.py:24-45): two VGG16 trunks + fused decoder
        net = get_model('fusion_fcn')({'rgb': 'rgb', 'depth': 'depth'}, {'rgb': 3, 'depth': 1}, U, C,
                                      batchsize=batch, seed=1, device=str(device))
        net.variables['depth_conv1_1/kernel'] = net.variables['depth_conv1_1/kernel'] / 256.0
        net._variables_changed()
        return net
    if fusion == 'bayes':
        net = get_model('bayes_fusion')(confusion_matrices={'rgb': g['cm_rgb'], 'depth': g['cm_depth']},
                                        prefixes={'rgb': 'rgb', 'depth': 'depth'}, **common)
    else:
        rng = np.random.default_rng(2)
        params = {'rgb': rng.uniform(0.5, 4.0, (C, C)), 'depth': rng.uniform(0.5, 4.0, (C, C)),
                  'class_counts': g['cm_depth'].sum(1)}
        net = get_model('dirichlet_fusion')(dirichlet_params=params, modalities=['rgb', 'depth'], sigma=1.0,
                                            delta=1e-2, beta=1e-2, **common)
    # a trained depth expert absorbs the raw uint16 range in conv1_1; random init needs the scale
    first = 'depth/conv1_1/kernel' if expert == 'fcn' else 'depth/block_0_1/kernel'
    net.variables[first] = net.variables[first] / 256.0
    net._variables_changed()
    return net


def synthetic_batch(device, batch, h, w, seed, ones=False):
    """rgb = integers(0, 256), depth = integers(0, 65536); ones=True: the degenerate tf.ones input of
    experiments/timing.py:26-27."""
    if ones:
        return {'rgb': torch.ones((batch, h, w, 3), device=device), 'depth': torch.ones((batch, h, w, 1), device=device)}
    gen = torch.Generator(device='cpu').manual_seed(seed)
    return {'rgb': torch.randint(0, 256, (batch, h, w, 3), generator=gen).float().to(device),
            'depth': torch.randint(0, 65536, (batch, h, w, 1), generator=gen).float().to(device)}


def pick_cpu_threads(h, w):
    """The thread count that runs a mid-network conv fastest (all logical CPUs is not it on a big host)."""
    avail = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    probe_x = torch.randn(1, 256, h // 4, w // 4)
    probe_w = torch.randn(256, 256, 3, 3)
    best = (None, 1)
    for nthr in sorted({min(avail, c) for c in (8, 16, 32, 64, 128)}):
        torch.set_num_threads(nthr)
        torch.nn.functional.conv2d(probe_x, probe_w, padding=1)
        t0 = time.perf_counter()
        for _ in range(3):
            torch.nn.functional.conv2d(probe_x, probe_w, padding=1)
        dtp = time.perf_counter() - t0
        if best[0] is None or dtp < best[0]:
            best = (dtp, nthr)
    torch.set_num_threads(best[1])
    return best[1], avail


def cpu_baseline_random(args, variables, cms, cores, avail):
    """The oracle (op-for-op fp32 restatement of the reference graph) timed on this host's cores on a bounded sample
    of the headline workload: whole 768x384 RGB-D images of the synthetic input, one at a time."""
    from oracle import fcn_oracle as fo
    from oracle import fusion_oracle as fu
    rng = np.random.default_rng(0)
    mats = [cms['rgb'].astype('float32').T, cms['depth'].astype('float32').T]

    def one():
        rgb = rng.integers(0, 256, (1, args.height, args.width, 3)).astype(np.float32)
        depth = rng.integers(0, 65536, (1, args.height, args.width, 1)).astype(np.float32)
        la = fo.argmax_last(fo.softmax(fo.fcn_forward(rgb, variables, 'rgb', 'fp32', keep=['score'])['score']))
        lb = fo.argmax_last(fo.softmax(fo.fcn_forward(depth, variables, 'depth', 'fp32', keep=['score'])['score']))
        return np.argmax(fu.bayes_fusion([la, lb], mats, 'data')[0], -1)

    one()                                    # warm-up
    t0 = time.perf_counter()
    n = 0
    while n < 64 and (time.perf_counter() - t0) < 12.0:
        one()
        n += 1
    dt = time.perf_counter() - t0
    return {'value': n / dt, 'unit': 'images/s', 'cores': cores, 'kind': 'port',
            'sample': '%d RGB-D images of %dx%d, batch 1, fp32 PyTorch-CPU oracle (two FCN experts + Bayes fusion), '
                      '%.1f s, %d threads (fastest of a probe; %d logical CPUs available)'
                      % (n, args.width, args.height, dt, cores, avail)}


def accuracy_and_cpu_baseline(args, device, cores, avail):
    """cpu_baseline leg with the accuracy evidence folded in: both SimpleFCN experts are trained on the procedural
    RGB-D task through the HIP fit(), the held-out images go through the HIP path (bf16 and fp8) and through the
    fp32 oracle -- that oracle pass is the timed CPU sample (same workload: two experts + fusion, 768x384, batch 1)."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import accuracy_evidence as ae
    from modular_semantic_segmentation_amd.datasets.synthetic import make_rgbd_shapes
    h, w = args.height, args.width
    variables, train = ae.train_experts(h, w, args.accuracy_steps, batch=8, device=device)
    measure = {k: v[:16] for k, v in train.items()}
    heldout = make_rgbd_shapes(args.accuracy_images, h, w, seed=1001)
    hip, cms, dparams = ae.hip_predictions(variables, measure, heldout, device=device)
    ref, dt, n = ae.oracle_predictions(variables, heldout, cms, dparams)
    acc = ae.compare(hip, ref, heldout['labels'])
    # 'fp8': the models' default, the accuracy-guarded plan per expert (calibrate() -> FcnEngine.calibrate_guarded);
    # 'fp8_fixed_plan': round 5's one global plan (e4m3 operands from conv2_2 on)
    for key, guarded in (('fp8', True), ('fp8_fixed_plan', False)):
        hip8 = ae.hip_predictions_fp8(variables, measure, heldout, cms, device=device, guarded=guarded)
        acc[key] = {'plan': hip8['plan']}
        for k in ('rgb', 'depth', 'bayes'):
            a, _ = ae._miou(heldout['labels'], hip8[k])
            acc[key][k] = {'miou_hip_fp8': round(a, 5),
                           'delta_miou_pp_vs_fp32': round(100 * (a - acc[k]['miou_fp32_oracle']), 4),
                           'label_agreement_vs_fp32': round(float((hip8[k] == ref[k]).mean()), 6)}
    acc['protocol'] = ('both experts trained %d Adam steps x 8 images on procedural RGB-D shapes (datasets/synthetic.py) at '
                       '%dx%d through the HIP fit(); %d held-out images; the same trained weights through the HIP path and '
                       'the fp32 CPU oracle; mean IoU over classes 1..%d (base_model.py:329)'
                       % (args.accuracy_steps, w, h, n, C - 1))
    base = {'value': n / dt, 'unit': 'images/s', 'cores': cores, 'kind': 'port',
            'sample': '%d held-out RGB-D images of %dx%d, batch 1, fp32 PyTorch-CPU oracle (two trained FCN experts + Bayes '
                      'and Dirichlet fusion), %.1f s, %d threads (fastest of a probe; %d logical CPUs available)'
                      % (n, w, h, dt, cores, avail)}
    return acc, base


def fence(device, world, dist):
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize(device)


def timed_blocks(step, steps, device, world, dist, min_seconds):
    """The timed region, made robust: blocks of EXACTLY `steps` steps, each bracketed by barrier + synchronize on both
    sides and reduced to the MAX over ranks, repeated until `min_seconds` of timed work have accumulated (a single block
    of 20 steps is 0.1 s, and box / clock state moved that by +-5 %).  Returns the list of block times (identical on every
    rank: the loop exit depends only on all-reduced numbers)."""
    times = []
    while True:
        fence(device, world, dist)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence(device, world, dist)
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        times.append(dt)
        if sum(times) >= min_seconds or len(times) >= 400:
            return times


def block_stats(times, steps):
    med = float(np.median(times))
    return {'ms_per_step': round(med / steps * 1e3, 4), 'ms_per_step_min': round(min(times) / steps * 1e3, 4),
            'ms_per_step_max': round(max(times) / steps * 1e3, 4), 'timed_blocks': len(times),
            'timed_seconds': round(sum(times), 3)}


def measure_inference(net, batch, device, steps, warmup, graph=True, serial=None, world=1, dist=None, fetch=False,
                      min_seconds=2.0):
    """Timed region + per-kernel roofline pass of one inference configuration.  Returns (block times of the timed region
    (max over ranks), [per-iteration seconds] if fetch, conv profile list, block times of the serialised pass)."""
    from modular_semantic_segmentation_amd import ops
    # serial: True = the experts back to back on one stream, False = side by side on two, None = the model's own choice
    # (basic_fusion_model.expert_streams: two streams while one expert's launches leave CUs idle)
    net.concurrent_experts = None if serial is None else not serial
    net._graph = None

    def step():
        return net._predict_batch(batch)

    for _ in range(warmup):
        step()
    net._bench_graph_used = False
    if graph:
        try:
            net.capture_graph(batch)
        except Exception as exc:       # noqa: BLE001  (a failed capture must not cost the line: the eager step is the same work)
            print('bench.py: hipGraph capture failed (%s: %s); timing eager launches' % (type(exc).__name__, exc), file=sys.stderr)
            net._graph = None
            torch.cuda.synchronize(device)
            graph = False
    if graph:
        g = net._graph[0]
        net._bench_graph_used = True

        def step():             # noqa: F811  (inputs already sit in the graph's static buffers)
            g.replay()
            return net._graph[2]
        step()
    # clock ramp: a GPU that has just come out of idle needs about a second under load before its clocks settle (the same
    # command measured 3 150 -> 3 300 images/s over consecutive runs on one box); untimed, on top of the W warmup steps
    ramp = float(os.environ.get('XV_BENCH_RAMP_S', '1.0'))
    tr = time.perf_counter()
    while ramp > 0 and time.perf_counter() - tr < ramp:
        for _ in range(10):
            step()
        torch.cuda.synchronize(device)
    per_iter = []
    if fetch:
        # experiments/timing.py:38-45: wall clock around every sess.run, whose fetch hands the label map to the host.  The
        # fetch is the product's (host_pipeline.ResultFetcher, what predict() does per batch): class indices below 256 cross
        # PCIe as ONE BYTE per pixel (xv_narrow_labels) into pinned memory; fetch='int64' also widens them to the reference's
        # int64 on the host inside the timed iteration; fetch='pageable' is rounds 3-4's `.cpu()` of the int64 map.
        import ctypes
        from modular_semantic_segmentation_amd import _lib
        probe = step()
        wire = torch.empty(probe.numel(), dtype=torch.uint8, device=device)
        pinned = torch.empty(probe.numel(), dtype=torch.uint8).pin_memory()
        host64 = np.empty(probe.numel(), np.int64)
        stream = torch.cuda.current_stream(device)

        def fstep():
            ts = time.perf_counter()
            out = step()
            if fetch == 'pageable' or out.dtype != torch.int64 or C > 256:
                out.cpu()
            else:
                _lib.check(_lib.lib().xv_narrow_labels(ctypes.c_void_p(out.data_ptr()), out.numel(), ctypes.c_void_p(wire.data_ptr()),
                                                       ctypes.c_void_p(stream.cuda_stream)), 'xv_narrow_labels')
                pinned.copy_(wire, non_blocking=True)
                stream.synchronize()
                if fetch == 'int64':
                    np.copyto(host64, pinned.numpy(), casting='unsafe')
            per_iter.append(time.perf_counter() - ts)
        times = timed_blocks(fstep, steps, device, world, dist, min_seconds)
    else:
        times = timed_blocks(step, steps, device, world, dist, min_seconds)

    # Per-kernel roofline pass: the same steps with the two experts serialised on one stream, so that a HIP-event
    # pair around a conv launch (recorded on the launch stream) times that kernel alone (in the timed region above
    # the RGB and depth experts overlap on two streams).  Same block scheme (this rank only: no collective).
    prof = []
    net._graph = None
    net.concurrent_experts = False
    net._predict_batch(batch)
    ops.CONV_PROFILE = prof
    # (an event pair around an EAGER launch also spans the gap if the host delivers the launch late: a collector pause of a
    # few milliseconds inside this pass showed as one block at half the rate -- `frac_min` 0.07-0.25 of a record whose other
    # 40 blocks agree to 2 %; the median never saw it.  No collections during the pass.)
    import gc
    gc_was = gc.isenabled()
    gc.disable()
    try:
        times_serial = timed_blocks(lambda: net._predict_batch(batch), steps, device, 1, None, min(min_seconds, 1.0))
    finally:
        if gc_was:
            gc.enable()
    ops.CONV_PROFILE = None
    net.concurrent_experts = None if serial is None else not serial
    return times, per_iter, prof, times_serial


def roofline_of(prof, kinds, peak, kernel, times_serial, steps, traffic=None):
    """Algorithmic FLOPs of the selected launches / their summed HIP-event durations, per block of the serialised pass;
    `achieved` is the MEDIAN block, the extremes are reported next to it."""
    nblk = max(len(times_serial), 1)
    per = len(prof) // nblk
    blocks = []
    fl_all = sec_all = 0.0
    cnt = 0
    for bi in range(nblk):
        fl = sec = 0.0
        for kind, flops, e0, e1 in prof[bi * per:(bi + 1) * per]:
            if kind in kinds:
                fl += flops
                sec += e0.elapsed_time(e1) * 1e-3
                cnt += 1
        if sec > 0:
            blocks.append(fl / sec / 1e12)
            fl_all += fl
            sec_all += sec
    if not blocks:
        return None
    achieved = float(np.median(blocks))
    return {'bound': 'mfma', 'kernel': kernel, 'achieved': round(achieved, 2), 'peak': peak, 'unit': 'TFLOP/s',
            'frac': round(achieved / peak, 4), 'traffic': traffic, 'launches': cnt,
            'avg_launch_ms': round(sec_all / cnt * 1e3, 4), 'gflop_per_launch': round(fl_all / cnt / 1e9, 2),
            'frac_min': round(min(blocks) / peak, 4), 'frac_max': round(max(blocks) / peak, 4), 'blocks': len(blocks),
            'measured': 'HIP events per launch, experts serialised on one stream (median of %d blocks of %d steps, '
                        '%.3f ms/step)' % (len(blocks), steps, float(np.median(times_serial)) / steps * 1e3)}


def committed_traffic(batch, h, w):
    """HBM bytes per conv launch from the committed rocprofv3 FETCH_SIZE / WRITE_SIZE passes of this same command
    (tools/pmc_summary.py): PMC counters cannot be read from inside the process.  The counters were collected on ONE
    version of the kernel: a summary of another version (source hash) is not reported."""
    pdir = os.path.join(ROOT, 'profiles')
    tfiles = sorted(f for f in os.listdir(pdir) if f.endswith('_conv_traffic.json')) if os.path.isdir(pdir) else []
    if not tfiles or (h, w) != (384, 768):
        return None
    tj = json.load(open(os.path.join(pdir, tfiles[-1])))
    srcs = [os.path.join(ROOT, 'modular_semantic_segmentation_amd', 'csrc', f)
            for f in ('conv_mfma.hip', 'conv_f8_dma.hip', 'conv_col_dma.hip', 'conv_first_fused.hip')]
    sha = hashlib.sha256(b''.join(open(f, 'rb').read() for f in srcs)).hexdigest()[:16]
    if tj.get('batch') == batch and tj.get('kernel_source_sha256_16') == sha:
        return tj.get('hbm_bytes_per_launch')
    return None


def extra_inference(device, label, fusion, batch, h, w, dtype='bf16', steps=10, warmup=2, ones=False, fetch=False,
                    streamk=False, fp8_deep=False, zero_operands=False, scalar_f32=False, fp8_start=None):
    """One more BASELINE.json configuration as an `extra` record (its own model, graph and roofline pass).
    zero_operands: a DIAGNOSTIC, not a workload -- every conv kernel, bias and input zero, so no MFMA operand toggles: the
    rate the same binaries reach when power does not hold the clock down ('Generation 4')."""
    if dtype == 'fp32':
        # the label-exact mode: A/B against round 4's vector-ALU kernel through the same engine (scalar_f32)
        from modular_semantic_segmentation_amd import fcn_exact
        fcn_exact.SCALAR_KERNEL = bool(scalar_f32)
    net = build_model(device, fusion=fusion, batch=batch, dtype=dtype, streamk=streamk, fp8_deep=fp8_deep, fp8_start=fp8_start)
    data = synthetic_batch(device, batch, h, w, seed=77, ones=ones)
    if zero_operands:
        for key in list(net.variables):
            if 'upscore' not in key and key.rsplit('/', 1)[-1] in ('kernel', 'bias'):
                net.variables[key] = np.zeros_like(net.variables[key])
        net._variables_changed()
        data = {k: torch.zeros_like(v) for k, v in data.items()}
    if dtype == 'fp8':
        net.calibrate(data)
    times, per_iter, prof, dt_serial = measure_inference(net, data, device, steps, warmup, fetch=fetch, min_seconds=1.0,
                                                         graph=dtype != 'fp32')
    dt = float(np.median(times))
    rec = {'workload': label, 'images_per_step': batch, 'dtype': dtype, 'value': round(batch * steps / dt, 2),
           'unit': 'images/s'}
    rec.update(block_stats(times, steps))
    if fetch:
        rec['seconds_per_image_mean_std'] = [round(float(np.mean(per_iter)), 6), round(float(np.std(per_iter)), 6)]
    if dtype == 'fp8':
        from modular_semantic_segmentation_amd.basic_fusion_model import fp8_plan_report
        rec['fp8_plan'] = {m: r['chosen'] for m, r in fp8_plan_report(net).items()}
        rec['roofline'] = roofline_of(prof, ('k3f8',), PEAK_TFLOPS['fp8'],
                                      'conv_f8_dma_kernel (v_mfma_scale_f32_32x32x64_f8f6f4, 3x3 launches on e4m3 operands; conv_mfma_kernel<F8> where a map does not tile in 16x32)',
                                      dt_serial, steps)
        rec['roofline_bf16_layers'] = roofline_of(prof, ('k3',), PEAK_TFLOPS['bf16'], 'the 3x3 convs left on bf16 operands (conv1_2)',
                                                  dt_serial, steps)
    elif dtype == 'fp32':
        rec['roofline'] = roofline_of(prof, ('k3f32',), PEAK_TFLOPS['fp32'],
                                      'conv_f32_kernel (round 4: fp32 FMAs on the vector ALU)' if scalar_f32 else
                                      'conv_f32_mfma_kernel (v_mfma_f32_32x32x2_f32: every 3x3 launch, conv1_1 included)', dt_serial, steps)
        from modular_semantic_segmentation_amd import fcn_exact
        fcn_exact.SCALAR_KERNEL = False
    else:
        rec['roofline'] = roofline_of(prof, ('k3', 'k3pair'), PEAK_TFLOPS['bf16'], 'conv_dma4_kernel / conv_dma5_kernel / conv_first_pair_kernel (+ fallback generations): every 3x3 launch incl. the fused first pair',
                                      dt_serial, steps)
    flops_img = conv_flops_per_image(h, w, 3) + conv_flops_per_image(h, w, 1)
    rec['conv_tflops_end_to_end'] = round(batch * steps * flops_img / dt / 1e12, 2)
    del net
    torch.cuda.empty_cache()
    return rec


def host_boundary_record(device, samples=256, batch=16, h=384, w=768):
    """predict() / score() / fit() fed from HOST numpy arrays -- the API the reference's callers use (base_model.py:180-331)
    -- through the pipelined boundary (host_pipeline.py: pinned staging, uploads / label downloads on copy streams, the step
    replayed from a hipGraph after two batches), next to the serial path (XV_HOST_PIPELINE=0: pageable copies, .cpu() per batch).
    The results are kept until the clock stops: freeing a 600 MB label array is the caller's business."""
    from modular_semantic_segmentation_amd import host_pipeline
    net = build_model(device, batch=batch)
    rng = np.random.default_rng(0)
    uniq = 32
    reps = (samples + uniq - 1) // uniq
    data = {'rgb': np.tile(rng.integers(0, 256, (uniq, h, w, 3)).astype(np.float32), (reps, 1, 1, 1))[:samples],
            'depth': np.tile(rng.integers(0, 65536, (uniq, h, w, 1)).astype(np.float32), (reps, 1, 1, 1))[:samples],
            'labels': np.tile(rng.integers(-1, C, (uniq, h, w)).astype(np.int32), (reps, 1, 1))[:samples]}
    inputs = {k: v for k, v in data.items() if k != 'labels'}
    rec = {'workload': 'predict() and score() over %d HOST-resident %dx%d RGB-D samples, batchsize %d (two SimpleFCN experts + '
                       'Bayes fusion)' % (samples, w, h, batch), 'unit': 'images/s', 'dtype': 'bf16'}

    def timed(fn, arg, reps=3):
        best = None
        for _ in range(reps):
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            result = fn(arg)
            torch.cuda.synchronize(device)
            dt = time.perf_counter() - t0
            del result
            best = dt if best is None else min(best, dt)
        return best

    for mode in ('pipelined', 'serial'):
        host_pipeline.ENABLED = mode == 'pipelined'
        net._graph = None
        try:
            warm = {k: v[:4 * batch] for k, v in data.items()}
            net.predict({k: v for k, v in warm.items() if k != 'labels'})
            sub = samples if mode == 'pipelined' else min(samples, 8 * batch)
            tp = timed(net.predict, {k: v[:sub] for k, v in inputs.items()})
            ts = timed(net.score, {k: v[:sub] for k, v in data.items()})
            rec[mode] = {'predict_images_per_s': round(sub / tp, 1), 'score_images_per_s': round(sub / ts, 1), 'samples': sub,
                         'host_bytes_per_image': {'in': int(h * w * 4 * 4), 'labels_out': int(h * w * 8)}}
        finally:
            host_pipeline.ENABLED = True
    rec['fit'] = host_fit_rate(device, {k: data[k][:8 * batch] for k in ('rgb', 'labels')}, batch)
    rec['value'] = rec['pipelined']['predict_images_per_s']
    rec['note'] = ('value = predict(); input staging 12.6 MB per RGB-D pair by worker threads into pinned memory, H2D on a copy '
                   'stream, labels back as one byte per pixel and widened to int64 into the result array')
    del net
    torch.cuda.empty_cache()
    return rec


def host_fit_rate(device, data, batch, steps=32):
    """fit() of one RGB expert from HOST arrays (base_model.py:180-261: the reference feeds its training loop from a tf.data
    prefetch): images/s of `steps` training steps including staging and upload, next to the same steps from batches resident in
    HBM."""
    from modular_semantic_segmentation_amd import get_model
    desc = ({'rgb': 'float32', 'labels': 'int32'}, {'rgb': (None, None, 3), 'labels': (None, None)}, C)
    net = get_model('fcn')('rgb', desc, 'rgb', num_units=U, batch_normalization=False, batchsize=batch, learning_rate=1e-4,
                           trainer='adam', seed=1, device=str(device), output_dir=None)
    out = {'steps': steps, 'batchsize': batch}
    resident = {k: torch.from_numpy(v[:4 * batch]).to(device) for k, v in data.items()}
    for name, d in (('host', data), ('resident', resident)):
        net.fit(d, 4, output=False)
        torch.cuda.synchronize(device)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            net.fit(d, steps, output=False)
            torch.cuda.synchronize(device)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out[name + '_images_per_s'] = round(steps * batch / best, 1)
    del net
    torch.cuda.empty_cache()
    return out


def committed_gpu_busy(name):
    """GPU-busy fraction of the training step from the committed kernel trace of the same command (tools/dp_regime.sh ->
    tools/gpu_busy.py: union of the kernel intervals / wall time over whole steps) -- a trace cannot be taken from inside the
    process.  None when the record is not in profiles/."""
    pdir = os.path.join(ROOT, 'profiles')
    files = sorted(f for f in os.listdir(pdir) if f.endswith('_train_%s_gpu_busy.json' % name)) if os.path.isdir(pdir) else []
    if not files:
        return None
    rec = json.load(open(os.path.join(pdir, files[-1])))
    rec['source'] = 'profiles/' + files[-1]
    rec.pop('largest_gaps_us', None)
    return rec


def dp_regime_records(args, device, dist):
    """The data-parallel regime on ONE GPU: the training step at 4 images -- the reference's shipped batchsize
    (experiments/example_config.yaml:18; with batch_normalization: true, :27) and the per-rank batch of a 32-image step on 8
    GPUs -- next to the 16-image step, plain and with batch norm, each with images/s per image relative to its 16-image form
    and the GPU-busy fraction of the committed kernel trace (is the chip waiting for the host's eager launches?)."""
    out = []
    ref = {}
    for bn in (False, True):
        for b in (16, 4):
            if b == 16 and not bn:
                continue        # the 16-image plain step is the record in front of these
            tr = measure_training(args, device, 1, 0, dist, b, 8, 2, batch_norm=bn)
            tr['workload'] = ('SimpleFCN RGB expert training step 768x384 (fwd + bwd + Adam%s), %d images%s'
                              % (', batch_normalization=true' if bn else '', b,
                                 ' -- the reference\'s shipped batchsize / the per-rank batch of a 32-image DP-8 step' if b == 4 else ''))
            if b == 4:
                tr['gpu_busy'] = committed_gpu_busy('b4_%s' % ('bn' if bn else 'plain'))
                if not bn:
                    # the API-level loop at this batch size: fit() from HOST arrays with the loss fetched every step (what the
                    # reference's training loop does, base_model.py:257) next to batches resident in HBM
                    rng = np.random.default_rng(0)
                    host = {'rgb': rng.integers(0, 256, (64, args.height, args.width, 3)).astype(np.float32),
                            'labels': rng.integers(-1, C, (64, args.height, args.width)).astype(np.int32)}
                    tr['fit_api'] = host_fit_rate(device, host, 4, steps=32)
            ref[(bn, b)] = tr['value']
            out.append(tr)
    return out, ref


def dirichlet_fit_record(device, batch=4, h=1024, w=2048, nbatches=4):
    """BASELINE configs[3]'s own workload: DirichletFusion.fit() at 2048x1024 (dirichlet_mix.py:175-273) -- both experts'
    forward to class probabilities, the sufficient statistics S[c,k] = sum_{label=c} log(1e-10 + p[k]) on the device
    (suffstats_kernel), the [C,C] reduction to the host, the Newton fit of the class-conditional Dirichlets on the host
    (dirichlet_fit.py).  Inputs resident in HBM; images/s counts RGB-D pairs through the statistics pass; the host fit is
    timed beside it (it does not depend on the number of images)."""
    from modular_semantic_segmentation_amd import ops
    net = build_model(device, fusion='dirichlet', batch=batch)
    n = batch * nbatches
    gen = torch.Generator(device='cpu').manual_seed(5)
    one = {'rgb': torch.randint(0, 256, (batch, h, w, 3), generator=gen).float().to(device),
           'depth': torch.randint(0, 65536, (batch, h, w, 1), generator=gen).float().to(device),
           'labels': torch.randint(0, C, (batch, h, w), generator=gen).int().to(device)}
    data = {k: v.repeat((nbatches,) + (1,) * (v.dim() - 1)) for k, v in one.items()}
    # random-init experts saturate their softmax (logits of scale 1e2: log(1e-10 + p) = -23 for eleven of twelve classes, a
    # statistic no trained expert produces and on which the host fit runs its 10 000 iterations per class): scale each
    # expert's score layer so that its logits have unit spread, as a trained expert's do
    from modular_semantic_segmentation_amd.basic_fusion_model import run_experts
    outs = run_experts(net, {k: v[:1] for k, v in one.items()}, ('score',))
    for m in net.modalities:
        spread = float(outs[m]['score'].std())
        for key in ('%s/score/kernel' % m, '%s/score/bias' % m):
            net.variables[key] = net.variables[key] / max(spread, 1e-6)
    net._variables_changed()
    net._get_sufficient_statistic(one)                         # warm-up: buffers, first-use costs
    torch.cuda.synchronize(device)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        S, counts = net._get_sufficient_statistic(data)
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    t0 = time.perf_counter()
    net._fit_sufficient_statistic(S, counts)
    t_host = time.perf_counter() - t0
    t0 = time.perf_counter()
    params = net.fit(data)
    torch.cuda.synchronize(device)
    t_fit = time.perf_counter() - t0
    # the statistics kernel alone: HIP events over 50 launches on one expert's probability map of one batch
    prob = torch.softmax(torch.randn((batch, h, w, C), generator=torch.Generator(device='cpu').manual_seed(6)), -1).to(device)
    S1 = torch.zeros((C, C), dtype=torch.float64, device=device)
    cnt = torch.zeros(C, dtype=torch.int64, device=device)
    for _ in range(3):
        ops.dirichlet_suffstats(prob, one['labels'], S1, cnt)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.dirichlet_suffstats(prob, one['labels'], S1, cnt)
    e1.record()
    torch.cuda.synchronize(device)
    sec = e0.elapsed_time(e1) * 1e-3 / 50
    nbytes = batch * h * w * (4 * C + 4)                        # 12 float32 probabilities + one int32 label per pixel
    flops_img = conv_flops_per_image(h, w, 3) + conv_flops_per_image(h, w, 1)
    rec = {'workload': 'configs[3] fit: DirichletFusion.fit() at %dx%d, %d RGB-D images in batches of %d (two SimpleFCN experts -> '
                       'probabilities -> sufficient statistics -> host Newton fit), U=%d, C=%d' % (w, h, n, batch, U, C),
           'images_per_step': batch, 'dtype': 'bf16', 'unit': 'images/s',
           'value': round(n / best, 2), 'ms_per_step': round(best / nbatches * 1e3, 3),
           'statistics_pass_s': round(best, 4), 'host_newton_fit_s': round(t_host, 4), 'fit_total_s': round(t_fit, 4),
           'fit_images_per_s_incl_host_fit': round(n / t_fit, 2),
           'conv_tflops_end_to_end': round(n * flops_img / best / 1e12, 2),
           'roofline_suffstats': {'bound': 'hbm', 'kernel': 'suffstats_kernel (fusion.hip): one expert, one batch', 'achieved': round(nbytes / sec / 1e9, 1),
                                  'peak': 8000.0, 'unit': 'GB/s', 'frac': round(nbytes / sec / 8e12, 4), 'traffic': None,
                                  'avg_launch_ms': round(sec * 1e3, 4), 'bytes_per_launch': nbytes,
                                  'measured': 'HIP events over 50 launches'},
           'params_finite': bool(all(np.isfinite(np.asarray(params[m])).all() for m in net.modalities)),
           'note': 'random-init experts with the score layer scaled to unit logit spread (a saturated softmax makes the host fit '
                   'run its 10 000 iterations per class: 9.4 s); the host fit is the reference\'s own numpy procedure '
                   '(dirichletDifferentiation.py:129-192 restated) and does not depend on the number of images'}
    del net
    torch.cuda.empty_cache()
    return rec


def make_trainer_net(args, device, expert, joint, batch, batch_norm=None):
    from modular_semantic_segmentation_amd import get_model
    desc = ({'rgb': 'float32', 'labels': 'int32'}, {'rgb': (None, None, 3), 'labels': (None, None)}, C)
    if joint:
        # the joint two-stream model (FusionFCN, [reference default] RMSProp): both trunks + fused decoder with batch norm
        net = get_model('fusion_fcn')({'rgb': 'rgb', 'depth': 'depth'}, {'rgb': 3, 'depth': 1}, U, C, batchsize=batch,
                                      learning_rate=1e-4, trainer='rmsprop', seed=1, device=str(device), sync_loss=False)
        net.variables['depth_conv1_1/kernel'] = net.variables['depth_conv1_1/kernel'] / 256.0
        net._variables_changed()
    elif expert == 'adapnet':
        net = get_model('adapnet')(desc, modality='rgb', num_units=U, batchsize=batch, learning_rate=1e-4,
                                   trainer='adam', seed=1, device=str(device), sync_loss=False)
    else:
        net = get_model('fcn')('rgb', desc, 'rgb', num_units=U,
                               batch_normalization=bool(args.batch_norm if batch_norm is None else batch_norm),
                               batchsize=batch, learning_rate=1e-4, trainer='adam', seed=1, device=str(device),
                               sync_loss=False)
    return net


def measure_training(args, device, world, rank, dist, batch, steps, warmup, expert='fcn', joint=False, h=None, w=None,
                     batch_norm=None):
    """Data-parallel expert training: every rank differentiates its own images, gradients are all-reduced in buckets on
    a side stream during backward (RCCL over xGMI).  Returns a record with images/s, the roofline of the MFMA convs
    (forward + data gradient + filter gradient FLOPs over their summed HIP-event times) and, for N > 1, the exposed
    all-reduce time (step time minus the same step with the reducer stubbed out)."""
    from modular_semantic_segmentation_amd import ops
    h, w = h or args.height, w or args.width
    gen = torch.Generator(device='cpu').manual_seed(99 + rank)
    data = {'rgb': torch.randint(0, 256, (batch, h, w, 3), generator=gen).float().to(device),
            'labels': torch.randint(-1, C, (batch, h, w), generator=gen).int().to(device)}
    if joint:
        data['depth'] = torch.randint(0, 65536, (batch, h, w, 1), generator=gen).float().to(device)
    net = make_trainer_net(args, device, expert, joint, batch, batch_norm)

    def timed(n, min_seconds=1.0):
        return timed_blocks(lambda: net._train_batch(data), n, device, world, dist, min_seconds)

    for _ in range(warmup):
        net._train_batch(data)
    times = timed(steps)
    dt = float(np.median(times))
    rec = {'value': round(batch * world * steps / dt, 2), 'unit': 'images/s', 'images_per_gpu_per_step': batch,
           'n_gpus': world}
    rec.update(block_stats(times, steps))
    if world > 1:
        # the same steps without the collective: what the all-reduce costs beyond what backward hides.  The replicas
        # drift apart while the reducer is off, so rank 0's parameters and optimizer slots are broadcast again afterwards.
        from modular_semantic_segmentation_amd.parallel import sync_trainer_from_rank0
        reducer, net._reducer = net._reducer, None
        try:
            net._train_batch(data)
            dt_local = float(np.median(timed(steps, 0.5)))
        finally:
            net._reducer = reducer
        sync_trainer_from_rank0(net.trainer)
        rec['ms_per_step_without_allreduce'] = round(dt_local / steps * 1e3, 3)
        rec['allreduce_ms_exposed'] = round((dt - dt_local) / steps * 1e3, 3)
        rec['gradient_bytes_per_step'] = int(net.trainer.grad.numel() * 4)
    # roofline of the MFMA convs of the step (rank 0's kernels; collectives run on their own stream)
    # (the filter gradients go back onto the compute stream for this pass: a launch timed while another stream's kernel
    # shares the chip would be charged that kernel's time as well -- as the inference pass serialises the two experts)
    from modular_semantic_segmentation_amd import trainer as trainer_mod
    prof = []
    side, trainer_mod._WGRAD_STREAM = trainer_mod._WGRAD_STREAM, False
    ops.CONV_PROFILE = prof
    torch.cuda.synchronize(device)
    try:
        for _ in range(0 if getattr(args, 'no_roofline_pass', False) else min(steps, 5)):
            net._train_batch(data)
        torch.cuda.synchronize(device)
    finally:
        ops.CONV_PROFILE = None
        trainer_mod._WGRAD_STREAM = side
    fl = sum(p[1] for p in prof)
    sec = sum(p[2].elapsed_time(p[3]) for p in prof) * 1e-3
    if getattr(args, 'layer_profile', False) and rank == 0 and prof:
        per = len(prof) // min(steps, 5)
        for i in range(per):
            evs = prof[i::per]
            ms = sum(e0.elapsed_time(e1) for _, _, e0, e1 in evs) / len(evs)
            print('  launch %2d %-9s %7.1f GF %8.1f us %7.0f TF/s' % (i, evs[0][0], evs[0][1] / 1e9, ms * 1e3,
                                                                    evs[0][1] / ms / 1e9), file=sys.stderr)
    if sec > 0:
        by = {}
        for kind, f, e0, e1 in prof:
            d = by.setdefault(kind.split('_')[0] if kind.startswith(('dgrad', 'wgrad')) else 'fwd', [0.0, 0.0])
            d[0] += f
            d[1] += e0.elapsed_time(e1) * 1e-3
        rec['roofline'] = {'bound': 'mfma', 'kernel': 'MFMA convs of the training step: forward + data gradient (conv_dma_kernel / '
                           'conv_mfma_kernel) + filter gradient (conv_wgrad_dma_kernel)', 'achieved': round(fl / sec / 1e12, 2),
                           'peak': PEAK_TFLOPS['bf16'], 'unit': 'TFLOP/s', 'frac': round(fl / sec / 1e12 / PEAK_TFLOPS['bf16'], 4),
                           'traffic': None, 'launches': len(prof),
                           'by_pass_tflops': {k: round(v[0] / v[1] / 1e12, 1) for k, v in by.items() if v[1] > 0},
                           'measured': 'HIP events per launch, every kernel on ONE stream (the step itself runs the filter '
                                       'gradients on a second stream)'}
    flops = 3.0 * conv_flops_per_image(h, w, 3) + (3.0 * conv_flops_per_image(h, w, 1) if joint else 0.0)
    if expert == 'adapnet':
        flops = 3.0 * adapnet_flops_per_image(h, w, 3)
    rec['conv_tflops_end_to_end'] = round(batch * world * steps * flops / dt / 1e12, 2)
    return rec


def bench_train(args, device, world, rank, dist):
    joint = args.fusion == 'joint' and args.expert == 'fcn'
    adap = args.expert == 'adapnet'
    rec = measure_training(args, device, world, rank, dist, args.batch, args.steps, args.warmup, args.expert, joint)
    if rank == 0:
        metric = 'images/sec, SimpleFCN RGB expert training step (fwd + bwd + Adam%s) at %dx%d' % (
            ', batch norm' if args.batch_norm else '', args.width, args.height)
        workload = 'SimpleFCN RGB %dx%d training, U=%d, C=%d, Adam' % (args.width, args.height, U, C)
        if joint:
            metric = 'RGB-D images/sec, fusion_fcn joint model training step (fwd + bwd + RMSProp, decoder batch norm) at %dx%d' % (
                args.width, args.height)
            workload = 'fusion_fcn RGB+Depth %dx%d training, U=%d, C=%d, RMSProp' % (args.width, args.height, U, C)
        if adap:
            metric = 'images/sec, AdapNet RGB expert training step (fwd + bwd + Adam, batch norm everywhere) at %dx%d' % (
                args.width, args.height)
            workload = 'AdapNet RGB %dx%d training, U=%d, C=%d, Adam' % (args.width, args.height, U, C)
        out = {'metric': metric, 'value': rec['value'], 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
               'warmup': args.warmup, 'ms_per_step': rec['ms_per_step'], 'ms_per_step_min': rec['ms_per_step_min'],
               'ms_per_step_max': rec['ms_per_step_max'], 'timed_blocks': rec['timed_blocks'], 'higher_is_better': True,
               'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
               'config': {'workload': workload, 'images_per_gpu_per_step': args.batch, 'global_batch': args.batch * world,
                          'parallelism': 'dp%d, bucketed gradient all-reduce overlapped with backward' % world},
               'conv_tflops_end_to_end': rec['conv_tflops_end_to_end'], 'roofline': rec.get('roofline'),
               'cpu_baseline': None,
               'cpu_baseline_note': 'training has no CPU leg: the oracle differentiates one 64x96 image in seconds (tests), a '
                                    '768x384 step would take minutes; the inference line carries the CPU baseline'}
        for k in ('ms_per_step_without_allreduce', 'allreduce_ms_exposed', 'gradient_bytes_per_step'):
            if k in rec:
                out[k] = rec[k]
        print(json.dumps(out), file=_JSON_OUT, flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch-norm', action='store_true', help='train mode: the batch_normalization=true training graph')
    ap.add_argument('--layer-profile', action='store_true', help='print per-conv-launch times of the roofline pass')
    ap.add_argument('--no-roofline-pass', action='store_true',
                    help='train mode: skip the per-launch HIP-event pass after the timed region (kernel traces of the step alone)')
    ap.add_argument('--batch', type=int, default=16, help='images per GPU per step')
    ap.add_argument('--height', type=int, default=384)
    ap.add_argument('--width', type=int, default=768)
    ap.add_argument('--fusion', default='bayes', choices=['bayes', 'dirichlet', 'joint'],
                    help="'joint' = the fusion_fcn baseline model instead of two experts + probabilistic fusion")
    ap.add_argument('--expert', default='fcn', choices=['fcn', 'adapnet'],
                    help="expert architecture of the two streams (default: the headline SimpleFCN; 'adapnet' = side measurement)")
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp8'],
                    help="'fp8': the block-scaled e4m3 conv path (BASELINE config 5; quote it with --height 1024 --width 2048)")
    ap.add_argument('--mode', default='infer', choices=['infer', 'train'],
                    help="'infer' (headline): two experts + fusion; 'train': one SimpleFCN training step (fwd+bwd+Adam)")
    ap.add_argument('--fp8-start', default=None,
                    help="--dtype fp8: first e4m3 conv, one layer name or rgb=<layer>,depth=<layer> ('bf16': that expert keeps bf16 operands)")
    ap.add_argument('--fp8-deep', action='store_true',
                    help="--dtype fp8: e4m3 operands from conv1_2 on (model config fp8_deep: faster, costs accuracy)")
    ap.add_argument('--record', default=None, choices=['dirichlet_fit', 'dp_regime'],
                    help='print ONE of the extra records alone (N = 1) instead of the headline line')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-accuracy', action='store_true', help='skip the trained-experts accuracy evidence (N = 1 only)')
    ap.add_argument('--accuracy-steps', type=int, default=1500)
    ap.add_argument('--accuracy-images', type=int, default=24)
    ap.add_argument('--no-extra', action='store_true', help='skip the extra configurations / the train_dp record')
    ap.add_argument('--dist-backend', default='nccl', help="'nccl' (= RCCL over xGMI); 'gloo' only for smoke tests")
    ap.add_argument('--share-device', action='store_true',
                    help='smoke test: all ranks use GPU 0 (gloo backend only)')
    ap.add_argument('--no-graph', dest='graph', action='store_false',
                    help='launch every kernel eagerly instead of replaying the step from a captured hipGraph')
    ap.add_argument('--serial-experts', action='store_true',
                    help='run the RGB and depth experts back to back on one stream at every batch size')
    ap.add_argument('--two-streams', action='store_true',
                    help='run the experts side by side on two streams at every batch size (default: the model chooses -- two '
                         'streams for small batches, one from the batch on at which every launch fills the chip)')
    ap.add_argument('--min-seconds', type=float, default=2.0,
                    help='the timed block of --steps steps is repeated until this many seconds of timed work; the median '
                         'block is reported (0 = exactly one block)')
    ap.add_argument('--train-dp-timeout', type=float, default=240.0,
                    help='N > 1: seconds the train_dp phase may take before the headline line is printed without it')
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return launch_ranks(args)
    # stdout carries exactly ONE JSON line: everything the models print (reference-style INFO / WARNING lines) goes to stderr
    # (at the file-descriptor level too: native libraries -- gloo's connection banner, a collective library's debug lines --
    # write to descriptor 1 directly)
    global _JSON_OUT
    sys.stdout.flush()
    try:
        saved = os.dup(1)
        os.dup2(2, 1)
        _JSON_OUT = os.fdopen(saved, 'w')
    except OSError:
        _JSON_OUT = sys.stdout
    sys.stdout = sys.stderr

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    import datetime
    import torch.distributed as dist
    if args.share_device:
        local_rank = 0
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local_rank)
        # a short collective timeout: a rank that fails alone makes the others' next collective RAISE instead of hanging
        tmo = datetime.timedelta(seconds=max(60.0, args.train_dp_timeout))
        if args.dist_backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank), timeout=tmo)
        else:
            dist.init_process_group(args.dist_backend, timeout=tmo)
    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)

    if args.record is not None:
        if args.record == 'dirichlet_fit':
            rec = dirichlet_fit_record(device)
        else:
            recs, ref = dp_regime_records(args, device, dist)
            rec = {'records': recs}
        print(json.dumps(rec), file=_JSON_OUT, flush=True)
        return
    if args.mode == 'train':
        return bench_train(args, device, world, rank, dist)
    default_line = (args.fusion, args.expert, args.dtype, args.height, args.width) == ('bayes', 'fcn', 'bf16', 384, 768)
    fp8_start = args.fp8_start
    if fp8_start and '=' in fp8_start:
        fp8_start = dict(kv.split('=') for kv in fp8_start.split(','))
    net = build_model(device, args.fusion, args.expert, args.batch, args.dtype, fp8_deep=args.fp8_deep, fp8_start=fp8_start)
    batch = synthetic_batch(device, args.batch, args.height, args.width, seed=1234 + rank)
    if args.dtype == 'fp8':
        net.calibrate(batch)
    times, _, prof, times_serial = measure_inference(net, batch, device, args.steps, args.warmup, graph=args.graph,
                                                     serial=True if args.serial_experts else (False if args.two_streams else None),
                                                     world=world, dist=dist,
                                                     min_seconds=args.min_seconds)
    dt = float(np.median(times))            # block times are already the max over ranks
    dt_serial = float(np.median(times_serial))

    if args.layer_profile and rank == 0:
        # per-launch-slot breakdown (launch order repeats every step): flops, mean time, TFLOP/s
        per = len(prof) // max(args.steps * len(times_serial), 1)
        for i in range(per):
            evs = prof[i::per]
            ms = sum(e0.elapsed_time(e1) for _, _, e0, e1 in evs) / len(evs)
            print('  conv launch %2d %s %7.1f GF %8.1f us %7.0f TF/s' % (i, evs[0][0], evs[0][1] / 1e9, ms * 1e3,
                                                                      evs[0][1] / ms / 1e9), file=sys.stderr)
    # dominant kernel: the 3x3 implicit-GEMM MFMA conv (fp8: the launches on e4m3 operands, against the fp8 peak)
    pair = both = None
    if args.dtype == 'fp8':
        roofline = roofline_of(prof, ('k3f8',), PEAK_TFLOPS['fp8'],
                               'conv_f8_dma_kernel (v_mfma_scale_f32_32x32x64_f8f6f4, 3x3 launches on e4m3 operands; conv_mfma_kernel<F8> where a map does not tile in 16x32)',
                               times_serial, args.steps)
    else:
        roofline = roofline_of(prof, ('k3',), PEAK_TFLOPS['bf16'],
                               'conv_dma4_kernel / conv_dma5_kernel (the 3x3 implicit-GEMM MFMA conv: every launch of conv2_1 .. '
                               'conv5_3; conv_dma_kernel / conv_mfma_kernel on shapes that do not tile)', times_serial,
                               args.steps, traffic=committed_traffic(args.batch, args.height, args.width))
        # conv1_1 + conv1_2 run as ONE kernel (conv_first_fused.hip): reported beside the dominant kernel, with the algorithmic
        # FLOPs of both layers against the whole kernel's time, and both together so that nothing is left out
        pair = roofline_of(prof, ('k3pair',), PEAK_TFLOPS['bf16'], 'conv_first_pair_kernel (conv1_1 + conv1_2 + pool1 fused; '
                           'algorithmic FLOPs of both layers)', times_serial, args.steps)
        both = roofline_of(prof, ('k3', 'k3pair'), PEAK_TFLOPS['bf16'], 'all of the above', times_serial, args.steps)
    per_image = conv_flops_per_image if args.expert == 'fcn' else adapnet_flops_per_image
    flops_img = per_image(args.height, args.width, 3) + per_image(args.height, args.width, 1)
    if args.expert == 'adapnet':
        # no single dominant kernel: the whole serialised expert step against the algorithmic conv FLOPs
        achieved = args.batch * args.steps * flops_img / dt_serial / 1e12
        roofline = {'bound': 'mfma', 'kernel': 'whole AdapNet step, every kernel (algorithmic conv FLOPs / step time)',
                    'achieved': round(achieved, 2), 'peak': PEAK_TFLOPS['bf16'], 'unit': 'TFLOP/s',
                    'frac': round(achieved / PEAK_TFLOPS['bf16'], 4), 'traffic': None,
                    'gflop_per_image_pair': round(flops_img / 1e9, 2),
                    'measured': 'wall clock of the serialised eager steps (%.3f ms/step)' % (dt_serial / args.steps * 1e3)}

    # ---- the headline record: complete BEFORE anything below can fail or hang --------------------------------------------
    images = args.batch * world * args.steps
    res = {
        'metric': 'images/sec at %dx%d RGB-D FCN (two SimpleFCN experts + %s fusion + argmax, inference)' % (
            args.width, args.height, args.fusion),
        'value': round(images / dt, 2), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
        'config': {'workload': 'two-stream SimpleFCN RGB+Depth %dx%d + %s fusion, U=%d, C=%d, random-init weights'
                               % (args.width, args.height, args.fusion, U, C),
                   'images_per_gpu_per_step': args.batch, 'global_batch': args.batch * world,
                   'expert_streams': 1 if args.serial_experts else (2 if args.two_streams else 'auto: 2 for batches whose launches leave CUs idle, else 1'), 'hip_graph': bool(getattr(net, '_bench_graph_used', args.graph)),
                   'parallelism': 'dp%d (batch sharding, no data-path collective)' % world},
        'conv_tflops_end_to_end': round(images * flops_img / dt / 1e12, 2),
        'roofline': roofline,
    }
    if pair is not None:
        res['roofline_first_pair'] = pair
        res['roofline_all_3x3'] = both
    res.update(block_stats(times, args.steps))
    res['timing'] = ('blocks of exactly --steps steps, barrier + synchronize on both sides, max over ranks, repeated until '
                     '%.1f s; value and ms_per_step are the MEDIAN block' % args.min_seconds)
    if args.dtype == 'fp8':
        res['config']['conv_dtype'] = ('e4m3 operands from conv1_2 on (fp8_deep), conv1_1 fp32 -> e4m3' if args.fp8_deep else
                                       'e4m3 operands from conv2_2 on, conv1_1 fp32, conv1_2 bf16, conv2_1 bf16 -> e4m3')
    if args.expert == 'adapnet':
        res['metric'] = 'images/sec at %dx%d RGB-D, two AdapNet experts + %s fusion + argmax (inference)' % (
            args.width, args.height, args.fusion)
        res['config']['workload'] = 'two-stream AdapNet RGB+Depth %dx%d + %s fusion, U=%d, C=%d, random-init weights' % (
            args.width, args.height, args.fusion, U, C)
    if args.fusion == 'joint':
        res['metric'] = 'images/sec at %dx%d, fusion_fcn joint RGB-D baseline (inference)' % (args.width, args.height)
        res['config']['workload'] = 'fusion_fcn (two VGG16 trunks + fused decoder) RGB+Depth %dx%d, U=%d, C=%d' % (
            args.width, args.height, U, C)
    res['cpu_baseline'] = None

    printed = threading.Lock()

    def emit():
        """Print the ONE line (rank 0), exactly once whoever gets here first: the normal end or the watchdog."""
        if rank == 0 and printed.acquire(blocking=False):
            print(json.dumps(res), file=_JSON_OUT, flush=True)

    # ---- records beyond the headline ---------------------------------------------------------------------------------
    extra = []
    if world == 1 and rank == 0:
        variables = dict(net.variables)
        del net
        torch.cuda.empty_cache()
        if not args.no_cpu_baseline and args.fusion != 'joint' and args.expert == 'fcn':
            cores, avail = pick_cpu_threads(args.height, args.width)
            g = np.load(os.path.join(ROOT, 'tests', 'golden', 'notebook_868.npz'))
            cpu = None
            if default_line and not args.no_accuracy:
                try:
                    res['accuracy'], cpu = accuracy_and_cpu_baseline(args, device, cores, avail)
                except Exception as exc:       # noqa: BLE001  (the headline and its CPU baseline must still be printed)
                    res['accuracy'] = {'error': '%s: %s' % (type(exc).__name__, exc)}
            if cpu is None:
                cpu = cpu_baseline_random(args, variables, {'rgb': g['cm_rgb'], 'depth': g['cm_depth']}, cores, avail)
            res['cpu_baseline'] = cpu

        def guarded(fn, *a, **kw):
            try:
                extra.append(fn(*a, **kw))
            except Exception as exc:           # noqa: BLE001  (an extra record never costs the headline line)
                extra.append({'workload': a[1] if len(a) > 1 and isinstance(a[1], str) else fn.__name__,
                              'error': '%s: %s' % (type(exc).__name__, exc)})

        if default_line and not args.no_extra:
            guarded(extra_inference, device, 'experiments/timing.py protocol: two SimpleFCN experts + Bayes fusion, batch 1, '
                                         'tf.ones([1,768,384,.]) input, label map fetched to the host every iteration as the '
                                         'product fetches it: one byte per pixel into pinned memory '
                                         '(Inference Time.ipynb:139 publishes 0.0461 s on a GTX 1080 Ti)', 'bayes', 1, 768, 384,
                                         steps=50, warmup=5, ones=True, fetch='bytes')
            guarded(extra_inference, device, 'the same protocol, the bytes widened to the reference\'s int64 on the host inside '
                                         'the timed iteration', 'bayes', 1, 768, 384, steps=50, warmup=5, ones=True, fetch='int64')
            guarded(extra_inference, device, 'the same protocol as rounds 3-4 measured it: `.cpu()` of the int64 label map '
                                         '(2.4 MB into pageable memory)', 'bayes', 1, 768, 384, steps=50, warmup=5, ones=True,
                                         fetch='pageable')
            guarded(extra_inference, device, 'two-stream SimpleFCN + Bayes fusion 768x384, batch 1, random input', 'bayes', 1,
                                         384, 768, steps=30, warmup=3)
            guarded(extra_inference, device, 'two-stream SimpleFCN + Bayes fusion 768x384, batch 1, random input, streamk=True',
                                         'bayes', 1, 384, 768, steps=30, warmup=3, streamk=True)
            guarded(extra_inference, device, "conv_dtype='fp32', the label-exact mode (argmax maps equal to the fp32 graph's): two "
                                             'SimpleFCN experts + Bayes fusion 768x384 x 16, fp32 matrix instruction', 'bayes', 16,
                                         384, 768, dtype='fp32', steps=5, warmup=1)
            guarded(extra_inference, device, "the same on round 4's vector-ALU kernel (XV_EXACT_SCALAR=1): the A/B baseline of the "
                                             'record above', 'bayes', 4, 384, 768, dtype='fp32', steps=2, warmup=1, scalar_f32=True)
            guarded(host_boundary_record, device)
            guarded(extra_inference, device, 'two-stream SimpleFCN + Dirichlet fusion 768x384 (configs[3] inference side)',
                                         'dirichlet', 16, 384, 768)
            guarded(extra_inference, device, 'DIAGNOSTIC, not a workload: the headline step with all-zero weights, biases and inputs '
                                             '(no MFMA operand toggles: what the same kernels reach when power does not hold the '
                                             'clock down)', 'bayes', 16, 384, 768, zero_operands=True)
            guarded(extra_inference, device, 'BayesFusion of RGB+Depth FCN experts 1024x512 (configs[2])', 'bayes', 8, 512, 1024)
            guarded(extra_inference, device, 'two-stream SimpleFCN + Bayes fusion 2048x1024 bf16', 'bayes', 4, 1024, 2048,
                                         steps=5)
            guarded(extra_inference, device, 'VGG-16-encoder FCN experts 2048x1024, fp8 MFMA conv path (configs[4])', 'bayes',
                                         4, 1024, 2048, dtype='fp8', steps=5)
            guarded(extra_inference, device, 'the same with fp8_deep=True (e4m3 operands from conv1_2 on: faster, and it costs '
                                             'accuracy -- DESIGN.md, the fp8 plan)', 'bayes', 4, 1024, 2048, dtype='fp8', steps=5,
                                         fp8_deep=True)
            plan = None
            try:
                plan = {m: r['chosen'] for m, r in res['accuracy']['fp8']['plan']['bayes'].items()}
            except (KeyError, TypeError):
                pass
            if plan is not None:
                guarded(extra_inference, device, 'configs[4] at north_star\'s accuracy bound: 2048x1024 x 4, conv_dtype=fp8 with the '
                                                 'ACCURACY-GUARDED plan calibrate() chose per expert on the trained experts of the '
                                                 'accuracy record (label agreement with the bf16 graph >= 0.995 on the calibration '
                                                 'batch; an expert no e4m3 plan serves keeps bf16 operands): %s'
                                                 % json.dumps(plan, sort_keys=True), 'bayes', 4, 1024, 2048, dtype='fp8', steps=5,
                                             fp8_start=plan)

            def training_record():
                tr = measure_training(args, device, 1, 0, dist, 16, 8, 2)
                tr['workload'] = 'SimpleFCN RGB expert training step 768x384 (fwd + bwd + Adam), 16 images'
                return tr
            guarded(training_record)

            def regime():
                recs, ref = dp_regime_records(args, device, dist)
                base = {False: next((e.get('value') for e in extra if str(e.get('workload', '')).startswith(
                    'SimpleFCN RGB expert training step 768x384 (fwd + bwd + Adam), 16 images')), None), True: ref.get((True, 16))}
                for r in recs:
                    bn = 'batch_normalization' in r['workload']
                    if r['images_per_gpu_per_step'] == 4 and base[bn]:
                        r['images_per_s_relative_to_the_16_image_step'] = round(r['value'] / base[bn], 4)
                extra.extend(recs[:-1])
                return recs[-1]
            guarded(regime)
            guarded(dirichlet_fit_record, device)
        if extra:
            res['extra'] = extra
    elif world > 1 and not args.no_extra and default_line:
        del net
        torch.cuda.empty_cache()
        # The data-parallel TRAINING step: the collective path an inference scaling curve never touches.  The headline
        # record above is already complete and must survive whatever happens here.  A rank that fails alone leaves the
        # others inside a collective: the process group's timeout turns that into an exception, and a watchdog on EVERY
        # rank ends the process after --train-dp-timeout seconds -- rank 0 prints the headline with the error noted --
        # instead of waiting for the collective library's own watchdog to abort it with the line lost.  (No re-exec: a
        # process that has touched the GPU is never replaced.)
        def watchdog():
            res['train_dp'] = {'error': 'timeout: the train_dp phase did not finish within %.0f s' % args.train_dp_timeout}
            emit()
            os._exit(3)         # a rank that gives up says so in its exit code (the record carries the reason)
        timer = threading.Timer(args.train_dp_timeout, watchdog)
        timer.daemon = True
        timer.start()
        from modular_semantic_segmentation_amd.parallel import agree_any
        failure = None
        try:
            train_dp = measure_training(args, device, world, rank, dist, 8, 8, 2)
            train_dp['workload'] = ('SimpleFCN RGB expert training 768x384, 8 images per GPU per step, dp%d: three gradient '
                                    'buckets all-reduced over RCCL on a side stream during backward' % world)
        except Exception as exc:       # noqa: BLE001
            failure = '%s: %s' % (type(exc).__name__, exc)
        try:
            # agree on the outcome collectively: a record is reported only if EVERY rank finished the phase
            if agree_any(failure is not None, device):
                train_dp = {'error': failure or 'another rank failed (see its stderr)'}
        except Exception as exc:       # noqa: BLE001
            train_dp = {'error': failure or '%s: %s' % (type(exc).__name__, exc)}
        timer.cancel()
        res['train_dp'] = train_dp

    emit()
    if world > 1:
        try:
            dist.destroy_process_group()
        except Exception:           # noqa: BLE001  (after a failed collective the group may already be unusable)
            pass


def sysfs_gpu_count():
    """GPUs of this node as the kernel driver lists them (KFD topology nodes with SIMDs; CPUs are nodes with none), read from
    sysfs: no HIP / HSA / amdsmi call, so nothing here opens /dev/kfd.  None when the topology is not readable (the ranks
    then find out themselves).  HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES lists narrow the count."""
    import glob
    nodes = glob.glob('/sys/class/kfd/kfd/topology/nodes/*/properties')
    if not nodes:
        return None
    count = 0
    for path in nodes:
        try:
            with open(path) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
        except OSError:
            return None
        if int(props.get('simd_count', '0')) > 0:
            count += 1
    for var in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES'):
        listed = os.environ.get(var)
        if listed is not None and listed.strip() != '':
            count = min(count, len([t for t in listed.split(',') if t.strip() != '']))
    return count


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks as a CHILD process (the same command
    line under torch.distributed.run, one rank per GPU over RCCL), relay rank 0's JSON line and exit with the child's
    code.  Nothing here touches the GPU: a process that has initialised it must never exec another program, so the
    ranks are children and this parent only waits.  The parent makes NO torch.cuda call at all -- torch.cuda.device_count()
    falls through to hipGetDeviceCount (which opens /dev/kfd) whenever its amdsmi probe fails -- and counts the GPUs from
    sysfs instead (tests/test_host_logic.py::test_bench_launcher_parent_never_touches_the_gpu)."""
    import socket
    have = None if args.share_device else sysfs_gpu_count()
    if have is not None and have < args.gpus:
        print('bench.py: --gpus %d but %d GPU(s) listed in the KFD topology (use --share-device --dist-backend gloo for a '
              'one-GPU smoke run of the multi-rank path)' % (args.gpus, have), file=sys.stderr)
        sys.exit(2)
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:
        out = out.strip()
        if out.startswith('{') and out.endswith('}'):
            line = out                  # rank 0's record (the last such line wins)
        elif out:
            print(out, file=sys.stderr)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    sys.exit(rc if rc != 0 or line is not None else 1)


if __name__ == '__main__':
    main()
