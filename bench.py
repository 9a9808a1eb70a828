#!/usr/bin/env python3
"""Headline benchmark: images/sec of the two-stream (RGB + depth) SimpleFCN experts + Bayes
fusion + argmax at 768x384 on MI355X (BASELINE.json configs[1]); protocol of the reference's
experiments/timing.py (inputs resident on the device, wall clock around the whole pipeline).

  python bench.py [--gpus N --steps K --warmup W] [--batch B] [--height 384 --width 768]
                  [--fusion bayes|dirichlet|joint] [--no-cpu-baseline]
N > 1 is launched by the driver through torch.distributed.run (one rank per GPU); images are
independent, so ranks shard the batch with no data-path collective ("weak" scaling) and only the
timing is reduced (MAX over ranks).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA, MI355X_MICROARCH.md chip table
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


def build_model(args, device):
    from modular_semantic_segmentation_amd import get_model
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'notebook_868.npz'))
    desc = ({'rgb': 'float32', 'depth': 'float32', 'labels': 'int32'},
            {'rgb': (None, None, 3), 'depth': (None, None, 1), 'labels': (None, None)}, C)
    common = dict(data_description=desc, num_units=U, num_channels={'rgb': 3, 'depth': 1}, expert_model=args.expert,
                  class_prior='data', batchsize=args.batch, seed=1, device=str(device))
    if args.fusion == 'joint':
        # the reference's joint baseline fusion_fcn (experiments/timing.py:24-45): two VGG16 trunks + fused decoder
        net = get_model('fusion_fcn')({'rgb': 'rgb', 'depth': 'depth'}, {'rgb': 3, 'depth': 1}, U, C,
                                      batchsize=args.batch, seed=1, device=str(device))
        net.variables['depth_conv1_1/kernel'] = net.variables['depth_conv1_1/kernel'] / 256.0
        net._variables_changed()
        return net
    if args.fusion == 'bayes':
        net = get_model('bayes_fusion')(confusion_matrices={'rgb': g['cm_rgb'], 'depth': g['cm_depth']},
                                        prefixes={'rgb': 'rgb', 'depth': 'depth'}, **common)
    else:
        rng = np.random.default_rng(2)
        params = {'rgb': rng.uniform(0.5, 4.0, (C, C)), 'depth': rng.uniform(0.5, 4.0, (C, C)),
                  'class_counts': g['cm_depth'].sum(1)}
        net = get_model('dirichlet_fusion')(dirichlet_params=params, modalities=['rgb', 'depth'], sigma=1.0,
                                            delta=1e-2, beta=1e-2, **common)
    # a trained depth expert absorbs the raw uint16 range in conv1_1; random init needs the scale
    first = 'depth/conv1_1/kernel' if args.expert == 'fcn' else 'depth/block_0_1/kernel'
    net.variables[first] = net.variables[first] / 256.0
    net._variables_changed()
    return net


def cpu_baseline(args, variables, cms):
    """The oracle (op-for-op fp32 restatement of the reference graph) timed on this host's cores
    on a bounded sample of the same workload: whole 768x384 RGB-D images, one at a time."""
    from oracle import fcn_oracle as fo
    from oracle import fusion_oracle as fu
    avail = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    # pick the thread count that runs a mid-network conv fastest (all logical CPUs is not it on a big host)
    probe_x = torch.randn(1, 256, args.height // 4, args.width // 4)
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
    cores = best[1]
    torch.set_num_threads(cores)
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


def bench_train(args, device, world, rank, dist):
    """Data-parallel expert training: every rank differentiates its own images, gradients are
    all-reduced in three buckets on a side stream during backward (RCCL over xGMI)."""
    from modular_semantic_segmentation_amd import get_model
    desc = ({'rgb': 'float32', 'labels': 'int32'}, {'rgb': (None, None, 3), 'labels': (None, None)}, C)
    joint = args.fusion == 'joint' and args.expert == 'fcn'
    adap = args.expert == 'adapnet'
    gen = torch.Generator(device='cpu').manual_seed(99 + rank)
    rgb = torch.randint(0, 256, (args.batch, args.height, args.width, 3), generator=gen).float().to(device)
    labels = torch.randint(-1, C, (args.batch, args.height, args.width), generator=gen).int().to(device)
    batch = {'rgb': rgb, 'labels': labels}
    if joint:
        # the joint two-stream model (FusionFCN, [reference default] RMSProp): both trunks + fused decoder with batch norm
        net = get_model('fusion_fcn')({'rgb': 'rgb', 'depth': 'depth'}, {'rgb': 3, 'depth': 1}, U, C, batchsize=args.batch,
                                      learning_rate=1e-4, trainer='rmsprop', seed=1, device=str(device), sync_loss=False)
        net.variables['depth_conv1_1/kernel'] = net.variables['depth_conv1_1/kernel'] / 256.0
        net._variables_changed()
        batch['depth'] = torch.randint(0, 65536, (args.batch, args.height, args.width, 1), generator=gen).float().to(device)
    elif adap:
        net = get_model('adapnet')(desc, modality='rgb', num_units=U, batchsize=args.batch, learning_rate=1e-4,
                                   trainer='adam', seed=1, device=str(device), sync_loss=False)
    else:
        net = get_model('fcn')('rgb', desc, 'rgb', num_units=U, batch_normalization=bool(args.batch_norm),
                               batchsize=args.batch, learning_rate=1e-4, trainer='adam', seed=1, device=str(device),
                               sync_loss=False)
    for _ in range(args.warmup):
        net._train_batch(batch)

    def fence():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        net._train_batch(batch)
    fence()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    if rank == 0:
        images = args.batch * world * args.steps
        flops = 3.0 * conv_flops_per_image(args.height, args.width, 3)
        if joint:
            flops += 3.0 * conv_flops_per_image(args.height, args.width, 1)
        metric = 'images/sec, SimpleFCN RGB expert training step (fwd + bwd + Adam%s) at %dx%d' % (
            ', batch norm' if args.batch_norm else '', args.width, args.height)
        workload = 'SimpleFCN RGB %dx%d training, U=%d, C=%d, Adam' % (args.width, args.height, U, C)
        if joint:
            metric = 'RGB-D images/sec, fusion_fcn joint model training step (fwd + bwd + RMSProp, decoder batch norm) at %dx%d' % (
                args.width, args.height)
            workload = 'fusion_fcn RGB+Depth %dx%d training, U=%d, C=%d, RMSProp' % (args.width, args.height, U, C)
        if adap:
            flops = 3.0 * adapnet_flops_per_image(args.height, args.width, 3)
            metric = 'images/sec, AdapNet RGB expert training step (fwd + bwd + Adam, batch norm everywhere) at %dx%d' % (
                args.width, args.height)
            workload = 'AdapNet RGB %dx%d training, U=%d, C=%d, Adam' % (args.width, args.height, U, C)
        print(json.dumps({
            'metric': metric,
            'value': round(images / dt, 2), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': workload,
                       'images_per_gpu_per_step': args.batch, 'global_batch': args.batch * world,
                       'parallelism': 'dp%d, bucketed gradient all-reduce overlapped with backward' % world},
            'conv_tflops_end_to_end': round(images * flops / dt / 1e12, 2)}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch-norm', action='store_true', help='train mode: the batch_normalization=true training graph')
    ap.add_argument('--layer-profile', action='store_true', help='print per-conv-launch times of the roofline pass')
    ap.add_argument('--batch', type=int, default=16, help='images per GPU per step')
    ap.add_argument('--height', type=int, default=384)
    ap.add_argument('--width', type=int, default=768)
    ap.add_argument('--fusion', default='bayes', choices=['bayes', 'dirichlet', 'joint'],
                    help="'joint' = the fusion_fcn baseline model instead of two experts + probabilistic fusion")
    ap.add_argument('--expert', default='fcn', choices=['fcn', 'adapnet'],
                    help="expert architecture of the two streams (default: the headline SimpleFCN; 'adapnet' = side measurement)")
    ap.add_argument('--mode', default='infer', choices=['infer', 'train'],
                    help="'infer' (headline): two experts + fusion; 'train': one SimpleFCN training step (fwd+bwd+Adam)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--dist-backend', default='nccl', help="'nccl' (= RCCL over xGMI); 'gloo' only for smoke tests")
    ap.add_argument('--share-device', action='store_true',
                    help='smoke test: all ranks use GPU 0 (gloo backend only)')
    ap.add_argument('--no-graph', dest='graph', action='store_false',
                    help='launch every kernel eagerly instead of replaying the step from a captured hipGraph')
    ap.add_argument('--serial-experts', action='store_true',
                    help='run the RGB and depth experts back to back on one stream (profiling: per-kernel times)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    import torch.distributed as dist
    if args.share_device:
        local_rank = 0
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local_rank)
        if args.dist_backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(args.dist_backend)
    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)

    from modular_semantic_segmentation_amd import ops
    if args.mode == 'train':
        return bench_train(args, device, world, rank, dist)
    net = build_model(args, device)
    gen = torch.Generator(device='cpu').manual_seed(1234 + rank)
    rgb = torch.randint(0, 256, (args.batch, args.height, args.width, 3), generator=gen).float().to(device)
    depth = torch.randint(0, 65536, (args.batch, args.height, args.width, 1), generator=gen).float().to(device)
    batch = {'rgb': rgb, 'depth': depth}

    net.concurrent_experts = not args.serial_experts

    def step():
        return net._predict_batch(batch)

    for _ in range(args.warmup):
        step()
    if args.graph:
        net.capture_graph(batch)
        graph = net._graph[0]

        def step():             # noqa: F811  (inputs already sit in the graph's static buffers)
            graph.replay()
            return net._graph[2]
        step()

    def fence():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0

    # Per-kernel roofline pass: the same steps with the two experts serialised on one stream, so that
    # a HIP-event pair around a conv launch (recorded on the launch stream) times that kernel alone
    # (in the timed region above the RGB and depth experts overlap on two streams).
    prof = []
    net._graph = None
    step = lambda: net._predict_batch(batch)        # noqa: E731  (the roofline pass is always eager)
    net.concurrent_experts = False
    step()
    ops.CONV_PROFILE = prof
    torch.cuda.synchronize(device)
    ts = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(device)
    dt_serial = time.perf_counter() - ts
    ops.CONV_PROFILE = None
    net.concurrent_experts = not args.serial_experts
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # dominant kernel: the 3x3 implicit-GEMM MFMA conv with Cout % 128 == 0 (conv2_1 .. conv5_3)
    kinds = {}
    for kind, flops, e0, e1 in prof:
        d = kinds.setdefault(kind, [0.0, 0.0, 0])
        d[0] += flops
        d[1] += e0.elapsed_time(e1) * 1e-3
        d[2] += 1
    if args.layer_profile and rank == 0:
        # per-launch-slot breakdown (launch order repeats every step): flops, mean time, TFLOP/s
        per = len(prof) // max(args.steps, 1)
        for i in range(per):
            evs = prof[i::per]
            ms = sum(e0.elapsed_time(e1) for _, _, e0, e1 in evs) / len(evs)
            print('  conv launch %2d %s %7.1f GF %8.1f us %7.0f TF/s' % (i, evs[0][0], evs[0][1] / 1e9, ms * 1e3,
                                                                      evs[0][1] / ms / 1e9), file=sys.stderr)
    dom = 'k3'
    roofline = None
    traffic = None
    tfiles = sorted(f for f in os.listdir(os.path.join(ROOT, 'profiles')) if f.endswith('_conv_traffic.json')) \
        if os.path.isdir(os.path.join(ROOT, 'profiles')) else []
    if tfiles and (args.height, args.width) == (384, 768):
        # HBM bytes per conv launch from the committed rocprofv3 FETCH_SIZE / WRITE_SIZE passes of this
        # same command (tools/pmc_summary.py); PMC counters cannot be read from inside the process
        tj = json.load(open(os.path.join(ROOT, 'profiles', tfiles[-1])))
        import hashlib
        src = os.path.join(ROOT, 'modular_semantic_segmentation_amd', 'csrc', 'conv_mfma.hip')
        sha = hashlib.sha256(open(src, 'rb').read()).hexdigest()[:16]
        # the counters were collected on ONE version of the kernel: a summary of another version is not reported
        if tj.get('batch') == args.batch and tj.get('kernel_source_sha256_16') == sha:
            traffic = tj.get('hbm_bytes_per_launch')
    if dom in kinds and kinds[dom][1] > 0:
        fl, sec, cnt = kinds[dom]
        achieved = fl / sec / 1e12
        roofline = {'bound': 'mfma', 'kernel': 'conv_dma_kernel / conv_mfma_kernel (3x3 implicit GEMM, all launches)',
                    'achieved': round(achieved, 2),
                    'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(achieved / PEAK_BF16_TFLOPS, 4),
                    'traffic': traffic, 'launches': cnt, 'avg_launch_ms': round(sec / cnt * 1e3, 4),
                    'gflop_per_launch': round(fl / cnt / 1e9, 2),
                    'measured': 'HIP events per launch, experts serialised on one stream (%.3f ms/step)'
                                % (dt_serial / args.steps * 1e3)}

    per_image = conv_flops_per_image if args.expert == 'fcn' else adapnet_flops_per_image
    flops_img = per_image(args.height, args.width, 3) + per_image(args.height, args.width, 1)
    if args.expert == 'adapnet':
        # no single dominant kernel: the whole serialised expert step against the algorithmic conv FLOPs
        achieved = args.batch * args.steps * flops_img / dt_serial / 1e12
        roofline = {'bound': 'mfma', 'kernel': 'whole AdapNet step, every kernel (algorithmic conv FLOPs / step time)',
                    'achieved': round(achieved, 2), 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': round(achieved / PEAK_BF16_TFLOPS, 4), 'traffic': None,
                    'gflop_per_image_pair': round(flops_img / 1e9, 2),
                    'measured': 'wall clock of the serialised eager steps (%.3f ms/step)' % (dt_serial / args.steps * 1e3)}
    if rank == 0:
        images = args.batch * world * args.steps
        res = {
            'metric': 'images/sec at %dx%d RGB-D FCN (two SimpleFCN experts + %s fusion + argmax, inference)' % (
                args.width, args.height, args.fusion),
            'value': round(images / dt, 2), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': 'two-stream SimpleFCN RGB+Depth %dx%d + %s fusion, U=%d, C=%d, random-init weights'
                                   % (args.width, args.height, args.fusion, U, C),
                       'images_per_gpu_per_step': args.batch, 'global_batch': args.batch * world,
                       'expert_streams': 1 if args.serial_experts else 2, 'hip_graph': bool(args.graph),
                       'parallelism': 'dp%d (batch sharding, no data-path collective)' % world},
            'conv_tflops_end_to_end': round(images * flops_img / dt / 1e12, 2),
            'roofline': roofline,
        }
        if args.expert == 'adapnet':
            res['metric'] = 'images/sec at %dx%d RGB-D, two AdapNet experts + %s fusion + argmax (inference)' % (
                args.width, args.height, args.fusion)
            res['config']['workload'] = 'two-stream AdapNet RGB+Depth %dx%d + %s fusion, U=%d, C=%d, random-init weights' % (
                args.width, args.height, args.fusion, U, C)
        if args.fusion == 'joint':
            res['metric'] = 'images/sec at %dx%d, fusion_fcn joint RGB-D baseline (inference)' % (args.width, args.height)
            res['config']['workload'] = 'fusion_fcn (two VGG16 trunks + fused decoder) RGB+Depth %dx%d, U=%d, C=%d' % (
                args.width, args.height, U, C)
        if world == 1 and not args.no_cpu_baseline and args.fusion != 'joint' and args.expert == 'fcn':
            g = np.load(os.path.join(ROOT, 'tests', 'golden', 'notebook_868.npz'))
            res['cpu_baseline'] = cpu_baseline(args, net.variables, {'rgb': g['cm_rgb'], 'depth': g['cm_depth']})
        else:
            res['cpu_baseline'] = None
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
