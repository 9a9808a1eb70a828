"""The host boundary of fit / predict / score as a pipeline (reference: xview/models/base_model.py:203-206,265-331 -- a
tf.data pipeline with 10 parallel map calls and a prefetch in front of every sess.run, whose fetch hands numpy arrays back).

The callers of the reference hand HOST arrays to predict() / score() / fit() and get host arrays back.  A 16-image RGB-D batch
at 768x384 is 75.5 MB in and 37.7 MB of int64 labels out around 4.4 ms of kernels, so a serial `pageable copy -> kernels ->
.cpu()` loop runs at a quarter of the resident-input rate.  Here the three stages overlap:

  stage-in   worker threads copy batch i+2 from the caller's (pageable) arrays into a ring of PINNED staging buffers
             (numpy copies release the GIL; the dtype conversion to float32 / int32 happens in the same pass)
  H2D        batch i+1 goes pinned -> HBM on a copy stream (its own DMA engine), an event per ring slot
  compute    batch i runs on the caller's stream, which waits for that event only
  D2H        the label map of batch i-1 leaves on a second copy stream into a pinned ring slot
  collect    a worker thread moves it into the result array while the GPU computes

Ring slots are reused behind events (device side) and futures (host side); nothing here launches a kernel of its own except
device-to-device copies (hipMemcpyAsync).  XV_HOST_PIPELINE=0 restores the serial path (A/B timing and the equality test)."""
import os
from collections import deque
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

ENABLED = os.environ.get('XV_HOST_PIPELINE', '1') != '0'
DEPTH = 4                       # input ring slots: two being staged, one uploaded ahead, one being consumed
_POOL = None
TRACE = None                    # tools/trace_pipeline.py: a dict collecting seconds the calling thread spent blocked, by cause


def _pool():
    """Stage-in / collect threads, shared by every model of the process."""
    global _POOL
    if _POOL is None:
        avail = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 2)
        _POOL = ThreadPoolExecutor(max_workers=max(2, min(8, avail)), thread_name_prefix='xv-host')
    return _POOL


_COLLECT_POOL = None


def _collect_pool():
    """The collect tasks of ResultFetcher wait for the sub-copies they hand to _pool(): they run on an executor of their own,
    so that however many predict() / score() calls are in flight on other threads (each keeps up to three collects alive) no
    collect can hold a worker of the pool its own sub-copies are queued on (three concurrent calls could fill all
    eight workers with collects waiting for sub-tasks that never got a worker)."""
    global _COLLECT_POOL
    if _COLLECT_POOL is None:
        _COLLECT_POOL = ThreadPoolExecutor(max_workers=4, thread_name_prefix='xv-collect')
    return _COLLECT_POOL


class _blocked(object):
    """with _blocked('cause'): ...  -- accumulates wall time into TRACE when tracing is on"""

    def __init__(self, cause):
        self.cause = cause

    def __enter__(self):
        if TRACE is not None:
            import time
            self.t0 = time.perf_counter()

    def __exit__(self, *exc):
        if TRACE is not None:
            import time
            TRACE[self.cause] = TRACE.get(self.cause, 0.0) + time.perf_counter() - self.t0


def _copy_rows(dst, src, lo, hi):
    np.copyto(dst[lo:hi], src[lo:hi], casting='unsafe')


def _parallel_copy(dst, src, n):
    """dst[:n] = src[:n] (numpy, any dtypes) split by rows over the pool; returns the futures."""
    pool = _pool()
    workers = pool._max_workers
    pieces = max(1, min(n, workers // 2))
    step = (n + pieces - 1) // pieces
    return [pool.submit(_copy_rows, dst, src, lo, min(n, lo + step)) for lo in range(0, n, step)]


class _InSlot(object):
    def __init__(self):
        self.pinned, self.dev = {}, {}
        self.h2d_done = torch.cuda.Event()
        self.free = torch.cuda.Event()          # recorded on the consumer's stream once it has enqueued its reads
        self.used = False

    def buffers(self, key, shape, dtype, device):
        p = self.pinned.get(key)
        if p is None or tuple(p.shape[1:]) != tuple(shape[1:]) or p.shape[0] < shape[0] or p.dtype != dtype:
            self.pinned[key] = torch.empty(tuple(shape), dtype=dtype, pin_memory=True)
            self.dev[key] = torch.empty(tuple(shape), dtype=dtype, device=device)
        return self.pinned[key], self.dev[key]


class DevicePrefetcher(object):
    """Iterate host batches (dicts of numpy arrays / torch tensors with a leading sample axis) as dicts of HBM-resident
    tensors of `dtypes[key]`, staged and uploaded ahead of the consumer.  Keys without an entry in `dtypes` are dropped;
    values that already live on the device are passed through (converted in place on the consumer's stream).

    The yielded tensors are views of ring buffers: they stay valid until the consumer asks for the batch after the next
    one (DEPTH - 2 batches later the slot is overwritten), which is the lifetime every caller here needs -- a training or
    inference step consumes its batch before it asks for another."""

    def __init__(self, device, batches, dtypes, depth=DEPTH):
        self.device, self.batches, self.dtypes, self.depth = torch.device(device), batches, dtypes, depth
        self.slots = [_InSlot() for _ in range(depth)]
        self.copy_stream = torch.cuda.Stream(device=self.device)

    def _stage(self, slot, batch):
        """-> (n, {key: future list}, {key: device tensor passed through})"""
        if slot.used:
            with _blocked('stage: previous upload of the slot'):
                slot.h2d_done.synchronize()          # the upload that read this slot's pinned buffers has finished
        staged, through, n = {}, {}, None
        for key, dtype in self.dtypes.items():
            if key not in batch:
                continue
            v = batch[key]
            n = len(v) if n is None else n
            if isinstance(v, torch.Tensor) and v.is_cuda:
                through[key] = v
                continue
            src = v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            pinned, _ = slot.buffers(key, src.shape, dtype, self.device)
            staged[key] = _parallel_copy(pinned.numpy(), src, len(src))
        return n, staged, through

    def __iter__(self):
        it = iter(self.batches)
        staged_q, uploaded_q = deque(), deque()
        state = {'i': 0, 'done': False}

        def submit():
            if state['done']:
                return
            try:
                batch = next(it)
            except StopIteration:
                state['done'] = True
                return
            slot = self.slots[state['i'] % self.depth]
            state['i'] += 1
            staged_q.append((slot,) + self._stage(slot, batch))

        def upload():
            slot, n, staged, through = staged_q.popleft()
            out = {}
            if staged:
                with _blocked('upload: stage-in copies'):
                    for futs in staged.values():
                        for f in futs:
                            f.result()
                with torch.cuda.stream(self.copy_stream):
                    if slot.used:
                        self.copy_stream.wait_event(slot.free)      # the consumer of this slot's previous batch is done
                    for key in staged:
                        slot.dev[key][:n].copy_(slot.pinned[key][:n], non_blocking=True)
                        out[key] = slot.dev[key][:n]
                    slot.h2d_done.record(self.copy_stream)
                slot.used = True
            uploaded_q.append((slot, bool(staged), out, through))

        # Uploads run ONE batch ahead of the consumer and staging two ahead of the uploads.  The upload of batch i+1 is
        # issued before the consumer enqueues batch i's kernels and its label download: on a copy queue shared by both
        # directions a download waiting for its kernels would otherwise hold back the next upload (measured: +1.3 ms per
        # 16-image batch).
        ahead = max(1, self.depth - 2)
        while True:
            while len(staged_q) < ahead and not state['done']:
                submit()
            while len(uploaded_q) < 2 and staged_q:
                upload()
                submit()                             # keep the stage-in threads busy while the GPU works
            if not uploaded_q:
                return
            slot, was_staged, out, through = uploaded_q.popleft()
            if was_staged:
                torch.cuda.current_stream(self.device).wait_event(slot.h2d_done)
            for key, v in through.items():
                out[key] = v.to(device=self.device, dtype=self.dtypes[key]).contiguous()
            yield out
            if was_staged:
                slot.free.record(torch.cuda.current_stream(self.device))


class _OutSlot(object):
    def __init__(self):
        self.dev = self.pinned = None
        self.ready = torch.cuda.Event()          # the device-side copy of the result into this slot
        self.d2h_done = torch.cuda.Event()
        self.collected = None                    # future of the host copy out of the pinned buffer
        self.used = False


class ResultFetcher(object):
    """Collect per-batch device tensors into ONE host array (predict's return value) without stalling the compute stream:
    result -> ring slot (device copy on the compute stream, so that a replayed hipGraph may overwrite its static output)
    -> pinned slot on a copy stream -> the result array, by a worker thread."""

    def __init__(self, device, total=None, depth=3, narrow_labels=False):
        """narrow_labels: int64 outputs hold class indices below 256 -- they cross PCIe as one byte per pixel
        (xv_narrow_labels) and are widened to int64 again on the way into the result array.  (The download is a copy
        kernel on this platform: 37.7 MB per 16-image batch held CUs for 0.8 ms beside the persistent conv grids and cost
        1.3 ms per batch; 4.7 MB does not.)"""
        self.device, self.total, self.depth, self.narrow = torch.device(device), total, depth, bool(narrow_labels)
        self.slots = [_OutSlot() for _ in range(depth)]
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.result, self.pos, self.count, self.chunks = None, 0, 0, []

    @staticmethod
    def _collect(slot, dst, n):
        slot.d2h_done.synchronize()
        # the result array is fresh memory: first touch costs more than the copy, so split it over the pool as well (this
        # task itself runs on _collect_pool(): it never occupies a worker of the pool it waits on)
        if _pool()._max_workers >= 8:
            for f in _parallel_copy(dst, slot.pinned.numpy(), n):
                f.result()
        else:
            np.copyto(dst, slot.pinned.numpy()[:n])

    def push(self, out):
        n = out.shape[0]
        slot = self.slots[self.count % self.depth]
        self.count += 1
        if slot.collected is not None:
            with _blocked('push: collect of the slot'):
                slot.collected.result()              # the pinned buffer has been emptied
        narrow = self.narrow and out.dtype == torch.int64 and out.is_contiguous()
        wire = torch.uint8 if narrow else out.dtype
        if slot.dev is None or tuple(slot.dev.shape[1:]) != tuple(out.shape[1:]) or slot.dev.shape[0] < n or \
                slot.dev.dtype != wire:
            slot.dev = torch.empty(tuple(out.shape), dtype=wire, device=self.device)
            slot.pinned = torch.empty(tuple(out.shape), dtype=wire, pin_memory=True)
        main = torch.cuda.current_stream(self.device)
        if slot.used:
            main.wait_event(slot.d2h_done)       # the download that read slot.dev has finished
        if narrow:
            from . import _lib
            import ctypes
            _lib.check(_lib.lib().xv_narrow_labels(ctypes.c_void_p(out.data_ptr()), out.numel(), ctypes.c_void_p(slot.dev.data_ptr()),
                                                   ctypes.c_void_p(main.cuda_stream)), 'xv_narrow_labels')
        else:
            slot.dev[:n].copy_(out, non_blocking=True)
        slot.ready.record(main)
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(slot.ready)
            slot.pinned[:n].copy_(slot.dev[:n], non_blocking=True)
            slot.d2h_done.record(self.copy_stream)
        slot.used = True
        host_dtype = np.int64 if narrow else slot.pinned.numpy().dtype
        if self.total is not None:
            if self.result is None:
                self.result = np.empty((self.total,) + tuple(out.shape[1:]), dtype=host_dtype)
            dst = self.result[self.pos:self.pos + n]
        else:
            dst = np.empty(tuple(out.shape), dtype=host_dtype)
            self.chunks.append(dst)
        self.pos += n
        slot.collected = _collect_pool().submit(self._collect, slot, dst, n)

    def finish(self):
        with _blocked('finish: last collects'):
            for slot in self.slots:
                if slot.collected is not None:
                    slot.collected.result()
        if self.total is not None:
            if self.result is None:
                raise ValueError('no data')
            return self.result[:self.pos]
        return np.concatenate(self.chunks)
