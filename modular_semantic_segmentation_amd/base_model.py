"""BaseModel: the drop-in API boundary of the hot path (reference: xview/models/base_model.py).

Same constructor contract, same method names / arguments / return types as the reference's
TensorFlow BaseModel -- `fit`, `predict`, `score`, `export_weights`, `import_weights`, `close`,
context-manager use -- but there is no graph or session: a model owns HIP-resident engines
(fcn.FcnEngine) and every per-pixel op runs in libxview_hip.so on the current HIP stream.

Data contract (base_model.py:10-38 transform_inputdata): `data` is either a dict of arrays with a
leading sample axis ({'rgb': [N,H,W,3] f32, 'depth': [N,H,W,1] f32, 'labels': [N,H,W] i32}), or any
iterable of per-sample dicts (the tf.data.Dataset case); both are cut into batches of
config['batchsize'].
"""
import os

import numpy as np
import torch


def iterate_batches(data, batchsize, max_batches=None):
    """Yield dicts of numpy arrays with a leading batch axis (<= batchsize samples each)."""
    count = 0
    if isinstance(data, dict):
        n = len(next(iter(data.values())))
        for i in range(0, n, batchsize):
            if max_batches is not None and count >= max_batches:
                return
            # numpy on the host, or torch tensors (possibly already resident in HBM) passed through as they are
            yield {k: (v[i:i + batchsize] if isinstance(v, torch.Tensor) else np.asarray(v[i:i + batchsize]))
                   for k, v in data.items()}
            count += 1
        return
    def stack(samples, k):
        if isinstance(samples[0][k], torch.Tensor):          # samples already resident in HBM stay there
            return torch.stack([s[k] for s in samples])
        return np.stack([np.asarray(s[k]) for s in samples])

    pending = []
    for sample in data:
        pending.append(sample)
        if len(pending) == batchsize:
            if max_batches is not None and count >= max_batches:
                return
            yield {k: stack(pending, k) for k in pending[0]}
            count += 1
            pending = []
    if pending and (max_batches is None or count < max_batches):
        yield {k: stack(pending, k) for k in pending[0]}


def score_measures(confusion_matrix):
    """Measures of base_model.py:315-329 from a [C,C] matrix (rows = ground truth)."""
    cm = np.asarray(confusion_matrix, np.float64)
    with np.errstate(divide='ignore', invalid='ignore'):
        measures = {'confusion_matrix': cm}
        diag = np.diag(cm)
        measures['recall'] = diag / cm.sum(1)
        measures['precision'] = diag / cm.sum(0)
        measures['F1'] = 2 * measures['precision'] * measures['recall'] / \
            (measures['precision'] + measures['recall'])
        measures['mean_F1'] = np.nanmean(measures['F1'])
        measures['total_accuracy'] = diag[1:].sum() / cm[1:, :].sum()   # class 0 = void is excluded
        measures['IoU'] = diag / (cm.sum(1) + cm.sum(0) - diag)
        measures['mean_IoU'] = np.nanmean(measures['IoU'][1:])
    return measures


class BaseModel(object):
    """Handles IO, batching, scoring and weight files; subclasses implement `_build_graph`
    (create engines / tables) and `_predict_batch`."""

    required_attributes = [["loss"], ["prediction"]]

    def __init__(self, data_description, name=None, output_dir=None, custom_training=False,
                 batchsize=1, **config):
        self.name = type(self).__name__ if name is None else name
        self.output_dir = output_dir
        self.custom_training = custom_training
        self.config = config
        self.config['batchsize'] = batchsize
        self.config['num_classes'] = data_description[2]
        # (dtypes, shapes, num_classes) as produced by DataBaseclass.get_data_description
        self.testdata_description = [data_description[0],
                                     {key: [None, *shape] for key, shape in data_description[1].items()}]
        self.device = torch.device(config.get('device', 'cuda'))
        self.variables = {}          # name -> np.ndarray, the reference's tf.global_variables()
        self.global_step = 0
        self._closed = False
        self._initialize_graph()

    # ---- to be provided by subclasses ----------------------------------------------------------
    def _build_graph(self):
        raise NotImplementedError

    def _predict_batch_impl(self, batch, output_attr=None):
        """batch: dict of numpy / device arrays -> device tensor (int64 [N,H,W] labels by default)."""
        raise NotImplementedError

    def _graph_flags(self):
        """Host-side switches a captured step depends on besides the weights: a graph captured under other values is not
        replayed (tests and A/B timing flip them between calls)."""
        return (getattr(self, 'concurrent_experts', None), self.config.get('fused_head', True),
                self.config.get('paired_launches', True))

    def _predict_batch(self, batch, output_attr=None):
        g = getattr(self, '_graph', None)
        if g is not None and output_attr is None and g[4] == self._graph_flags() and \
                all(k in batch and tuple(batch[k].shape) == shp for k, shp in g[3].items()):
            graph, static, out = g[:3]
            for k in static:
                static[k].copy_(self._to_device(batch[k], torch.float32))
            graph.replay()
            return out
        return self._predict_batch_impl(batch, output_attr)

    # ---- the pipelined host boundary (host_pipeline.py) -----------------------------------------------------------------
    def _host_dtypes(self, labels):
        """{key: torch dtype} of the arrays a batch carries onto the device: every described input as float32 (the
        reference's placeholders, base_model.py:86-94), labels as int32."""
        d = {k: torch.float32 for k in self.testdata_description[0] if k != 'labels'}
        if labels:
            d['labels'] = torch.int32
        return d

    def _device_batches(self, batches, labels):
        """`batches` (host or device dicts) as HBM-resident dicts, staged and uploaded ahead of the consumer."""
        from . import host_pipeline
        if not host_pipeline.ENABLED or self.device.type != 'cuda':
            return batches
        return host_pipeline.DevicePrefetcher(self.device, batches, self._host_dtypes(labels))

    def _graph_capturable(self):
        """May predict() / score() capture the inference step into a hipGraph on their own?  Off by config
        (`auto_graph: False`) and after a failed attempt (a step with a host synchronisation inside fails its first capture
        and stays eager)."""
        return self.config.get('auto_graph', True) and not getattr(self, '_graph_failed', False)

    def _predict_batch_auto(self, batch, output_attr, state):
        """_predict_batch, with the step captured into a hipGraph once two batches of one shape have run eagerly (the
        third is the capture's warm-up and first replay): ~50 launches on two streams become one graph launch.  A model
        whose step cannot be captured (a host synchronisation inside it) falls back to eager launches for good."""
        if output_attr is None and self._graph_capturable() and self.device.type == 'cuda':
            # (the keys capture_graph makes static: an extra key in the batch must not make every batch look new)
            sig = tuple(sorted((k, tuple(v.shape)) for k, v in batch.items()
                               if k != 'labels' and k in self.testdata_description[0]))
            state['count'] = state.get('count', 0) + 1 if state.get('sig') == sig else 1
            state['sig'] = sig
            g = getattr(self, '_graph', None)
            have = g is not None and g[4] == self._graph_flags() and dict(sig) == g[3]
            if state['count'] >= 3 and not have:
                try:
                    self.capture_graph(batch)
                except Exception:               # noqa: BLE001  (capture is an optimisation: never fail the call over it)
                    self._graph, self._graph_failed = None, True
                    torch.cuda.synchronize(self.device)
        return self._predict_batch(batch, output_attr=output_attr)

    def _train_batch(self, batch):
        raise UserWarning('ERROR: Model %s does not support training' % self.name)

    def _initialize_graph(self):
        self._graph = None           # a captured hipGraph replays the old tables / weight pointers
        self._build_graph()
        if not self.custom_training:
            missing = [attrs for attrs in self.required_attributes
                       if True not in [hasattr(self, attr) for attr in attrs]]
            if missing:
                raise AttributeError('Model class requires attributes %s' % missing)
        elif not hasattr(self, 'prediction'):
            raise AttributeError('Model class required attribute prediction')

    # ---- hipGraph replay of the inference step ---------------------------------------------------------
    def capture_graph(self, batch):
        """Capture `_predict_batch` for inputs of this shape into a hipGraph (launch-bound small batches:
        ~55 kernel launches on two streams collapse into one graph launch).  Later `_predict_batch` calls
        with the same input shapes copy into the static input buffers and replay.  All buffers the
        kernels touch are cached per shape in the engines, so the captured pointers stay valid."""
        keys = [k for k in batch if k != 'labels' and k in self.testdata_description[0]]
        static = {k: self._to_device(batch[k], torch.float32).clone() for k in keys}
        for _ in range(2):                       # warm-up: one-time attribute calls, buffer allocation
            self._predict_batch_impl(static)
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = self._predict_batch_impl(static)
        self._graph = (graph, static, out, {k: tuple(v.shape) for k, v in static.items()}, self._graph_flags())
        return self

    # ---- helpers ------------------------------------------------------------------------------------
    def _to_device(self, array, dtype):
        """numpy (host) or torch (already resident in HBM) -> contiguous device tensor of `dtype`."""
        if isinstance(array, torch.Tensor):
            return array.to(device=self.device, dtype=dtype).contiguous()
        t = torch.from_numpy(np.ascontiguousarray(array))
        return t.to(device=self.device, dtype=dtype, non_blocking=False).contiguous()

    def _confusion_of_batch(self, batch, cm_dev, state=None):
        from . import ops
        pred = self._predict_batch(batch) if state is None else self._predict_batch_auto(batch, None, state)
        labels = self._to_device(batch['labels'], torch.int32)
        ops.confusion_matrix(labels, pred.contiguous(), cm_dev)

    # ---- public API -----------------------------------------------------------------------------------
    def fit(self, dataset, iterations, output=True, validation_dataset=None, validation_interval=100,
            additional_eval_datasets={}):
        """Train for `iterations` steps (base_model.py:180-261)."""
        if self.custom_training:
            raise UserWarning('ERROR: Model %s does not support training' % self.name)

        def samples():
            # dataset.repeat() of the reference (base_model.py:203-206): epochs are chained sample by
            # sample before batching, so every training batch has exactly `batchsize` images
            while True:
                empty = True
                if isinstance(dataset, dict):
                    for i in range(len(next(iter(dataset.values())))):
                        empty = False
                        yield {k: v[i] for k, v in dataset.items()}
                else:
                    for sample in dataset:
                        empty = False
                        yield sample
                if empty:
                    return

        def endless():
            if isinstance(dataset, dict):
                # same sample order as chaining the epochs, without a per-sample round trip through numpy: index the
                # arrays (numpy on the host or torch tensors already resident in HBM) with wrapped indices
                n, bs, start = len(next(iter(dataset.values()))), self.config['batchsize'], 0
                while n > 0:
                    if start + bs <= n:
                        # a contiguous run: views, no gather (the staging threads copy straight out of the caller's arrays;
                        # a fancy-indexed copy of a 57 MB batch on this thread took longer than the training step)
                        batch = {k: v[start:start + bs] for k, v in dataset.items()}
                    else:
                        idx = (start + np.arange(bs)) % n
                        batch = {k: (v[torch.from_numpy(idx).to(v.device)] if isinstance(v, torch.Tensor) else
                                     np.asarray(v)[idx]) for k, v in dataset.items()}
                    start = (start + bs) % n
                    yield batch
                return
            return (yield from iterate_batches(samples(), self.config['batchsize']))

        # The reference writes loss / accuracy / IoU (and one scalar per additional dataset) as tf.summary events into
        # output_dir every validation interval (base_model.py:191-195,226-251); here the same scalars go to a JSON-lines
        # file there, one record per validation step.
        log = None
        if self.output_dir is not None:
            import json
            os.makedirs(self.output_dir, exist_ok=True)
            log = open(os.path.join(self.output_dir, 'training_log.jsonl'), 'a')
        if output:
            print('INFO: Start training')
        # the next batches are staged into pinned memory and uploaded while the current step runs (host_pipeline.py)
        batches = iter(self._device_batches(endless(), labels=True))
        for i in range(iterations):
            loss = self._train_batch(next(batches))
            self.global_step += 1
            if i % validation_interval == 0 and validation_dataset is not None:
                score, _ = self.score(validation_dataset)
                if output:
                    print('{:4d}: loss {:.4f} accuracy {:.2f}, IoU {:.2f}'.format(
                        i, float(loss), score['total_accuracy'], score['mean_IoU']))
                record = {'step': i, 'global_step': int(self.global_step), 'loss': float(loss),
                          'accuracy': float(score['total_accuracy']), 'IoU': float(score['mean_IoU'])}
                for key, extra in additional_eval_datasets.items():
                    record[key] = float(self.score(extra)[0]['mean_IoU'])
                if log is not None:
                    log.write(json.dumps(record) + '\n')
                    log.flush()
                if 'abort_at_iou' in self.config:
                    # collective decision: a rank that stopped alone would leave the others waiting in the next
                    # gradient all-reduce
                    from .parallel import agree_any
                    if agree_any(score['mean_IoU'] > self.config['abort_at_iou'], self.device):
                        break
        if log is not None:
            log.close()
        if output:
            print('INFO: Training finished.')

    def predict(self, data, output_attr=None):
        """Semantic segmentation of `data` (base_model.py:263-292): np.int64 [N,H,W], or the named
        output (e.g. 'prob', 'fused_score') when `output_attr` names something the model exposes."""
        from . import host_pipeline
        batches = iterate_batches(data, self.config['batchsize'])
        if not host_pipeline.ENABLED or self.device.type != 'cuda':
            ret = []
            for batch in batches:
                out = self._predict_batch(batch, output_attr=output_attr)
                ret.append(out.cpu().numpy())
            return np.concatenate(ret)
        # pipelined: batch i+2 is staged into pinned memory and batch i+1 uploaded while batch i computes and the labels
        # of batch i-1 travel back (base_model.py:203-206: the reference's tf.data prefetch in front of sess.run)
        total = len(next(iter(data.values()))) if isinstance(data, dict) else None
        fetch = host_pipeline.ResultFetcher(self.device, total, narrow_labels=output_attr is None and
                                            self.config['num_classes'] <= 256)
        state = {}
        for batch in self._device_batches(batches, labels=False):
            fetch.push(self._predict_batch_auto(batch, output_attr, state))
        return fetch.finish()

    def score(self, data, max_iterations=None):
        """(measures dict, confusion matrix float64 [C,C]) over `data` (base_model.py:294-331)."""
        C = self.config['num_classes']
        cm_dev = torch.zeros((C, C), dtype=torch.int64, device=self.device)
        state = {}
        for batch in self._device_batches(iterate_batches(data, self.config['batchsize'], max_iterations), labels=True):
            self._confusion_of_batch(batch, cm_dev, state)
        if self.config.get('reduce_score_over_ranks', False):
            # one process per GPU, each scored its own shard: sum the [C,C] counts (RCCL all-reduce)
            from .parallel import allreduce_sum_
            allreduce_sum_(cm_dev)
        confusion_matrix = cm_dev.cpu().numpy().astype(np.float64)
        return score_measures(confusion_matrix), confusion_matrix

    # ---- weights ----------------------------------------------------------------------------------------
    def _variables_changed(self):
        """Called after self.variables was modified; subclasses re-upload to the GPU.  The engines allocate
        new weight tensors when they load, so a captured hipGraph (which replays the old pointers) is dropped;
        `capture_graph` must be called again."""
        self._graph = None

    def export_weights(self, save_dir=None):
        """npz of every variable keyed by its TF op name, `{name}_weights_{step}.npz`
        (base_model.py:361-393)."""
        if save_dir is None and self.output_dir is None:
            print('ERROR: No path specified to save weights to.')
            return
        save_dict = {k: np.asarray(v) for k, v in self.variables.items()}
        save_dict['global_step'] = np.asarray(self.global_step, np.int32)
        output_path = os.path.join(save_dir if save_dir is not None else self.output_dir,
                                   '{}_weights_{}.npz'.format(self.name, int(self.global_step)))
        np.savez_compressed(output_path, **save_dict)
        print('INFO: Weights saved to {}'.format(output_path))
        return output_path

    def import_weights(self, filepath, translate_prefix=False, chill_mode=False, warnings=True):
        """Assign variables from an npz by name (base_model.py:396-451): optional prefix translation,
        legacy `prefix_layer/...` names, optimizer slots skipped; a variable stored with another
        shape is reported and skipped (with `chill_mode` the assign is attempted and fails, as in the
        reference); `global_step` is restored for trainable models."""
        if warnings:
            print(filepath)
        weights = np.load(filepath)
        keys = list(weights.keys())
        import_prefix = keys[0].split('/')[0].split('_')[0]

        def translate_name(name):
            if not translate_prefix or not name.startswith(translate_prefix):
                return name
            splitted = name.split('/')
            further = splitted[0].split('_')
            if further[0] == 'forest':
                return name
            further[0] = import_prefix
            splitted[0] = '_'.join(further)
            return '/'.join(splitted)

        for var_name in list(self.variables):
            name = translate_name(var_name)
            if 'grad' in name or 'Adam' in name or 'RMS' in name:
                continue
            legacy = name.replace('/', '_', 1)
            if name in weights or legacy in weights:
                if legacy in weights:
                    name = legacy
                value = weights[name]
                if tuple(value.shape) != tuple(self.variables[var_name].shape):
                    # base_model.py:438-445, literally: the warning is printed, then OUTSIDE chill mode the variable
                    # is skipped (keeps its value) while IN chill mode the assign is attempted anyway -- which
                    # tf.assign rejects for a different static shape with a ValueError
                    if warnings:
                        print('WARNING: wrong shape found for {}, but ignored in chill mode'.format(name))
                        print('stored shape: ', value.shape, 'expected shape: ', self.variables[var_name].shape)
                    if not chill_mode:
                        continue
                    raise ValueError('Shapes %s and %s are incompatible (%s)' % (
                        tuple(self.variables[var_name].shape), tuple(value.shape), name))
                self.variables[var_name] = np.asarray(value, self.variables[var_name].dtype)
            elif warnings:
                print('WARNING: {} not found in saved weights'.format(name))
        # global_step is a tf.global_variable of every trainable model (base_model.py:153-156), so the loop above
        # restores it in the reference; fusion models (custom_training) have none
        if not self.custom_training and 'global_step' in weights:
            self.global_step = int(weights['global_step'])
        self._variables_changed()

    def load_weights(self, filepath):
        """The reference restores a TF checkpoint here (base_model.py:333-339); this build's
        checkpoint format is the npz schema, so it forwards to import_weights."""
        self.import_weights(filepath, warnings=False)

    def close(self):
        self._closed = True

    def __enter__(self):
        return self

    def __exit__(self, *args):
        self.close()
