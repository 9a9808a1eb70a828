"""The reference's evaluation / fusion experiment flows without the sacred experiment database
(reference: experiments/evaluation.py:14-41,62-110, experiments/bayes_fusion.py:21-33,146-195,
experiments/dirichlet_fusion.py:19-81, experiments/training.py).

Datasets here are dicts of arrays ({'rgb': [N,H,W,3], 'depth': [N,H,W,1], 'labels': [N,H,W]}) or any iterable of
per-sample dicts (the data contract of base_model.iterate_batches); the reference's `tf.data` readers, the
sacred observers and the experiment-id look-ups of `starting_weights` are out of scope -- `starting_weights` is a
path to an npz in the reference's variable-name schema, a list of such paths, or a {prefix: path} dict.
"""
from copy import deepcopy

import numpy as np

from . import get_model
from .bayes_mix import BayesFusion
from .dirichlet_mix import DirichletFusion


def import_weights_into_network(net, starting_weights, **kwargs):
    """experiments/evaluation.py:62-110 minus the experiment database: import one or several npz files."""
    if starting_weights is None or starting_weights == '':
        return
    if isinstance(starting_weights, dict):
        starting_weights = list(starting_weights.values())
    if isinstance(starting_weights, (list, tuple)):
        for path in starting_weights:
            net.import_weights(path, warnings=False, **kwargs)
    else:
        net.import_weights(starting_weights, warnings=False, **kwargs)


def split_test_data(testset, test_size=.5, random_state=1):
    """experiments/bayes_fusion.py:21-33: the test set is split in halves, one to measure the experts' statistics on
    (confusion matrices / Dirichlet sufficient statistics) and one to test the fusion on.  Deterministic shuffle like
    sklearn.model_selection.train_test_split(random_state=1)."""
    n = len(next(iter(testset.values())))
    from sklearn.model_selection import train_test_split
    measure_idx, test_idx = train_test_split(np.arange(n), test_size=test_size, random_state=random_state)
    take = lambda idx: {k: np.asarray(v)[idx] for k, v in testset.items()}   # noqa: E731
    return take(measure_idx), take(test_idx)


def evaluate(net, testset, print_results=True, labelinfo=None):
    """experiments/evaluation.py:14-41."""
    measures, confusion_matrix = net.score(testset)
    if print_results:
        print('total accuracy {:.3f} mean F1 {:.3f} IoU {:.3f}'.format(
            measures['total_accuracy'], measures['mean_F1'], measures['mean_IoU']))
        for label in (labelinfo or {}):
            print('{:>15}: {:.2f} precision, {:.2f} recall, {:.2f} IoU'.format(
                labelinfo[label]['name'], measures['precision'][label], measures['recall'][label],
                measures['IoU'][label]))
    return measures, confusion_matrix


def train_and_evaluate(modelname, net_config, data_description, trainset, testset, num_iterations, starting_weights=None,
                       validation_set=None, output_dir=None):
    """experiments/training.py: build the model, optionally warm-start, fit, export the weights, evaluate."""
    info = {}
    with get_model(modelname)(data_description=data_description, output_dir=output_dir, **net_config) as net:
        import_weights_into_network(net, starting_weights)
        net.fit(trainset, num_iterations, output=False, validation_dataset=validation_set)
        info['weights'] = net.export_weights() if output_dir else None
        info['measurements'], info['confusion_matrix'] = evaluate(net, testset, print_results=False)
    return info


def fit_and_evaluate_bayes_fusion(net_config, data_description, measure_set, test_set, starting_weights):
    """experiments/bayes_fusion.py:146-195: score every expert on the measurement set (its confusion matrix feeds the
    fusion) and on the test set, then score the Bayes fusion on the test set.
    net_config: expert_model, prefixes, num_channels, num_units, class_prior, ...; starting_weights: {prefix: npz}."""
    info = {'measurements': {}}
    model = get_model(net_config['expert_model'])
    confusion_matrices = {}
    for expert in net_config['num_channels']:
        model_config = deepcopy(net_config)
        prefix = net_config['prefixes'][expert]
        for key in ('prefixes', 'num_channels', 'expert_model', 'class_prior'):
            model_config.pop(key, None)
        model_config.setdefault('batch_normalization', False)
        with model(prefix, data_description, expert, **model_config) as net:
            import_weights_into_network(net, starting_weights[prefix] if isinstance(starting_weights, dict)
                                        else starting_weights)
            _, conf_mat = net.score(measure_set)
            confusion_matrices[expert] = conf_mat
            info['measurements'][expert], _ = net.score(test_set)
    info['confusion_matrices'] = confusion_matrices
    with BayesFusion(data_description=data_description, confusion_matrices=confusion_matrices, **net_config) as net:
        import_weights_into_network(net, starting_weights)
        info['measurements']['fusion'], info['confusion_matrix'] = net.score(test_set)
    return info


def fit_and_evaluate_dirichlet_fusion(net_config, data_description, measure_set, test_set, starting_weights):
    """experiments/dirichlet_fusion.py:58-81: fit the Dirichlet parameters on the measurement set, re-import the
    weights, score on the test set."""
    info = {}
    with DirichletFusion(data_description=data_description, **net_config) as net:
        import_weights_into_network(net, starting_weights)
        info['dirichlet_params'] = net.fit(measure_set)
        import_weights_into_network(net, starting_weights)
        info['measurements'], info['confusion_matrix'] = net.score(test_set)
    return info
