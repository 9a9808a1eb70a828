"""xview/models/utils.py's cross_entropy under its own name, on the HIP loss kernel."""
import torch

from . import ops


def cross_entropy(log_predictions, labels, return_gradient=False):
    """Mean cross entropy over the labelled pixels of a batch (utils.py:43-53):
    -sum(labels * log_predictions) / (1e-20 + sum(labels)).

    log_predictions: float32 [N,H,W,C] device tensor (tf.nn.log_softmax of the scores -- or the scores themselves: the
    kernel takes the log-softmax of what it is given, and log_softmax is idempotent).  labels: the reference's one-hot
    float [N,H,W,C] (an all-zero row = unlabelled pixel, base_model.py:198-201) or the int32 [N,H,W] class map it was
    made from (values outside [0, C) = unlabelled).  Returns the loss as a float64 device scalar; return_gradient=True also
    returns d loss / d scores = (softmax - onehot) / count (float32 [N,H,W,C]), what the training step feeds backward."""
    c = int(log_predictions.shape[-1])
    logits = log_predictions.to(torch.float32).contiguous()
    dev = logits.device
    if labels.dim() == logits.dim():                # one-hot rows -> class indices (-1 where the row is empty)
        hot = labels.to(dev)
        lab = torch.where(hot.sum(-1) > 0, hot.argmax(-1), torch.full(hot.shape[:-1], -1, device=dev)).to(torch.int32)
    else:
        lab = labels.to(device=dev, dtype=torch.int32)
    lab = lab.contiguous()
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    loss = torch.zeros(1, dtype=torch.float64, device=dev)
    ops.count_valid_labels(lab, c, count)
    grad = ops.softmax_ce_dense(logits, lab, count, c, loss, torch.empty_like(logits))
    return (loss[0], grad) if return_gradient else loss[0]
