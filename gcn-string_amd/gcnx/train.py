"""The training harness of src/scripts/gcn.py around the model step, host side.

* ``PiecewiseConstantDecay`` -- the Keras schedule of gcn.py:321-324, indexed by optimizer STEP (the reference passes
  epoch-derived boundaries ``[0, floor(0.3 * epochs)]`` to a per-step schedule; that quirk is kept: 0.02 for step 0,
  0.002 up to step floor(0.3 * epochs), 0.0002 afterwards).
* ``fit`` -- the loop of gcn.py:364-385: train over ``loader_tr``, after every epoch evaluate on ``loader_te`` (batch-size
  weighted means, gcn.py:362), print the reference's progress line, collect ``model.get_weights()`` and the test
  accuracy per epoch.
* ``roc_curve`` / ``auc`` -- what gcn.py:402-403 takes from scikit-learn, restated in NumPy (same thresholds, same
  trapezoid rule), so the script needs no scikit-learn.
Loaders may be ``DisjointLoader`` (host batches, uploaded per step) or ``DeviceDisjointLoader`` (batches assembled on
the GPU).
"""
from __future__ import annotations

import numpy as np


class PiecewiseConstantDecay:
    """tf.keras.optimizers.schedules.PiecewiseConstantDecay(boundaries, values): values[0] while step <=
    boundaries[0], values[i] while boundaries[i-1] < step <= boundaries[i], values[-1] afterwards."""

    def __init__(self, boundaries, values):
        if len(values) != len(boundaries) + 1:
            raise ValueError("The length of boundaries should be 1 less than the length of values")
        self.boundaries, self.values = list(boundaries), list(values)

    def __call__(self, step):
        for b, v in zip(self.boundaries, self.values):
            if step <= b:
                return float(v)
        return float(self.values[-1])

    @classmethod
    def reference(cls, epochs):
        """The schedule gcn.py:321-324 builds."""
        return cls([0, int(np.floor(0.3 * epochs))], [0.02, 0.002, 0.0002])


def _as_batch(model, inputs, target, normalize):
    from .models import DeviceBatch
    if isinstance(inputs, DeviceBatch):
        return inputs
    return DeviceBatch.from_host(model.ctx, inputs, target, normalize=normalize)


def evaluate(model, loader, normalize=None):
    """gcn.py:342-362: one pass of ``loader.steps_per_epoch`` batches (the test loader is infinite, epochs=None:
    the step count ends the pass) with training=False, eagerly; per-batch loss and accuracy averaged with the batch
    sizes as weights.  Returns ((loss, acc), [probabilities per batch]).  The loss follows ``model.cce_eval``."""
    output, preds = [], []
    step = 0
    while step < loader.steps_per_epoch:
        step += 1
        inputs, target = loader.__next__()
        batch = _as_batch(model, inputs, target, normalize)
        loss, acc, pred = model.evaluate_batch(batch, None)
        preds.append(pred)
        output.append((loss, acc, batch.n_graphs))
    output = np.array(output)
    return tuple(np.average(output[:, :-1], 0, weights=output[:, -1])), preds


def fit(model, loader_tr, loader_te=None, epochs=1, schedule=None, normalize=None, verbose=True):
    """gcn.py:364-385.  ``loader_tr`` must have been built with the same ``epochs`` (it ends the loop, as in the
    reference).  Returns {"history": [(train_loss, train_acc, test_loss, test_acc) per epoch], "weights": [...],
    "performance": [test_acc per epoch]}."""
    schedule = schedule or PiecewiseConstantDecay.reference(epochs)
    epoch = step = 0
    it = 0                                             # optimizer iterations: the schedule's argument
    results, history, weights, performance = [], [], [], []
    for inputs, target in loader_tr:
        step += 1
        batch = _as_batch(model, inputs, target, normalize)
        # the step's loss and accuracy stay on the device until the epoch ends (gcn.py:372 appends them to `results`, which
        # is only read at :375): no host round trip per step, the host queues step k + 1 while the GPU runs step k
        model.train_step(batch, None, lr=schedule(it), fetch="stash")
        it += 1
        if step == loader_tr.steps_per_epoch:
            step = 0
            epoch += 1
            results = model.collect_metrics()
            te = evaluate(model, loader_te, normalize)[0] if loader_te is not None else (float("nan"), float("nan"))
            tr = tuple(np.mean(results, 0))
            if verbose:
                print("Ep. {} - Loss: {:.3f} - Acc: {:.3f} - Test loss: {:.3f} - Test acc: {:.3f}".format(epoch, *tr, *te))
            history.append((*tr, *te))
            weights.append(model.get_weights())
            performance.append(te[-1])
            results = []
    return {"history": history, "weights": weights, "performance": performance}


def roc_curve(labels, scores):
    """sklearn.metrics.roc_curve(labels, scores) for binary labels in {0, 1} (drop_intermediate=False): thresholds
    are the distinct scores in decreasing order, preceded by +inf.  Returns (fpr, tpr, thresholds)."""
    y = np.asarray(labels).astype(np.float64).ravel()
    s = np.asarray(scores, np.float64).ravel()
    order = np.argsort(-s, kind="mergesort")
    y, s = y[order], s[order]
    distinct = np.where(np.diff(s))[0]
    idx = np.r_[distinct, y.size - 1]
    tps = np.cumsum(y)[idx]
    fps = 1 + idx - tps
    tps, fps = np.r_[0, tps], np.r_[0, fps]
    thr = np.r_[np.inf, s[idx]]
    fpr = fps / fps[-1] if fps[-1] > 0 else np.full_like(fps, np.nan, dtype=np.float64)
    tpr = tps / tps[-1] if tps[-1] > 0 else np.full_like(tps, np.nan, dtype=np.float64)
    return fpr, tpr, thr


def auc(x, y):
    """sklearn.metrics.auc: trapezoidal area under the points (x monotonic)."""
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    dx = np.diff(x)
    if np.any(dx < 0):
        if np.all(dx <= 0):
            return -float(np.trapezoid(y, x)) if hasattr(np, "trapezoid") else -float(np.trapz(y, x))
        raise ValueError("x is neither increasing nor decreasing")
    return float(np.trapezoid(y, x)) if hasattr(np, "trapezoid") else float(np.trapz(y, x))
