"""Result / weight files of the training script (host side).

``save_to_npz`` mirrors src/scripts/gcn.py:59-64 (same argument list, same keys: probas, labels, weights,
performance).  The reference passes ``np.array(weights)`` -- a list of per-epoch ``model.get_weights()`` lists, i.e. a
ragged object array that NumPy can only store pickled.  Here every weight tensor additionally gets a key of its
own (``w{epoch}_{index}``), so a file can be read back with ``allow_pickle=False``; ``load_weights_npz`` reads only
those keys and never unpickles anything.
"""
from __future__ import annotations

import os

import numpy as np


def save_to_npz(outputs_file, output_name, probas, labels, weights, performance):
    """gcn.py:59-64.  ``weights``: list (epochs) of lists (Keras variable order) of arrays, as collected at
    gcn.py:383-384; ``performance``: list of per-epoch test accuracies."""
    weights = [list(w) for w in weights]
    flat = {f"w{e}_{i}": np.asarray(a) for e, w in enumerate(weights) for i, a in enumerate(w)}
    counts = np.asarray([len(w) for w in weights], np.int64)
    path = os.path.join(outputs_file, output_name)
    np.savez(path, probas=np.asarray(probas), labels=np.asarray(labels), performance=np.asarray(performance),
             weights_per_epoch=counts, **flat)
    return path if path.endswith(".npz") else path + ".npz"


def load_weights_npz(path, epoch=-1):
    """The weight list of one epoch (default: the last) from a file written by ``save_to_npz``, ready for
    ``model.set_weights``.  Reads plain arrays only (``allow_pickle=False``)."""
    with np.load(path, allow_pickle=False) as z:
        counts = z["weights_per_epoch"]
        e = int(epoch) % len(counts)
        return [z[f"w{e}_{i}"] for i in range(int(counts[e]))]


def best_epoch(path):
    """Index of the epoch with the best stored performance (the reference keeps `performance` for re-loading the
    model with specific weights, gcn.py:382-385)."""
    with np.load(path, allow_pickle=False) as z:
        return int(np.argmax(z["performance"]))
