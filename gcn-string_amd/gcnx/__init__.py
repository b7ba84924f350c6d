"""gcnx -- MI355X-native GCN forward/backward for DisjointLoader batches.

Host mirror of the Spektral call surface used by Sum02dean/GCN-STRING's src/scripts/gcn.py
(DisjointLoader, GCNConv, GlobalSumPool, model(inputs, training=), train_step) over
hand-written HIP kernels in libgcnx.so (C ABI: include/gcnx.h), bound with ctypes.
No PyTorch, no TensorFlow, no CPU fallback.
"""
from . import _lib
from .io import best_epoch, load_weights_npz, save_to_npz
from .loader import Dataset, DisjointLoader, Graph, ListDataset, NetworkxDataset, SparseTensor, format_graph, from_networkx
from .train import PiecewiseConstantDecay, auc, fit, roc_curve

__all__ = ["Dataset", "DisjointLoader", "Graph", "ListDataset", "NetworkxDataset", "from_networkx", "format_graph", "SparseTensor", "Context", "default_context", "GCNConv", "GeneralConv",
           "GlobalSumPool", "GlobalAvgPool", "GlobalMaxPool", "Dense", "GCN2", "GeneralGNN", "DeviceBatch",
           "save_to_npz", "load_weights_npz", "best_epoch", "DeviceDataset", "DeviceDisjointLoader",
           "PiecewiseConstantDecay", "fit", "roc_curve", "auc"]


def __getattr__(name):  # device-side names load libgcnx lazily, host-only use needs no .so
    if name in ("Context", "default_context", "DeviceArray", "DeviceCSR", "Segments"):
        from . import device
        return getattr(device, name)
    if name in ("GCNConv", "GeneralConv", "GlobalSumPool", "GlobalAvgPool", "GlobalMaxPool", "Dense"):
        from . import layers
        return getattr(layers, name)
    if name in ("DeviceDataset", "DeviceDisjointLoader", "collate_on_device"):
        from . import device_loader
        return getattr(device_loader, name)
    if name in ("GCN2", "GeneralGNN", "DeviceBatch", "evaluate"):
        from . import models
        return getattr(models, name)
    raise AttributeError(name)
