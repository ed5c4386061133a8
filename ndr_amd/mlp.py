"""Device-side Fourier-feature MLP density field: host mirror of ``networks.MLP`` (networks.py:128-185) for
inference (the forward pass of every design iteration, train_xdg.py:282-287).  Weights come from a reference
``MLP`` instance / checkpoint (``utils.save_weights`` keeps ``B`` beside the ``state_dict``, utils.py:259-299);
the fused MFMA kernel lives in libvfem (``vfem_mlp_*``, include/vfem.h)."""
import ctypes

import numpy as np
import torch

from . import _lib
from .pyVoxelFEM import _dev, _ptr, _stream


class MLP:
    """Same constructor keywords as the reference (ReLU hidden activation, output_act None or Sigmoid,
    out_features = 1, no dropout); parameters are set with ``load_reference`` / ``load_arrays``."""

    def __init__(self, in_features=3, out_features=1, n_neurons=256, n_layers=4, embedding_size=256, scale=0,
                 dropout_rate=-1, hidden_act=None, output_act=None):
        if in_features != 3 or out_features != 1:
            raise RuntimeError("the device MLP supports in_features=3, out_features=1")
        if dropout_rate is not None and dropout_rate > 0:
            raise RuntimeError("dropout is a training-time feature; the device MLP is inference only")
        if hidden_act is not None and type(hidden_act).__name__ != "ReLU":
            raise RuntimeError("only ReLU hidden activations are supported")
        name = None if output_act is None else type(output_act).__name__
        if name not in (None, "Sigmoid"):
            raise RuntimeError("output_act must be None or Sigmoid")
        _lib.require_gpu()
        self._lib = _lib.load()
        self.embedding_size, self.n_neurons, self.n_layers, self.scale = embedding_size, n_neurons, n_layers, scale
        h = ctypes.c_void_p()
        _lib.check(self._lib.vfem_mlp_create(ctypes.byref(h), int(embedding_size), int(n_neurons), int(n_layers),
                                             int(name == "Sigmoid")))
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                self._lib.vfem_mlp_destroy(h)
            except Exception:
                pass
            self._h = None

    def load_arrays(self, B, weights, biases):
        """B [es,3]; weights/biases: the Linear layers in order, torch layout ([out, in])."""
        f = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float32))
        nl = self.n_layers
        if len(weights) != nl or len(biases) != nl:
            raise RuntimeError("expected %d Linear layers" % nl)
        B = f(B)
        W1 = f(weights[0])
        Wh = f(np.stack([np.asarray(w) for w in weights[1:-1]])) if nl > 2 else np.zeros((0,), np.float32)
        bs = f(np.stack([np.asarray(b) for b in biases[:-1]]))
        wout = f(np.asarray(weights[-1]).reshape(-1))
        bout = float(np.asarray(biases[-1]).reshape(-1)[0])
        if B.shape != (self.embedding_size, 3) or W1.shape != (self.n_neurons, 2 * self.embedding_size):
            raise RuntimeError("weight shapes do not match the network configuration")
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        _lib.check(self._lib.vfem_mlp_load_weights(self._h, p(B), p(W1), p(Wh), p(bs), p(wout), bout))

    def load_reference(self, model):
        """from a reference ``networks.MLP`` (or anything with ``.B`` and a ``net`` Sequential of Linear layers)"""
        lin = [m for m in model.net if type(m).__name__ == "Linear"]
        self.load_arrays(model.B.detach().cpu().numpy(), [m.weight.detach().cpu().numpy() for m in lin],
                         [m.bias.detach().cpu().numpy() for m in lin])

    def forward(self, coords):
        """coords [..., 3] float32 -> densities [..., 1] float32 (torch CUDA tensors)"""
        c = torch.as_tensor(coords, dtype=torch.float32, device=_dev()).contiguous()
        n = c.numel() // 3
        out = torch.empty(n, dtype=torch.float32, device=_dev())
        _lib.check(self._lib.vfem_mlp_forward(self._h, _ptr(c), n, _ptr(out), None, _stream()))
        return out.reshape(c.shape[:-1] + (1,))

    __call__ = forward

    def forward_grid(self, sidelen, domain=None, out_f64=None):
        """whole-grid evaluation with coordinates generated on the fly (utils.get_mgrid rule); optionally also
        writes float64 densities (what the solver consumes, fem.py:121) into ``out_f64``"""
        n = (ctypes.c_int64 * 3)(*[int(s) for s in sidelen])
        dom = domain if domain is not None else [[0.0, 1.0]] * 3
        lo = (ctypes.c_double * 3)(*[float(d[0]) for d in dom])
        hi = (ctypes.c_double * 3)(*[float(d[1]) for d in dom])
        nv = int(np.prod([int(s) for s in sidelen]))
        out = torch.empty(nv, dtype=torch.float32, device=_dev())
        o64 = _ptr(out_f64) if out_f64 is not None else None
        _lib.check(self._lib.vfem_mlp_forward_grid(self._h, n, lo, hi, _ptr(out), o64, _stream()))
        return out.reshape(tuple(int(s) for s in sidelen))
