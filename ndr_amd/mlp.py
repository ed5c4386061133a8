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
        # "fp32" (default): the reference's precision -- the fused kernel with split fp16 operands (hi + lo 2^-11, three MFMA
        # products per product, fp32 accumulation; kernels_mlp_x3.hip), outputs within fp32 rounding of networks.MLP.forward.
        # "fp16": plain fp16 operands, three times the throughput, ~4e-4 on the logits.  The backward pass runs at the reference's
        # precision in both modes (split operands in every product; `set_backward_terms(1)` keeps only the hi x hi product in the
        # weight-gradient GEMMs).
        self.precision = "fp32"
        h = ctypes.c_void_p()
        _lib.check(self._lib.vfem_mlp_create(ctypes.byref(h), int(embedding_size), int(n_neurons), int(n_layers),
                                             int(name == "Sigmoid")))
        self._h = h

    def set_backward_terms(self, terms):
        """3 (default): hi hi + hi lo + lo hi in the weight-gradient products; 1: hi hi only (VFEM_MLP_OPT_BWD_TERMS)"""
        _lib.check(self._lib.vfem_mlp_set_option(self._h, 1, int(terms)))

    def set_keep_first_layer(self, keep):
        """training: a reference-precision grid forward keeps the first layer's activations (2 KB per voxel) and the backward pass of
        the same grid starts from them instead of recomputing the first layer (VFEM_MLP_OPT_KEEP_FIRST); same results bit for bit"""
        _lib.check(self._lib.vfem_mlp_set_option(self._h, 2, int(bool(keep))))

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

    def load_tensors(self, B, weights, biases):
        """as ``load_arrays`` from float32 CUDA tensors, without a trip through host memory (training loop)"""
        nl = self.n_layers
        f = lambda t: t.detach().to(device=_dev(), dtype=torch.float32).contiguous()
        Bt, W1 = f(B), f(weights[0])
        Wh = torch.stack([f(w) for w in weights[1:-1]]).contiguous() if nl > 2 else torch.zeros(1, device=_dev())
        bs = torch.stack([f(b) for b in biases[:-1]]).contiguous()
        wout = f(weights[-1]).reshape(-1)
        bout = float(biases[-1].detach().reshape(-1)[0].item())
        if tuple(Bt.shape) != (self.embedding_size, 3) or tuple(W1.shape) != (self.n_neurons, 2 * self.embedding_size):
            raise RuntimeError("weight shapes do not match the network configuration")
        _lib.check(self._lib.vfem_mlp_load_weights(self._h, _ptr(Bt), _ptr(W1), _ptr(Wh), _ptr(bs), _ptr(wout), bout))

    def forward_grid_range(self, sidelen, first_voxel, num_voxels, domain=None, out_f64=None):
        """densities of the voxels [first_voxel, first_voxel + num_voxels) of the grid (flat index, z fastest): what one
        rank of an x-slab decomposition evaluates (its planes are a contiguous voxel range)"""
        n = (ctypes.c_int64 * 3)(*[int(s) for s in sidelen])
        dom = domain if domain is not None else [[0.0, 1.0]] * 3
        lo = (ctypes.c_double * 3)(*[float(d[0]) for d in dom])
        hi = (ctypes.c_double * 3)(*[float(d[1]) for d in dom])
        out = torch.empty(int(num_voxels), dtype=torch.float32, device=_dev())
        o64 = _ptr(out_f64) if out_f64 is not None else None
        _lib.check(self._entry("vfem_mlp_forward_grid_range")(self._h, n, lo, hi, int(first_voxel), int(num_voxels), _ptr(out), o64,
                                                              _stream()))
        return out

    def _entry(self, name):
        if self.precision == "fp16":
            return getattr(self._lib, name)
        if self.precision != "fp32":
            raise RuntimeError("precision must be 'fp16' or 'fp32'")
        return getattr(self._lib, {"vfem_mlp_forward": "vfem_mlp_forward_f32", "vfem_mlp_forward_grid_range": "vfem_mlp_forward_grid_range_f32"}[name])

    # ---- training ----
    def _grad_buffers(self):
        nn_, nl, es = self.n_neurons, self.n_layers, self.embedding_size
        z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=_dev())
        return z(nn_, 2 * es), z(max(nl - 2, 1), nn_, nn_), z(nl - 1, nn_), z(nn_), z(1)

    @staticmethod
    def _auto_scale(g):
        """power-of-two loss scale that brings max |g| to about 64 (the split fp16 operands of the backward products then sit well
        inside fp16's normal range)"""
        m = float(g.abs().max().item())
        return 1.0 if not (m > 0) else float(2.0 ** int(np.round(np.log2(64.0 / m))))

    def _unpack(self, bufs):
        dW1, dWh, db, dwout, dbout = bufs
        nl = self.n_layers
        gw = [dW1] + [dWh[l] for l in range(nl - 2)] + [dwout.reshape(1, -1)]
        gb = [db[j] for j in range(nl - 1)] + [dbout]
        return gw, gb

    def backward(self, coords, g_out, loss_scale=None):
        """gradients of L wrt the Linear weights / biases (lists in layer order, torch layout) given
        g_out = dL/d(out) for the coordinate list of the matching ``forward`` call"""
        c = torch.as_tensor(coords, dtype=torch.float32, device=_dev()).contiguous()
        g = torch.as_tensor(g_out, dtype=torch.float32, device=_dev()).contiguous().reshape(-1)
        n = c.numel() // 3
        if g.numel() != n:
            raise RuntimeError("g_out must hold one value per coordinate")
        bufs = self._grad_buffers()
        sc = self._auto_scale(g) if loss_scale is None else float(loss_scale)
        _lib.check(self._lib.vfem_mlp_backward(self._h, _ptr(c), n, _ptr(g), sc, *[_ptr(b) for b in bufs], _stream()))
        return self._unpack(bufs)

    def backward_grid(self, sidelen, g_out, domain=None, loss_scale=None):
        n = (ctypes.c_int64 * 3)(*[int(s) for s in sidelen])
        dom = domain if domain is not None else [[0.0, 1.0]] * 3
        lo = (ctypes.c_double * 3)(*[float(d[0]) for d in dom])
        hi = (ctypes.c_double * 3)(*[float(d[1]) for d in dom])
        g = torch.as_tensor(g_out, dtype=torch.float32, device=_dev()).contiguous().reshape(-1)
        if g.numel() != int(np.prod([int(s) for s in sidelen])):
            raise RuntimeError("g_out must hold one value per voxel")
        bufs = self._grad_buffers()
        sc = self._auto_scale(g) if loss_scale is None else float(loss_scale)
        _lib.check(self._lib.vfem_mlp_backward_grid(self._h, n, lo, hi, _ptr(g), sc, *[_ptr(b) for b in bufs], _stream()))
        return self._unpack(bufs)

    def backward_grid_range(self, sidelen, first_voxel, num_voxels, g_out, domain=None, loss_scale=None):
        """partial parameter gradients from the voxels [first_voxel, first_voxel + num_voxels) (g_out indexed from the
        start of the range); summed over the ranks of a slab decomposition they are the whole-grid gradients"""
        n = (ctypes.c_int64 * 3)(*[int(s) for s in sidelen])
        dom = domain if domain is not None else [[0.0, 1.0]] * 3
        lo = (ctypes.c_double * 3)(*[float(d[0]) for d in dom])
        hi = (ctypes.c_double * 3)(*[float(d[1]) for d in dom])
        g = torch.as_tensor(g_out, dtype=torch.float32, device=_dev()).contiguous().reshape(-1)
        if g.numel() != int(num_voxels):
            raise RuntimeError("g_out must hold one value per voxel of the range")
        bufs = self._grad_buffers()
        sc = self._auto_scale(g) if loss_scale is None else float(loss_scale)
        _lib.check(self._lib.vfem_mlp_backward_grid_range(self._h, n, lo, hi, int(first_voxel), int(num_voxels), _ptr(g), sc,
                                                          *[_ptr(b) for b in bufs], _stream()))
        return self._unpack(bufs)

    def forward(self, coords):
        """coords [..., 3] float32 -> densities [..., 1] float32 (torch CUDA tensors)"""
        c = torch.as_tensor(coords, dtype=torch.float32, device=_dev()).contiguous()
        n = c.numel() // 3
        out = torch.empty(n, dtype=torch.float32, device=_dev())
        _lib.check(self._entry("vfem_mlp_forward")(self._h, _ptr(c), n, _ptr(out), None, _stream()))
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
        if self.precision == "fp16":
            _lib.check(self._lib.vfem_mlp_forward_grid(self._h, n, lo, hi, _ptr(out), o64, _stream()))
        else:
            _lib.check(self._entry("vfem_mlp_forward_grid_range")(self._h, n, lo, hi, 0, nv, _ptr(out), o64, _stream()))
        return out.reshape(tuple(int(s) for s in sidelen))


class _GridDensity(torch.autograd.Function):
    """density field of the whole grid as an autograd node: forward = fused MFMA kernel, backward = vfem_mlp_backward_grid"""

    @staticmethod
    def forward(ctx, module, *params):
        module._sync()
        ctx.module = module
        if module.voxel_range is None:
            return module.kernel.forward_grid(module.sidelen, module.domain).reshape(-1)
        return module.kernel.forward_grid_range(module.sidelen, module.voxel_range[0], module.voxel_range[1], module.domain)

    @staticmethod
    def backward(ctx, grad_out):
        mod = ctx.module
        if mod.voxel_range is None:
            gw, gb = mod.kernel.backward_grid(mod.sidelen, grad_out, mod.domain)
        else:
            gw, gb = mod.kernel.backward_grid_range(mod.sidelen, mod.voxel_range[0], mod.voxel_range[1], grad_out, mod.domain)
        grads = []
        for w, b in zip(gw, gb):
            grads += [w, b.reshape(-1)]
        if mod.voxel_range is not None and mod.reduce_gradients:
            grads = mod._all_reduce(grads)       # every rank holds a slab of the field and the same weights
        return (None,) + tuple(grads)


class TrainableMLP(torch.nn.Module):
    """``networks.MLP`` (networks.py:126-185) as a trainable torch module whose forward and backward passes run in the
    libvfem kernels: same constructor keywords, same ``net`` Sequential of Linear/ReLU modules (so ``state_dict`` keys and
    checkpoints interchange with the reference, utils.py:259-299), same orthogonal initialisation (networks.py:242-256),
    ``B`` kept outside the parameters.  ``forward_grid()`` returns the flattened density field of the grid set with
    ``set_grid`` (utils.get_mgrid rule); any torch optimiser works on ``parameters()``, ``adam_step`` is the fused one."""

    def __init__(self, in_features=3, out_features=1, n_neurons=256, n_layers=4, embedding_size=256, scale=0,
                 dropout_rate=-1, hidden_act=None, output_act=None):
        super().__init__()
        self.kernel = MLP(in_features, out_features, n_neurons, n_layers, embedding_size, scale, dropout_rate, hidden_act,
                          output_act)
        self.kernel.set_keep_first_layer(True)       # a training step is forward_grid followed by backward_grid of the same grid
        self.embedding_size, self.in_features, self.n_neurons, self.scale = embedding_size, in_features, n_neurons, scale
        self.B = (torch.normal(0, 1, size=(embedding_size, in_features)) * scale).to(_dev())
        layers = []
        for i in range(n_layers):
            if i == 0:
                layers += [torch.nn.Linear(embedding_size * 2, n_neurons), torch.nn.ReLU()]
            elif i == n_layers - 1:
                layers += [torch.nn.Linear(n_neurons, out_features)] + ([torch.nn.Sigmoid()] if output_act is not None else [])
            else:
                layers += [torch.nn.Linear(n_neurons, n_neurons), torch.nn.ReLU()]
        self.net = torch.nn.Sequential(*layers)
        gain = 1.0 * np.sqrt(max(n_neurons / embedding_size, 1))
        for m in self.net:
            if isinstance(m, torch.nn.Linear):
                torch.nn.init.orthogonal_(m.weight, gain=gain)
                torch.nn.init.constant_(m.bias, 0.0)
        self.to(_dev())
        self.sidelen, self.domain, self.voxel_range = None, None, None
        self.reduce_gradients = True
        self._adam_state, self._adam_t = None, 0

    def _linears(self):
        return [m for m in self.net if isinstance(m, torch.nn.Linear)]

    def _sync(self):
        lin = self._linears()
        self.kernel.load_tensors(self.B, [m.weight for m in lin], [m.bias for m in lin])

    def set_grid(self, sidelen, domain=None, voxel_range=None):
        """voxel_range = (first_voxel, num_voxels): this rank evaluates (and differentiates through) only that contiguous part
        of the grid -- the x-planes of its slab; parameter gradients are then summed over the process group"""
        self.sidelen, self.domain = tuple(int(s) for s in sidelen), domain
        self.voxel_range = None if voxel_range is None else (int(voxel_range[0]), int(voxel_range[1]))

    @staticmethod
    def _all_reduce(grads):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return grads
        flat = torch.cat([g.reshape(-1) for g in grads])          # one all-reduce of the 1.57 M gradients
        if dist.get_backend() == "gloo":
            h = flat.cpu()
            dist.all_reduce(h)
            flat = h.to(flat.device)
        else:
            dist.all_reduce(flat)
        out, o = [], 0
        for g in grads:
            out.append(flat[o:o + g.numel()].reshape(g.shape))
            o += g.numel()
        return out

    def forward_grid(self):
        if self.sidelen is None:
            raise RuntimeError("call set_grid first")
        params = []
        for m in self._linears():
            params += [m.weight, m.bias]
        return _GridDensity.apply(self, *params)

    def forward(self, coords=None):
        """with no argument: the grid set by ``set_grid``; with coordinates: inference on that list (no autograd)"""
        if coords is None:
            return self.forward_grid()
        self._sync()
        return self.kernel.forward(coords)

    def homogeneous_init(self, v0):
        """fem.homogeneous_init (fem.py:350-374): the last Linear gets weight ~ N(0, 1e-4) and bias v0, so the initial
        field is uniform"""
        last = self._linears()[-1]
        with torch.no_grad():
            last.weight.normal_(0.0, 1e-4)
            last.bias.fill_(float(v0))

    def adam_step(self, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        """torch.optim.Adam update of every parameter from its ``.grad`` with the fused libvfem kernel"""
        ps = [p for p in self.parameters() if p.grad is not None]
        if self._adam_state is None:
            self._adam_state = {id(p): (torch.zeros_like(p), torch.zeros_like(p)) for p in self.parameters()}
        self._adam_t += 1
        lib = _lib.load()
        for p in ps:
            m, v = self._adam_state[id(p)]
            g = p.grad.contiguous()
            _lib.check(lib.vfem_adam_step(p.numel(), _ptr(p.data), _ptr(g), _ptr(m), _ptr(v), float(lr), float(betas[0]),
                                          float(betas[1]), float(eps), self._adam_t, _stream()))
