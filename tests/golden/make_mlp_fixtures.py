"""Generates tests/golden/mlp_*.npz by importing the reference's networks.MLP (CPU torch) in this container.
Run once from the repo root:  python tests/golden/make_mlp_fixtures.py
Only inputs and outputs (weights, B, coordinates, densities) are stored; no reference source travels."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
import networks  # noqa: E402  (reference module)

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from helpers import seeded_mlp_weights  # noqa: E402  (tests/helpers.py: the build's own seeded initialiser)


def mgrid(sidelen):
    axes = [torch.linspace(0.0, 1.0, steps=n) for n in sidelen]
    return torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=-1)[None]      # utils.get_mgrid, flatten=False


def dump(tag, es, nn_, nl, sigma, sidelen, sigmoid, seed, weight_scale=None):
    torch.manual_seed(seed)
    act = torch.nn.Sigmoid() if sigmoid else None
    model = networks.MLP(in_features=3, out_features=1, n_neurons=nn_, n_layers=nl, embedding_size=es, scale=sigma,
                         hidden_act=torch.nn.ReLU(), output_act=act)
    lin = [m for m in model.net if isinstance(m, torch.nn.Linear)]
    with torch.no_grad():
        for m in lin:                      # non-zero biases so that the bias path is exercised
            m.bias.normal_(0.0, 0.1)
            if weight_scale is not None:
                m.weight.normal_(0.0, weight_scale / np.sqrt(m.weight.shape[1]))
    coords = mgrid(sidelen)
    with torch.no_grad():
        out = model(coords)
    # gradients of L = sum_v gout[v] * out[v] by the reference's own autograd path (train_xdg.py:282-329)
    gen = torch.Generator().manual_seed(seed + 1000)
    gout = torch.randn(out.shape, generator=gen) * torch.exp(2.0 * torch.randn(out.shape, generator=gen))
    model.zero_grad()
    (model(coords) * gout).sum().backward()
    arrays = {"B": model.B.numpy(), "coords": coords.numpy(), "out": out.numpy(), "gout": gout.numpy(),
              "cfg": np.array([es, nn_, nl, int(sigmoid)]), "sigma": np.array([sigma])}
    for i, m in enumerate(lin):
        arrays["W%d" % i] = m.weight.detach().numpy()
        arrays["b%d" % i] = m.bias.detach().numpy()
        arrays["gW%d" % i] = m.weight.grad.detach().numpy()
        arrays["gb%d" % i] = m.bias.grad.detach().numpy()
    np.savez_compressed(os.path.join(HERE, "mlp_%s.npz" % tag), **arrays)
    print(tag, out.shape, float(out.min()), float(out.max()))


def dump_full_size(tag, es, nn_, nl, sigma, sidelen, seed, row_stride=8):
    """reference networks.MLP with the seeded weights loaded; forward (logits and sigmoid head) + autograd gradients; weight
    gradients are stored for every `row_stride`-th output row only"""
    B, Ws, bs = seeded_mlp_weights(es, nn_, nl, sigma, seed)
    arrays = {"cfg": np.array([es, nn_, nl, 0]), "sigma": np.array([sigma]), "seed": np.array([seed]),
              "row_stride": np.array([row_stride]),
              "weight_checksum": np.array([float(np.sum([np.abs(w.astype(np.float64)).sum() for w in Ws + bs + [B]]))])}
    coords = mgrid(sidelen)
    arrays["coords"] = coords.numpy()
    for sig in (False, True):
        model = networks.MLP(in_features=3, out_features=1, n_neurons=nn_, n_layers=nl, embedding_size=es, scale=sigma,
                             hidden_act=torch.nn.ReLU(), output_act=torch.nn.Sigmoid() if sig else None)
        model.B = torch.from_numpy(B)
        lin = [m for m in model.net if isinstance(m, torch.nn.Linear)]
        with torch.no_grad():
            for m, w, b in zip(lin, Ws, bs):
                m.weight.copy_(torch.from_numpy(w))
                m.bias.copy_(torch.from_numpy(b))
            out = model(coords)
        arrays["out_sig" if sig else "out"] = out.numpy()
        if sig:
            continue
        gen = torch.Generator().manual_seed(seed + 1000)
        gout = torch.randn(out.shape, generator=gen) * torch.exp(2.0 * torch.randn(out.shape, generator=gen))
        model.zero_grad()
        (model(coords) * gout).sum().backward()
        arrays["gout"] = gout.numpy()
        for i, m in enumerate(lin):
            g = m.weight.grad.detach().numpy()
            arrays["gW%d" % i] = g[::row_stride] if g.shape[0] > 1 else g
            arrays["gb%d" % i] = m.bias.grad.detach().numpy()
    np.savez_compressed(os.path.join(HERE, "mlpfull_%s.npz" % tag), **arrays)
    print(tag, arrays["out"].shape, float(arrays["out"].min()), float(arrays["out"].max()))


if __name__ == "__main__":
    dump("es32_nn32_nl4_s1", 32, 32, 4, 1.0, (8, 4, 4), False, 1)
    dump("es64_nn64_nl3_s2p5_sig", 64, 64, 3, 2.5, (8, 4, 4), True, 2)
    dump("es64_nn128_nl2_s4", 64, 128, 2, 4.0, (5, 3, 7), False, 3)
    dump("es128_nn256_nl4_s4_sig", 128, 256, 4, 4.0, (6, 6, 6), True, 4, weight_scale=1.0)
    # the run.md network (train_xdg.py:190-201: n_neurons 512, n_layers 4, embedding 1024) at sigma = 4 on 256 voxels:
    # the only shape the 512-wide kernel specialisation runs at
    dump_full_size("es1024_nn512_nl4_s4", 1024, 512, 4, 4.0, (8, 8, 4), 5)
