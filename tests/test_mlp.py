"""Fourier-feature MLP: golden vectors produced by the reference's networks.MLP (tests/golden/make_mlp_fixtures.py).
CPU: the oracle restatement must reproduce them (fp32).  GPU: the fused MFMA kernels (forward at the reference's precision and with
plain fp16 operands, backward at the reference's precision) against the same vectors, and the grid entry points against the
explicit-coordinate ones."""
import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
FIXTURES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "mlp_*.npz")))
FULL = os.path.join(ROOT, "tests", "golden", "mlpfull_es1024_nn512_nl4_s4.npz")

# fp16 operands, fp32 accumulation: absolute tolerance on the (pre-sigmoid O(1)) output.  Measured maxima on the fixtures:
# see DESIGN.md section 3.5 (the bound is dominated by the fp16 rounding of the Fourier features and activations,
# eps_f16 = 4.9e-4 relative per operand, accumulated over K = 2048 / 512 terms with random signs)
TOL_F16 = 6e-3


def _load_full():
    from helpers import mlp_weight_checksum, seeded_mlp_weights
    z = np.load(FULL)
    es, nn_, nl, _ = [int(v) for v in z["cfg"]]
    B, Ws, bs = seeded_mlp_weights(es, nn_, nl, float(z["sigma"][0]), int(z["seed"][0]))
    # the weights are regenerated from the seed (same numpy build as the generator's): a different stream is a broken
    # fixture, not a parity failure
    assert abs(mlp_weight_checksum(B, Ws, bs) - float(z["weight_checksum"][0])) < 1e-9 * float(z["weight_checksum"][0])
    return z, es, nn_, nl, B, Ws, bs


def test_oracle_reproduces_reference_mlp_full_size():
    """the run.md network (1024 features, 512 neurons, 4 layers, sigma 4) on 256 voxels, logits and sigmoid head"""
    from oracle import vfem_oracle as vo
    z, es, nn_, nl, B, Ws, bs = _load_full()
    out = vo.mlp_forward(z["coords"], B, Ws, bs, False).reshape(z["out"].shape)
    assert np.abs(out - z["out"]).max() < 2e-5
    out = vo.mlp_forward(z["coords"], B, Ws, bs, True).reshape(z["out"].shape)
    assert np.abs(out - z["out_sig"]).max() < 2e-5


@pytest.mark.gpu
def test_hip_mlp_full_size_specialisation_matches_reference():
    """k_mlp_forward<true> (n_neurons = 512: the instantiation bench.py times and config 4 runs) and the 512-wide backward
    against the reference's networks.MLP forward and autograd gradients"""
    import torch
    from ndr_amd.mlp import MLP
    z, es, nn_, nl, B, Ws, bs = _load_full()
    sigma, side = float(z["sigma"][0]), z["coords"].shape[1:4]
    report = {}
    for sig, key in ((False, "out"), (True, "out_sig")):
        m = MLP(3, 1, nn_, nl, es, sigma, output_act=torch.nn.Sigmoid() if sig else None)
        m.load_arrays(B, Ws, bs)
        m.precision = "fp16"                     # the plain fp16-operand kernel (the default is the reference-precision one, below)
        got = m.forward(torch.from_numpy(z["coords"]).cuda()).cpu().numpy().reshape(z[key].shape)
        grid = m.forward_grid(side).cpu().numpy().reshape(z[key].shape)
        report[key] = (float(np.abs(got - z[key]).max()), float(np.abs(grid - z[key]).max()))
        assert report[key][0] < TOL_F16 and report[key][1] < TOL_F16, report
    m = MLP(3, 1, nn_, nl, es, sigma)
    m.load_arrays(B, Ws, bs)
    stride = int(z["row_stride"][0])
    gout = torch.from_numpy(z["gout"]).cuda()
    for gw, gb in (m.backward(torch.from_numpy(z["coords"]).cuda(), gout), m.backward_grid(side, gout)):
        for i in range(nl):
            w = gw[i].cpu().numpy().reshape(Ws[i].shape)
            w = w[::stride] if w.shape[0] > 1 else w
            ew, eb = _rel_l2(w, z["gW%d" % i]), _rel_l2(gb[i].cpu().numpy().reshape(-1), z["gb%d" % i].reshape(-1))
            report.setdefault("g%d" % i, []).append((ew, eb))
    from helpers import record_deltas
    record_deltas("mlp_full_size", report)
    # Round 4: the backward pass runs at the reference's precision (split operands in every product, kernels_mlp_bwd.hip): measured
    # 4.8e-7 .. 6.8e-7 against the reference's fp32 autograd on this fixture through the explicit coordinates (rounds 1-3, fp16
    # operands: 2.6 .. 5.8 %).  The grid entry point forms its coordinates as lo + i * step in fp32, one rounding away from
    # torch.linspace's: 2 pi sigma |B| times that moves the first layers' gradients by ~1e-4 (as it moves the forward's logits).
    for i in range(nl):
        (ew_c, eb_c), (ew_g, eb_g) = report["g%d" % i]
        assert ew_c < 5e-6 and eb_c < 5e-6, report
        assert ew_g < 1e-3 and eb_g < 1e-3, report


def _load(path):
    z = np.load(path)
    es, nn_, nl, sig = [int(v) for v in z["cfg"]]
    Ws = [z["W%d" % i] for i in range(nl)]
    bs = [z["b%d" % i] for i in range(nl)]
    return z, es, nn_, nl, bool(sig), Ws, bs


@pytest.mark.parametrize("path", FIXTURES)
def test_oracle_reproduces_reference_mlp(path):
    from oracle import vfem_oracle as vo
    z, es, nn_, nl, sig, Ws, bs = _load(path)
    out = vo.mlp_forward(z["coords"], z["B"], Ws, bs, sig).reshape(z["out"].shape)
    assert np.abs(out - z["out"]).max() < 2e-5


def test_mgrid_rule():
    from oracle import vfem_oracle as vo
    z = np.load(FIXTURES[0])
    g = vo.get_mgrid(z["coords"].shape[1:4])
    assert np.array_equal(g, z["coords"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES)
def test_hip_mlp_matches_reference(path):
    import torch
    from ndr_amd.mlp import MLP
    z, es, nn_, nl, sig, Ws, bs = _load(path)
    m = MLP(3, 1, nn_, nl, es, float(z["sigma"][0]), output_act=torch.nn.Sigmoid() if sig else None)
    m.load_arrays(z["B"], Ws, bs)
    m.precision = "fp16"
    got = m.forward(torch.from_numpy(z["coords"]).cuda()).cpu().numpy().reshape(z["out"].shape)
    err = np.abs(got - z["out"]).max()
    assert err < TOL_F16, err
    side = z["coords"].shape[1:4]
    grid = m.forward_grid(side).cpu().numpy().reshape(z["out"].shape)
    assert np.abs(grid - got).max() < 2e-3      # same kernel, coordinates regenerated on the fly in fp32
    rho64 = torch.empty(int(np.prod(side)), dtype=torch.float64, device="cuda")
    m.forward_grid(side, out_f64=rho64)
    assert np.abs(rho64.cpu().numpy().reshape(z["out"].shape) - grid).max() < 1e-7


TOL_F32 = 5e-5      # reference precision: summation order, sin / cos rounding and the dropped lo x lo products (2^-22) differ


@pytest.mark.gpu
def test_hip_mlp_fp32_mode_matches_reference_to_fp32_rounding():
    """precision = "fp32" (the default: fused kernel with split fp16 operands, fp32 features formed as the reference forms them):
    every fixture, the full-size network, explicit coordinates, the whole grid, a voxel range that starts and ends inside
    64-voxel blocks, and the float64 copy"""
    import torch
    from ndr_amd.mlp import MLP
    from helpers import record_deltas
    worst = 0.0
    for path in FIXTURES:
        z, es, nn_, nl, sig, Ws, bs = _load(path)
        m = MLP(3, 1, nn_, nl, es, float(z["sigma"][0]), output_act=torch.nn.Sigmoid() if sig else None)
        m.load_arrays(z["B"], Ws, bs)
        m.precision = "fp32"
        got = m.forward(torch.from_numpy(z["coords"]).cuda()).cpu().numpy().reshape(z["out"].shape)
        side = z["coords"].shape[1:4]
        rho64 = torch.empty(int(np.prod(side)), dtype=torch.float64, device="cuda")
        grid = m.forward_grid(side, out_f64=rho64).cpu().numpy().reshape(z["out"].shape)
        worst = max(worst, float(np.abs(got - z["out"]).max()), float(np.abs(grid - z["out"]).max()))
        assert np.array_equal(rho64.cpu().numpy().reshape(grid.shape), grid.astype(np.float64))
    z, es, nn_, nl, B, Ws, bs = _load_full()
    for sig, key in ((False, "out"), (True, "out_sig")):
        m = MLP(3, 1, nn_, nl, es, float(z["sigma"][0]), output_act=torch.nn.Sigmoid() if sig else None)
        m.load_arrays(B, Ws, bs)
        m.precision = "fp32"
        got = m.forward(torch.from_numpy(z["coords"]).cuda()).cpu().numpy().reshape(z[key].shape)
        worst = max(worst, float(np.abs(got - z[key]).max()))
    record_deltas("mlp_fp32_mode", {"max_abs_error_vs_reference": worst})
    assert worst < TOL_F32, worst
    # a voxel range equals the slice of the whole grid
    side = (40, 32, 24)
    whole = m.forward_grid(side).reshape(-1)
    part = m.forward_grid_range(side, 9000, 17000)
    assert float((part - whole[9000:26000]).abs().max()) < 1e-5
    m.precision = "fp64"
    with pytest.raises(RuntimeError):
        m.forward_grid(side)


@pytest.mark.gpu
def test_hip_mlp_requires_weights_and_valid_shapes():
    import torch
    from ndr_amd.mlp import MLP
    with pytest.raises(RuntimeError):
        MLP(3, 1, 100, 4, 64, 1.0)                     # n_neurons not a multiple of 32
    m = MLP(3, 1, 64, 3, 32, 1.0)
    with pytest.raises(RuntimeError):
        m.forward(torch.zeros(4, 3))


@pytest.mark.gpu
def test_hip_mlp_reports_values_outside_the_split_operands_range():
    """the reference-precision kernels carry fp16(x) as the high half of every operand: a weight >= 65504 is refused when it is loaded,
    a hidden activation that large is reported by the next call instead of silently becoming inf -> NaN (the fp32 reference, and the
    SGEMM path of rounds 1-2, stay finite there)"""
    import torch
    from ndr_amd.mlp import MLP
    rng = np.random.default_rng(2)
    es, nn_, nl = 32, 64, 3
    B = rng.standard_normal((es, 3)).astype(np.float32)
    Ws = [rng.standard_normal((nn_, 2 * es)).astype(np.float32) * 0.1, rng.standard_normal((nn_, nn_)).astype(np.float32) * 0.1,
          rng.standard_normal((1, nn_)).astype(np.float32) * 0.1]
    bs = [np.zeros(nn_, np.float32), np.zeros(nn_, np.float32), np.zeros(1, np.float32)]
    m = MLP(3, 1, nn_, nl, es, 1.0)
    big = [w.copy() for w in Ws]
    big[1][3, 5] = 7.0e4
    with pytest.raises(RuntimeError, match="65504"):
        m.load_arrays(B, big, bs)
    # weights in range, first-layer bias of 1e5: the first hidden activations leave fp16's range
    m.load_arrays(B, Ws, [np.full(nn_, 1.0e5, np.float32), bs[1], bs[2]])
    x = torch.rand(256, 3, device="cuda")
    m.forward(x)                                       # raises the flag ...
    with pytest.raises(RuntimeError, match="fp16's range"):
        m.forward(x)                                   # ... which the next call reports
    m.load_arrays(B, Ws, bs)                           # a sane network afterwards works
    out = m.forward(x)
    assert bool(torch.isfinite(out).all())
    m.forward(x)


def _rel_l2(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / max(np.linalg.norm(np.asarray(b, np.float64)), 1e-300))


# relative L2 error of every gradient tensor against the reference's fp32 autograd gradients; set from measurement
TOL_GRAD = 5e-6           # explicit coordinates: fp32 autograd to rounding (measured <= 6.5e-7)
TOL_GRAD_GRID = 1e-3      # grid entry point: its fp32 coordinates are one rounding away from torch.linspace's (see above)


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES)
def test_hip_mlp_gradients_match_reference_autograd(path):
    """dL/dW, dL/db of L = sum_v gout[v] out[v] against the gradients torch.autograd computed on the reference's
    networks.MLP (fixtures); explicit coordinate list and grid entry points"""
    import torch
    from ndr_amd.mlp import MLP
    z, es, nn_, nl, sig, Ws, bs = _load(path)
    m = MLP(3, 1, nn_, nl, es, float(z["sigma"][0]), output_act=torch.nn.Sigmoid() if sig else None)
    m.load_arrays(z["B"], Ws, bs)
    coords = torch.from_numpy(z["coords"]).cuda()
    gout = torch.from_numpy(z["gout"]).cuda()
    for tol, (gw, gb) in ((TOL_GRAD, m.backward(coords, gout)), (TOL_GRAD_GRID, m.backward_grid(z["coords"].shape[1:4], gout))):
        for i in range(nl):
            ew = _rel_l2(gw[i].cpu().numpy().reshape(z["gW%d" % i].shape), z["gW%d" % i])
            eb = _rel_l2(gb[i].cpu().numpy().reshape(z["gb%d" % i].shape), z["gb%d" % i])
            assert ew < tol and eb < tol, (i, ew, eb)
    # the cheaper setting of the weight-gradient products (hi x hi only): fp16 rounding of the operands, 3e-4
    m.set_backward_terms(1)
    gw, gb = m.backward(coords, gout)
    for i in range(nl):
        assert _rel_l2(gw[i].cpu().numpy().reshape(z["gW%d" % i].shape), z["gW%d" % i]) < 2e-3, i


@pytest.mark.gpu
def test_hip_mlp_gradients_multi_chunk_and_linearity():
    """a voxel set larger than one chunk (2^20 rows): gradients are linear in g_out and independent of the loss scale"""
    import torch
    from ndr_amd.mlp import MLP
    rng = np.random.default_rng(5)
    es, nn_, nl = 64, 128, 4
    m = MLP(3, 1, nn_, nl, es, 3.0)
    B = (rng.standard_normal((es, 3)) * 3.0).astype(np.float32)
    Ws = [rng.standard_normal((nn_, 2 * es)).astype(np.float32) / np.sqrt(2 * es)] + \
         [rng.standard_normal((nn_, nn_)).astype(np.float32) / np.sqrt(nn_) for _ in range(nl - 2)] + \
         [rng.standard_normal((1, nn_)).astype(np.float32) / np.sqrt(nn_)]
    bs = [rng.standard_normal(nn_).astype(np.float32) * 0.1 for _ in range(nl - 1)] + [np.array([0.1], np.float32)]
    m.load_arrays(B, Ws, bs)
    side = (130, 96, 100)                                      # 1 248 000 voxels: two chunks, ragged tail
    gen = torch.Generator(device="cuda").manual_seed(11)
    g1 = torch.randn(int(np.prod(side)), device="cuda", generator=gen)
    g2 = torch.randn(int(np.prod(side)), device="cuda", generator=gen)
    a_w, a_b = m.backward_grid(side, g1, loss_scale=16.0)
    b_w, b_b = m.backward_grid(side, g2, loss_scale=16.0)
    c_w, c_b = m.backward_grid(side, g1 + 2.0 * g2, loss_scale=4.0)
    # white-noise g_out: the gradient sums cancel to ~sqrt(N) of their terms, which magnifies whatever rounding there is
    for i in range(nl):
        assert _rel_l2((a_w[i] + 2.0 * b_w[i]).cpu().numpy(), c_w[i].cpu().numpy()) < 1e-4, i
        assert _rel_l2((a_b[i] + 2.0 * b_b[i]).cpu().numpy(), c_b[i].cpu().numpy()) < 1e-4, i
    # the first chunk alone, through the explicit-coordinate entry point, against the grid's own coordinates
    from oracle import vfem_oracle as vo
    sub = (4, 96, 100)
    coords = torch.from_numpy(vo.get_mgrid(side)[0, :4].reshape(-1, 3).astype(np.float32)).cuda()
    gsub = g1[:coords.shape[0]].clone()
    gfull = torch.zeros_like(g1)
    gfull[:coords.shape[0]] = gsub
    e_w, e_b = m.backward(coords, gsub)
    f_w, f_b = m.backward_grid(side, gfull)
    for i in range(nl):
        assert _rel_l2(e_w[i].cpu().numpy(), f_w[i].cpu().numpy()) < 5e-3, i      # (grid coordinates in fp32 vs the list's: one rounding apart, as above)


@pytest.mark.gpu
def test_backward_from_kept_first_layer_activations_is_bitwise_the_recomputed_one():
    """VFEM_MLP_OPT_KEEP_FIRST: the grid forward keeps the first layer's activations and the backward pass of the same grid skips the
    first layer's recomputation -- same gradients bit for bit, also over several chunks with a ragged tail and for a voxel range; a
    backward pass of another grid, or after new weights, falls back to the recomputation"""
    import torch
    from ndr_amd.mlp import MLP
    rng = np.random.default_rng(9)
    es, nn_, nl = 64, 128, 4
    B = (rng.standard_normal((es, 3)) * 2.0).astype(np.float32)
    Ws = [rng.standard_normal((nn_, 2 * es)).astype(np.float32) / np.sqrt(2 * es)] + \
         [rng.standard_normal((nn_, nn_)).astype(np.float32) / np.sqrt(nn_) for _ in range(nl - 2)] + \
         [rng.standard_normal((1, nn_)).astype(np.float32) / np.sqrt(nn_)]
    bs = [rng.standard_normal(nn_).astype(np.float32) * 0.1 for _ in range(nl - 1)] + [np.array([0.1], np.float32)]
    m = MLP(3, 1, nn_, nl, es, 2.0)
    m.load_arrays(B, Ws, bs)
    side = (130, 96, 100)                                      # 1 248 000 voxels: two chunks, ragged tail
    g = torch.randn(int(np.prod(side)), device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    ref_w, ref_b = m.backward_grid(side, g, loss_scale=16.0)
    ref = [t.clone() for t in ref_w + ref_b]
    m.set_keep_first_layer(True)
    out_keep = m.forward_grid(side)
    got_w, got_b = m.backward_grid(side, g, loss_scale=16.0)
    for a, b in zip(got_w + got_b, ref):
        assert torch.equal(a, b)
    m.set_keep_first_layer(False)
    assert torch.equal(m.forward_grid(side), out_keep)
    m.set_keep_first_layer(True)
    # a voxel range (what a slab rank differentiates through)
    first, count = 96 * 100 * 7 + 13, 96 * 100 * 40
    m.set_keep_first_layer(False)
    r_w, r_b = m.backward_grid_range(side, first, count, g[first:first + count], loss_scale=16.0)
    r = [t.clone() for t in r_w + r_b]
    m.set_keep_first_layer(True)
    m.forward_grid_range(side, first, count)
    k_w, k_b = m.backward_grid_range(side, first, count, g[first:first + count], loss_scale=16.0)
    for a, b in zip(k_w + k_b, r):
        assert torch.equal(a, b)
    # another range than the kept one, and new weights: recomputation, same values
    o_w, o_b = m.backward_grid(side, g, loss_scale=16.0)
    for a, b in zip(o_w + o_b, ref):
        assert torch.equal(a, b)
    m.forward_grid(side)
    m.load_arrays(B, [w * 1.5 for w in Ws], bs)
    n_w, n_b = m.backward_grid(side, g, loss_scale=16.0)
    m.set_keep_first_layer(False)
    f_w, f_b = m.backward_grid(side, g, loss_scale=16.0)
    for a, b in zip(n_w + n_b, f_w + f_b):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_fused_adam_matches_torch_adam():
    import ctypes
    import torch
    from ndr_amd import _lib
    lib = _lib.load()
    torch.manual_seed(0)
    p0 = torch.randn(10007)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=3e-3, betas=(0.9, 0.99), eps=1e-8)
    p = p0.clone().cuda()
    mm, vv = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 6):
        g = torch.randn(10007) * (1.0 + step)
        ref.grad = g.clone()
        opt.step()
        gd = g.cuda()
        _lib.check(lib.vfem_adam_step(p.numel(), ctypes.c_void_p(p.data_ptr()), ctypes.c_void_p(gd.data_ptr()),
                                      ctypes.c_void_p(mm.data_ptr()), ctypes.c_void_p(vv.data_ptr()), 3e-3, 0.9, 0.99, 1e-8,
                                      step, None))
    torch.cuda.synchronize()
    assert float((p.cpu() - ref.detach()).abs().max()) < 2e-6


@pytest.mark.gpu
def test_trainable_mlp_fits_a_target_field_and_keeps_reference_state_dict_layout():
    """train_xdg-style loop on a toy objective: density = mlp(grid); loss(density).backward(); Adam.  The loss must drop,
    state_dict keys must be those of the reference module (net.<i>.weight / bias), homogeneous_init gives a uniform field"""
    import torch
    from ndr_amd.mlp import TrainableMLP
    torch.manual_seed(1)
    net = TrainableMLP(3, 1, 64, 4, 64, 2.0, output_act=torch.nn.Sigmoid())
    assert sorted(net.state_dict().keys()) == sorted(["net.%d.%s" % (i, k) for i in (0, 2, 4, 6) for k in ("weight", "bias")])
    side = (24, 16, 16)
    net.set_grid(side)
    net.homogeneous_init(0.0)
    rho0 = net.forward_grid()
    assert float((rho0 - 0.5).abs().max()) < 5e-3                       # sigmoid(0) everywhere
    x = torch.linspace(0, 1, side[0], device="cuda")[:, None, None].expand(side).reshape(-1)
    target = (x > 0.5).float() * 0.8 + 0.1
    losses = []
    for it in range(60):
        net.zero_grad()
        rho = net.forward_grid()
        loss = ((rho - target) ** 2).mean()
        loss.backward()
        net.adam_step(lr=3e-3)
        losses.append(float(loss.item()))
    assert losses[-1] < 0.25 * losses[0], (losses[0], losses[-1])


@pytest.mark.gpu
def test_grid_range_equals_slices_of_the_whole_grid():
    """what a rank of the slab decomposition evaluates (a contiguous voxel range) is bit-identical to that part of the
    whole-grid evaluation, float32 and float64 outputs"""
    import torch
    from ndr_amd.mlp import MLP
    z, es, nn_, nl, sig, Ws, bs = _load(FIXTURES[-1])
    m = MLP(3, 1, nn_, nl, es, float(z["sigma"][0]), output_act=torch.nn.Sigmoid() if sig else None)
    m.load_arrays(z["B"], Ws, bs)
    side = (20, 9, 13)
    full = m.forward_grid(side).reshape(-1)
    plane = side[1] * side[2]
    for x0, x1 in ((0, 5), (5, 6), (6, 20), (0, 20)):
        o64 = torch.empty((x1 - x0) * plane, dtype=torch.float64, device="cuda")
        part = m.forward_grid_range(side, x0 * plane, (x1 - x0) * plane, out_f64=o64)
        assert torch.equal(part, full[x0 * plane:x1 * plane])
        assert torch.equal(o64, part.double())
    with pytest.raises(RuntimeError):
        m.forward_grid_range(side, 19 * plane, 2 * plane)
