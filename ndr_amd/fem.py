"""Driver-level entry points of the accelerated path with the names the reference's ``fem.py`` exports: the ground-truth
OC loop (``ground_truth_topopt``; here a thin wrapper over ``DesignLoop``) and the autograd bridge used by train_xdg
(``VoxelFEMFunction``), on top of ``ndr_amd.pyVoxelFEM``.  Plotting of the reference drivers is out of scope (SURVEY
8f-4); everything numerical keeps the reference's hard-coded settings (E0 = 1, Emin = 1e-4, tol 1e-4, one FMG cycle per
CG iteration, 2+2 sweeps)."""
import sys
import time

import numpy as np
import torch
import torch.autograd as autograd

from . import pyVoxelFEM


def initializeTensorProductSimulator(orderFEM, domainCorners, numberElements, uniformDensity, E0, Emin, SIMPExp,
                                     materialPath, bcsPath):
    """Same arguments as the reference helper (ipopt_helpers.py:7-15): simulator with material, BCs and SIMP law set."""
    tps = pyVoxelFEM.TensorProductSimulator(orderFEM, domainCorners, numberElements)
    tps.readMaterial(materialPath)
    tps.setUniformDensities(uniformDensity)
    tps.applyDisplacementsAndLoadsFromFile(bcsPath)
    tps.E_0, tps.E_min, tps.gamma = E0, Emin, SIMPExp
    return tps


class DesignLoop:
    """One ground-truth optimisation run (what fem.ground_truth_topopt sets up, fem.py:20-106) as an object: simulator,
    compliance objective, volume constraint, smoothing + projection filters and the optimality-criterion update.  All
    vectors stay in HBM between iterations; only the scalar compliance comes back per step.

    Reference behaviour kept: E0 = 1 and Emin = 1e-4 whatever the problem file says (fem.py:31-32), the solver settings of
    fem.py:64-70, the design starts at the uniform volume fraction, compliance is reported as f.u = 2 x evaluateObjective."""

    SOLVER = {"tol": 1e-4, "mgIterations": 1, "fullMultigrid": True, "zeroInit": False, "mgSmoothingIterations": 2}

    def __init__(self, material, bcs, order, corners, grid, simp_exponent, volume_fraction, mg_levels, use_multigrid=True):
        corners = [np.asarray(c, dtype=np.float64) for c in corners]
        self.order = list(order)
        self.tps = initializeTensorProductSimulator(self.order, corners, grid, volume_fraction, 1, 1e-4, simp_exponent,
                                                    material, bcs)
        if use_multigrid:
            self.objective = pyVoxelFEM.MultigridComplianceObjective(self.tps.multigridSolver(mg_levels))
            for name, value in self.SOLVER.items():
                setattr(self.objective, name, value)
        else:
            self.objective = pyVoxelFEM.ComplianceObjective(self.tps)
        self.problem = pyVoxelFEM.TopologyOptimizationProblem(
            self.tps, self.objective, [pyVoxelFEM.TotalVolumeConstraint(volume_fraction)],
            [pyVoxelFEM.SmoothingFilter(), pyVoxelFEM.ProjectionFilter()])
        self.history = []
        self.adaptive_filtering = None          # stored for the drivers, never read on the OC path (fem.py:54-55)

    def seed(self, design=None):
        """start from `design` (array or tensor of design variables) or from the simulator's current densities"""
        if design is None:
            design = self.tps.getDensities()
        elif isinstance(design, torch.Tensor):
            design = design.detach().cpu().numpy()
        self.problem.setVars(np.asarray(design, dtype=np.float64).reshape(-1))

    def compliance(self):
        return 2.0 * self.problem.evaluateObjective()

    def run(self, steps, log=None):
        oc = pyVoxelFEM.OCOptimizer(self.problem)
        clock = time.perf_counter()
        for k in range(steps):
            c = self.compliance()
            self.history.append(c)
            if log is not None:
                log.write('Total Steps: {:d}, Runtime: {:.1f}, Compliance loss {:.6f}\n'.format(k, time.perf_counter() - clock, c))
            clock = time.perf_counter()
            oc.step()
        return self.history

    def thresholded_compliance(self):
        """compliance of the design rounded to {0, 1} at 0.5 (utils.compute_binary_compliance_loss); the design is restored"""
        x = self.tps.getDensities()
        self.problem.setVars((x > 0.5).astype(np.float64))
        c = self.compliance()
        self.problem.setVars(x)
        return c


def ground_truth_topopt(MATERIAL_PATH, BC_PATH, orderFEM, domainCorners, gridDimensions, SIMPExponent, maxVolume,
                        optimizer, multigrid_levels, use_multigrid=True, adaptive_filtering=[1, 1, 1, 1],
                        max_iter=100, init=None, obj_history=False, verbose=True, **kwargs):
    """Signature and return value of the reference's fem.ground_truth_topopt (fem.py:20-106); optimizer 'OC' only (the
    L-BFGS branch is IPOPT, out of scope).  Returns (tps | densities for 2-D, final compliance, thresholded compliance
    [, history])."""
    if optimizer != 'OC':
        raise ValueError('Optimizer {} is unknown or not implemented.'.format(optimizer))
    loop = DesignLoop(MATERIAL_PATH, BC_PATH, orderFEM, domainCorners, gridDimensions, SIMPExponent, maxVolume,
                      multigrid_levels, use_multigrid)
    loop.adaptive_filtering = adaptive_filtering
    if init is not None:
        loop.seed(init)
    loop.seed()
    loop.run(max_iter, sys.stderr if verbose else None)
    binary = loop.thresholded_compliance()
    result = (loop.tps if len(orderFEM) == 3 else loop.tps.getDensities(), loop.compliance(), binary)
    return result + (loop.history,) if obj_history else result


def _compliance_and_sensitivity(top, densities):
    """f.u and d(f.u)/d(rho) for physical densities `densities` (any shape, float): the solver reads a device tensor in
    place, a host tensor through numpy; the sensitivity comes back in float32 on the input's device (fem.py:125)"""
    rho = densities.detach().to(torch.float64).reshape(-1)
    if rho.is_cuda:
        top.setVars(rho)
        value = 2.0 * top.evaluateObjective()
        sens = top.evaluateObjectiveGradient_device().to(torch.float32)
    else:
        top.setVars(rho.numpy())
        value = 2.0 * top.evaluateObjective()
        sens = torch.from_numpy(top.evaluateObjectiveGradient().astype(np.float32))
    return value, sens.reshape(densities.shape)


class VoxelFEMFunction(autograd.Function):
    """Compliance of the predicted densities as an autograd node (the reference's fem.VoxelFEMFunction, fem.py:109-134):
    forward solves and caches the sensitivity, backward scales it by the incoming gradient."""

    @staticmethod
    def forward(ctx, densities, top):
        value, sens = _compliance_and_sensitivity(top, densities)
        ctx.save_for_backward(sens)
        return torch.tensor(value, device=densities.device, dtype=torch.float32)

    @staticmethod
    def backward(ctx, grad_output):
        return ctx.saved_tensors[0] * grad_output, None


def save_for_interactive_vis(density, grid_dimensions, title, visualize, path):
    """utils.save_for_interactive_vis (utils.py:350-376): cell-data ``.vtr`` of the density field for ParaView"""
    if not visualize:
        return None
    from . import io
    if isinstance(density, torch.Tensor):
        density = density.detach().cpu().numpy()
    elif hasattr(density, "getDensities"):
        density = density.getDensities()
    elif not isinstance(density, np.ndarray):
        raise TypeError('Datatype "{}" not understood.\n'.format(type(density)))
    density = density.reshape(grid_dimensions)
    fname = io.grid_to_vtr(path + title, np.arange(density.shape[0] + 1), np.arange(density.shape[1] + 1),
                           np.arange(density.shape[2] + 1), cellData={"data": density.copy()})
    sys.stderr.write("{}.vtr has been saved to {}.\n".format(title, path))
    return fname


def save_densities_mesh(tps, path):
    """utils.save_densities, 3-D branch (utils.py:313-316): Gmsh field file of the simulator's densities"""
    from . import io
    mfw = io.MSHFieldWriter(path, *tps.getMesh())
    mfw.addField("density", tps.getDensities())
    return path


# ------------------------------------------------------------------------------------------------------
# volume-constraint satisfiers of the train_xdg closure (fem.py:137-307): hard modes shift the logits by the scalar b(x)
# that solves mean(projection(x + b)) = V with the implicit-function derivative; soft modes return a penalty term.
# torch tensor ops only (elementwise + reductions on the density field, device-resident).
# ------------------------------------------------------------------------------------------------------
class _ShiftToMean(autograd.Function):
    """b(x) with mean(projection(x + b)) = average, found by bisection (at most 128 halvings, tolerance 1e-12 on the
    bracket, fem.FindRootFunction); backward uses db/dx_i = -(df/dx_i) / (df/db)."""

    @staticmethod
    def forward(ctx, x, average, lower, upper, projection, dprojection, allsum=None):
        # allsum(t): sum of a small tensor over the ranks that hold the other parts of x (None: x is the whole field)
        lo, hi = float(lower), float(upper)
        avg = float(average)
        red = (lambda t: t) if allsum is None else allsum
        with torch.no_grad():
            n = float(red(x.new_tensor([float(x.numel())]))[0])
            step = 0
            while step < 128 and hi - lo >= 1e-12:
                mid = 0.5 * (lo + hi)
                if float(red(projection(x + mid).sum().reshape(1))[0]) / n - avg > 0:
                    hi = mid
                else:
                    lo = mid
                step += 1
            b = 0.5 * (lo + hi)
            d = dprojection(x + b)
            dmean = float(red(d.sum().reshape(1))[0]) / n
        ctx.save_for_backward(d)
        ctx.n, ctx.dmean, ctx.red = n, dmean, red
        return x.new_tensor(b)

    @staticmethod
    def backward(ctx, grad_output):
        (d,) = ctx.saved_tensors
        # f = mean(P(x + b)) - avg:  df/dx_i = P'(x_i + b) / n,  df/db = mean P'; b is shared by all parts of x, so the
        # incoming gradient (dL/db) is the sum over the ranks
        g = ctx.red(grad_output.reshape(1).clone())[0]
        return -(d / ctx.n) / ctx.dmean * g, None, None, None, None, None, None


def logit(p):
    p = torch.clamp(torch.as_tensor(p, dtype=torch.float32), 0, 1)
    return torch.log(p) - torch.log1p(-p)


def _with_constrained_mean(x, average, projection, dprojection, allsum=None, allmax=None):
    lg = logit(average).to(x.device)
    xmax, xmin = x.max().detach().reshape(1), (-x.min()).detach().reshape(1)
    if allmax is not None:
        xmax, xmin = allmax(xmax), allmax(xmin)
    b = _ShiftToMean.apply(x, average, lg - xmax[0], lg + xmin[0], projection, dprojection, allsum)
    return projection(x + b)


def sigmoid_with_constrained_mean(x, average, projection=torch.sigmoid, allsum=None, allmax=None):
    """fem.sigmoid_with_constrained_mean (fem.py:213-229); `allsum` / `allmax` reduce a small tensor over the ranks when x is
    one rank's part of a sharded field (the mean constraint is on the whole field)"""
    if projection is torch.sigmoid:
        dproj = lambda t: torch.sigmoid(t) * (1 - torch.sigmoid(t))
    else:
        def dproj(t):
            t = t.detach().requires_grad_(True)
            with torch.enable_grad():
                return autograd.grad(projection(t).sum(), t)[0]
    return _with_constrained_mean(x, average, projection, dproj, allsum, allmax)


def physical_density(x, maxVolume):
    return sigmoid_with_constrained_mean(x, maxVolume)


def compute_volume_loss_scaler(compliance_loss, volume_loss, mode='clip', constant=500.):
    """fem.compute_volume_loss_scaler (fem.py:312-334)"""
    with torch.no_grad():
        scaler = compliance_loss / volume_loss
        if mode == 'clip':
            return torch.clamp_max(scaler, max=constant) if scaler >= constant else scaler
        if mode == 'equalize':
            return scaler
    raise ValueError('The mode "{}" does not exist'.format(mode))


def type_of_volume_constaint_satisfier(mode):
    hard = {'constrained_sigmoid': True, 'constrained_projection': True, 'add_mean': False, 'one_sided_max': False,
            'maxed_barrier': False, 'thresholded_barrier': False}
    if mode not in hard:
        raise ValueError('The mode "{}" does not exist'.format(mode))
    return hard[mode]


def satisfy_volume_constraint(density, max_volume, compliance_loss=None, mode='constrained_sigmoid', scaler_mode='clip',
                              constant=500., **kwargs):
    """fem.satisfy_volume_constraint (fem.py:257-309): hard modes return the constrained density, soft modes the weighted
    volume penalty to add to the compliance"""
    max_volume = torch.as_tensor(max_volume, dtype=density.dtype, device=density.device)
    if mode == 'constrained_sigmoid':
        return sigmoid_with_constrained_mean(density, max_volume, torch.sigmoid)
    if mode == 'constrained_projection':
        projection = kwargs.get('projection')
        if projection is None:
            raise ValueError("constrained_projection needs projection=<callable>")
        return sigmoid_with_constrained_mean(density, max_volume, projection)
    current = density.mean()
    zero = torch.zeros_like(current)
    eps = 1e-7
    if mode == 'add_mean':
        volume_loss = torch.abs(current - max_volume)
    elif mode == 'one_sided_max':
        volume_loss = torch.maximum(current - max_volume, zero) ** 2
    elif mode == 'maxed_barrier':
        volume_loss = torch.maximum(-torch.log(1 + max_volume + eps - current), zero)
    elif mode == 'thresholded_barrier':
        a = (1 + max_volume + eps - current).detach() if bool(current <= max_volume) else torch.ones_like(current)
        volume_loss = torch.log(a / (1 + max_volume + eps - current)) ** 2
    else:
        raise ValueError('The mode "{}" does not exist'.format(mode))
    return volume_loss * compute_volume_loss_scaler(compliance_loss, volume_loss, scaler_mode, constant)


def homogeneous_init(model, constant):
    """fem.homogeneous_init (fem.py:350-374) for modules with Linear layers: every Linear whose weight has a dimension of
    size 1 or 2 (the output layer) gets weight ~ N(0, 1e-4) and bias = constant"""
    for m in model.modules():
        if isinstance(m, torch.nn.Linear) and (1 in m.weight.shape or 2 in m.weight.shape):
            with torch.no_grad():
                m.weight.normal_(0.0, 1e-4)
                m.bias.fill_(float(constant))
