"""Driver-level mirror of the reference's ``fem.py`` for the accelerated path: the ground-truth OC loop
(``fem.ground_truth_topopt``, fem.py:20-106) and the autograd bridge used by train_xdg
(``fem.VoxelFEMFunction``, fem.py:109-134), on top of ``ndr_amd.pyVoxelFEM``.  Plotting / .vtr export of the
reference drivers is out of scope; everything numerical keeps the reference's call sequence and its
hard-coded settings (E0 = 1, Emin = 1e-4, tol 1e-4, one FMG cycle per CG iteration, 2+2 sweeps)."""
import sys
import time

import numpy as np
import torch
import torch.autograd as autograd

from . import pyVoxelFEM


class _History:
    def __init__(self):
        self.objective = []


class _ProblemObj:
    def __init__(self, top):
        self.problem = top
        self.history = _History()
        self.beta_interval = self.beta_scaler = self.radius_interval = self.radius_scaler = None


def initializeTensorProductSimulator(orderFEM, domainCorners, numberElements, uniformDensity, E0, Emin, SIMPExp,
                                     materialPath, bcsPath):
    """VoxelFEM/python/helpers/ipopt_helpers.py:7-15"""
    tps = pyVoxelFEM.TensorProductSimulator(orderFEM, domainCorners, numberElements)
    tps.readMaterial(materialPath)
    tps.setUniformDensities(uniformDensity)
    tps.applyDisplacementsAndLoadsFromFile(bcsPath)
    tps.E_0 = E0
    tps.E_min = Emin
    tps.gamma = SIMPExp
    return tps


def ground_truth_topopt(MATERIAL_PATH, BC_PATH, orderFEM, domainCorners, gridDimensions, SIMPExponent, maxVolume,
                        optimizer, multigrid_levels, use_multigrid=True, adaptive_filtering=[1, 1, 1, 1],
                        max_iter=100, init=None, obj_history=False, verbose=True, **kwargs):
    """fem.ground_truth_topopt (fem.py:20-106), optimizer 'OC' only (the L-BFGS branch needs cyipopt)."""
    E0, Emin = 1, 1e-4                                             # fem.py:31-32 (the JSON values are ignored)
    constraints = [pyVoxelFEM.TotalVolumeConstraint(maxVolume)]
    filters = [pyVoxelFEM.SmoothingFilter(), pyVoxelFEM.ProjectionFilter()]
    domain = [np.asarray(domainCorners[0], dtype=np.float64), np.asarray(domainCorners[1], dtype=np.float64)]
    tps = initializeTensorProductSimulator(orderFEM, domain, gridDimensions, maxVolume, E0, Emin, SIMPExponent,
                                           MATERIAL_PATH, BC_PATH)
    if use_multigrid:
        objective = pyVoxelFEM.MultigridComplianceObjective(tps.multigridSolver(multigrid_levels))
    else:
        objective = pyVoxelFEM.ComplianceObjective(tps)
    top = pyVoxelFEM.TopologyOptimizationProblem(tps, objective, constraints, filters)
    problemObj = _ProblemObj(top)
    if adaptive_filtering is not None:
        (problemObj.beta_interval, problemObj.beta_scaler, problemObj.radius_interval,
         problemObj.radius_scaler) = adaptive_filtering
    if init is not None:
        init = np.asarray(init.detach().cpu().numpy() if isinstance(init, torch.Tensor) else init, dtype=np.float64)
        top.setVars(init.flatten())
    if use_multigrid:
        objective.tol = 1e-4                                       # fem.py:64-70
        objective.mgIterations = 1
        objective.fullMultigrid = True
        objective.zeroInit = False
        objective.mgSmoothingIterations = 2
    if optimizer != 'OC':
        raise ValueError('Optimizer {} is unknown or not implemented.'.format(optimizer))
    oco = pyVoxelFEM.OCOptimizer(top)
    top.setVars(tps.getDensities())
    iter_start_time = 0
    for idx in range(max_iter):
        iter_time = time.perf_counter() - iter_start_time
        objective_value = 2.0 * top.evaluateObjective()
        problemObj.history.objective.append(objective_value)
        if verbose:
            sys.stderr.write('Total Steps: {:d}, Runtime: {:.1f}, Compliance loss {:.6f}\n'.format(idx, iter_time, objective_value))
        iter_start_time = time.perf_counter()
        oco.step()
    x0 = tps.getDensities()
    density_binary = (x0 > 0.5) * 1.0                              # utils.compute_binary_compliance_loss
    top.setVars(density_binary.astype(np.float64))
    binary_objective = 2.0 * top.evaluateObjective()
    top.setVars(x0)
    out = (tps if len(orderFEM) == 3 else tps.getDensities(), 2.0 * top.evaluateObjective(), binary_objective)
    return out + (problemObj.history.objective,) if obj_history else out


class VoxelFEMFunction(autograd.Function):
    """fem.VoxelFEMFunction (fem.py:109-134): compliance of the predicted densities as an autograd node.  Densities
    may live on the GPU; the sensitivity is returned in float32 on the input's device, as the reference does."""

    @staticmethod
    def forward(ctx, densities, top):
        dev = densities.device
        if densities.is_cuda:                                      # zero-copy: the solver consumes the device tensor
            top.setVars(densities.detach().to(torch.float64).reshape(-1))
            output_objective = 2.0 * top.evaluateObjective()
            grad = top.evaluateObjectiveGradient_device().to(torch.float32).reshape(densities.shape)
        else:
            top.setVars(densities.detach().to(torch.float64).numpy())
            output_objective = 2.0 * top.evaluateObjective()
            grad = torch.from_numpy(top.evaluateObjectiveGradient().astype(np.float32)).reshape(densities.shape)
        ctx.save_for_backward(grad)
        return torch.tensor(output_objective, device=dev).float()

    @staticmethod
    def backward(ctx, grad_output):
        (grad,) = ctx.saved_tensors
        return grad * grad_output, None


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
