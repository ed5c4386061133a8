#!/usr/bin/env python3
"""Neural density field (Fourier-feature MLP) trained against the compliance solve on the MI355X path: the command line of the
reference's training/train_xdg.py for the flags that matter to the hot path (--jid --grid --prob --v0 --mgl --vcs --es --nn --nl
--lr --iter --cs --sigma; run from the repository root).  Every step: MLP logits -> volume-constraint satisfier -> compliance
through the multigrid-PCG solve (autograd node with the device sensitivities) -> backward through the MLP -> Adam.
    python training/train_xdg.py --jid demo --grid "[64, 32, 32]" --prob problems/3d/bridge.json --v0 0.4 --mgl 3 --sigma 3 --iter 50
The filters of the reference's closure (kornia Gaussian smoothing etc.) are out of scope and not applied."""
import argparse
import ast
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--jid', default='run')
    ap.add_argument('--grid', help='grid dimensions as a list, e.g. "[64, 32, 32]" (default: the problem file\'s)')
    ap.add_argument('--prob', required=True)
    ap.add_argument('--v0', help='volume fraction (default: the problem file\'s)')
    ap.add_argument('--mgl', default=2)
    ap.add_argument('--vcs', default='constrained_sigmoid', help='volume-constraint satisfier (fem.satisfy_volume_constraint modes)')
    ap.add_argument('--es', default=1024, help='Fourier features (embedding size)')
    ap.add_argument('--nn', default=512, help='neurons per hidden layer')
    ap.add_argument('--nl', default=4, help='layers')
    ap.add_argument('--lr', default=3e-4)
    ap.add_argument('--iter', default=5000)
    ap.add_argument('--cs', default=100, help='a weight checkpoint every iter/cs steps')
    ap.add_argument('--sigma', required=True, help='scale of the Fourier-feature Gaussian')
    ap.add_argument('--out', default='logs')
    ap.add_argument('--mlp_precision', default='fp32', choices=['fp16', 'fp32'],
                    help='fp32: the reference network\'s precision (fused kernel with split fp16 operands); fp16: plain fp16 operands, 3x faster')
    args = ap.parse_args(argv)
    from ndr_amd import fem, pyVoxelFEM
    from ndr_amd.mlp import TrainableMLP

    with open(args.prob) as fh:
        cfg = json.load(fh)
    grid = tuple(ast.literal_eval(args.grid)) if args.grid else tuple(cfg['gridDimensions'])
    v0 = float(args.v0) if args.v0 is not None else cfg['maxVolume'][0]
    torch.manual_seed(cfg.get('seed', 88))
    hard = fem.type_of_volume_constaint_satisfier(args.vcs)
    tps = fem.initializeTensorProductSimulator(cfg['orderFEM'], cfg['domainCorners'], list(grid), v0, 1, 1e-4, 3,
                                               cfg['MATERIAL_PATH'], cfg['BC_PATH'])          # train_xdg forces SIMP exponent 3
    objective = pyVoxelFEM.MultigridComplianceObjective(tps.multigridSolver(int(args.mgl)))
    for name, value in fem.DesignLoop.SOLVER.items():
        setattr(objective, name, value)
    top = pyVoxelFEM.TopologyOptimizationProblem(tps, objective, [pyVoxelFEM.TotalVolumeConstraint(v0)], [])
    net = TrainableMLP(3, 1, int(args.nn), int(args.nl), int(args.es), float(args.sigma),
                       output_act=None if hard else torch.nn.Sigmoid())
    net.kernel.precision = args.mlp_precision
    net.set_grid(grid)
    fem.homogeneous_init(net, v0)
    max_volume = torch.tensor(v0, device="cuda")
    steps, every = int(args.iter), max(1, int(args.iter) // max(1, int(args.cs)))
    wdir = os.path.join(args.out, 'weights', 'ff', str(args.jid))
    os.makedirs(wdir, exist_ok=True)
    history, start = [], time.perf_counter()
    for step in range(steps):
        net.zero_grad()
        density = net.forward_grid().view(grid)
        if hard:
            density = fem.satisfy_volume_constraint(density, max_volume, mode=args.vcs)
        else:
            density = torch.clamp(density, 0.0, 1.0)
        loss = fem.VoxelFEMFunction.apply(density.flatten(), top)
        if not hard:
            loss = loss + fem.satisfy_volume_constraint(density, max_volume, compliance_loss=loss.detach(), scaler_mode='clip',
                                                        constant=1500, mode=args.vcs)
        loss.backward()
        net.adam_step(lr=float(args.lr))
        history.append(float(loss.detach()))
        sys.stderr.write('Total Steps: {:d}, Resolution Steps: {:d}, Compliance loss {:.6f}\n'.format(step + 1, step, history[-1]))
        if (step + 1) % every == 0 or step + 1 == steps:
            torch.save({'model_state_dict': net.state_dict(), 'B': net.B, 'step': step + 1}, os.path.join(wdir, '{}_iter{}.pt'.format(args.jid, step + 1)))
    with open(os.path.join(wdir, '{}_loss.json'.format(args.jid)), 'w') as fh:
        json.dump(history, fh)
    sys.stderr.write('\nOverall runtime: {}\n'.format(time.perf_counter() - start))
    return history


if __name__ == '__main__':
    main()
