#!/usr/bin/env python3
"""Ground-truth topology optimisation on the MI355X path: the command line of the reference's training/train_voxelfem.py
(--jid --grid --prob --v0 --mgl --iter --optim --af; run from the repository root), the solve through ndr_amd.fem.
    python training/train_voxelfem.py --jid demo --grid "[128, 64, 64]" --prob problems/3d/cantilever_flexion.json --v0 0.5 --mgl 3 --iter 20
Prints the reference's progress lines ("Total Steps: k, Runtime: s, Compliance loss c") to stderr and writes, under
logs/{loss,densities}/gt/<jid>/, the compliance history (JSON) and the final density field (.vtr for 3-D grids)."""
import argparse
import ast
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--jid', default='run', help='experiment id (names the output directories)')
    ap.add_argument('--grid', help='grid dimensions as a list, e.g. "[40, 20, 10]" (default: the problem file\'s)')
    ap.add_argument('--prob', required=True, help='problem JSON (material, BCs, domain, SIMP exponent ...)')
    ap.add_argument('--v0', help='volume fraction (default: the problem file\'s)')
    ap.add_argument('--mgl', default=2, help='multigrid coarsening levels (every grid dimension must be divisible by 2^mgl; 0: direct-solve objective)')
    ap.add_argument('--iter', default=5000, help='number of optimality-criterion iterations')
    ap.add_argument('--optim', default='OC', help='only "OC" (the L-BFGS branch of the reference is IPOPT)')
    ap.add_argument('--af', default="[1, 1, 1, 1]", help='adaptive-filtering settings (stored, not used on the OC path)')
    ap.add_argument('--out', default='logs', help='base directory of the outputs')
    args = ap.parse_args(argv)
    from ndr_amd import fem

    with open(args.prob) as fh:
        cfg = json.load(fh)
    grid = ast.literal_eval(args.grid) if args.grid else cfg['gridDimensions']
    v0 = float(args.v0) if args.v0 is not None else cfg['maxVolume'][0]
    levels = int(args.mgl)
    sys.stderr.write('VoxelFEM problem configs: {}\nMultigrid levels: {}\n'.format(dict(cfg, gridDimensions=grid, maxVolume=[v0]), levels))
    np.random.seed(cfg.get('seed', 88))
    start = time.perf_counter()
    result, final, binary, history = fem.ground_truth_topopt(
        cfg['MATERIAL_PATH'], cfg['BC_PATH'], cfg['orderFEM'], cfg['domainCorners'], grid, cfg['SIMPExponent'], v0,
        optimizer=args.optim, multigrid_levels=levels, use_multigrid=levels > 0, adaptive_filtering=ast.literal_eval(args.af),
        max_iter=int(args.iter), obj_history=True)
    sys.stderr.write('Final step, Compliance loss {:.6f}, Binary Compliance loss {:.6f} \n'.format(final, binary))
    title = '{}_voxelfem_optim-{}_{}_{}_{}_Vol{}'.format(args.jid, args.optim, 'x'.join(str(g) for g in grid), args.iter, cfg['problem_name'], v0)
    for sub in ('loss', 'densities'):
        os.makedirs(os.path.join(args.out, sub, 'gt', str(args.jid)), exist_ok=True)
    with open(os.path.join(args.out, 'loss', 'gt', str(args.jid), title + '.json'), 'w') as fh:
        json.dump({'compliance': history, 'final': final, 'binary': binary}, fh)
    if len(grid) == 3:
        fem.save_for_interactive_vis(result, grid, title, True, os.path.join(args.out, 'densities', 'gt', str(args.jid)) + os.sep)
    sys.stderr.write('\nOverall runtime: {}\n'.format(time.perf_counter() - start))
    return history


if __name__ == '__main__':
    main()
