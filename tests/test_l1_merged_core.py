"""not gpu: the per-mirror-class evaluation of a level-1 node row (ndr_amd/csrc/l1_merged_core.h, the arithmetic of
kernels_l1_merged.hip) compiled for the host and compared with the direct sum over the 8 incident elements x 8 children with
mirror-image child matrices (MultigridSolver.hh:199-220, 639-657): random symmetric cK0[0], random moduli (one case with a
whole side of zero moduli, as at a grid face), random neighbour values."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_merged_rows_equal_the_sum_over_elements(tmp_path):
    exe = str(tmp_path / "l1_merged_host")
    subprocess.run(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "l1_merged_host.cpp")], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert float(out.stdout.split()[-1]) < 1e-12


def test_numpy_identity():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "l1_merged_check.py")], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr
