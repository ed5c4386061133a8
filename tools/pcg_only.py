import sys, time
import torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from bench import pcg_rate
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lv = int(sys.argv[2]) if len(sys.argv) > 2 else 5
print(pcg_rate((n, n, n), lv, ([0, 0, 0], [2, 1, 1])))
