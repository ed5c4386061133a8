"""Import shim for the reference drivers (they append VoxelFEM/python/helpers to sys.path and import these names,
fem.py:13): simulator set-up and the object that carries the optimisation history.  IPOPT itself is out of scope
(cyipopt absent), so `initializeIpoptProblem` returns no solver."""
import pyVoxelFEM  # noqa: F401  (the shim next door: puts the repository root on sys.path)
from ndr_amd.fem import initializeTensorProductSimulator  # noqa: F401,E402


class optimizationHistory:
    def __init__(self):
        self.objective, self.density, self.nondiscreteness = [], [], []


class problemObjectWrapper:
    """holder of the problem and its history; the drivers hang their adaptive-filter settings on it (fem.py:54-55)"""

    def __init__(self, problem, previousHistory=None):
        self.problem = problem
        self.history = previousHistory if previousHistory else optimizationHistory()
        self.recordingHistory = True

    def setRecording(self, recording):
        self.recordingHistory = recording


def initializeIpoptProblem(TOP, previousHistory=None, recording=True):
    """(None, problemObj): only problemObj is used on the OC path (fem.py:47,55,80)"""
    wrapper = problemObjectWrapper(TOP, previousHistory)
    wrapper.setRecording(recording)
    return None, wrapper
