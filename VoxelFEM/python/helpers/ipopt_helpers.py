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

    # the four callbacks an NLP solver asks of this object (reference: ipopt_helpers.py:59-73): each sets the design variables,
    # then evaluates one quantity of the wrapped TopologyOptimizationProblem
    def _at(self, x, quantity):
        self.problem.setVars(x)
        return getattr(self.problem, quantity)()

    def objective(self, x):
        return self._at(x, "evaluateObjective")

    def gradient(self, x):
        return self._at(x, "evaluateObjectiveGradient")

    def constraints(self, x):
        return self._at(x, "evaluateConstraints")

    def jacobian(self, x):
        return self._at(x, "evaluateConstraintsJacobian")

    def intermediate(self, alg_mod, iter_count, obj_value, *solver_state):
        """per-iteration callback of the NLP solver: records the history the drivers read back (objective and density per
        iteration); the reference's version also prints a banner and tracks non-discreteness (ipopt_helpers.py:78-100)"""
        if self.recordingHistory:
            self.history.objective.append(obj_value)
            self.history.density.append(self.problem.getDensities())
        return True


def initializeIpoptProblem(TOP, previousHistory=None, recording=True):
    """(None, problemObj): only problemObj is used on the OC path (fem.py:47,55,80)"""
    wrapper = problemObjectWrapper(TOP, previousHistory)
    wrapper.setRecording(recording)
    return None, wrapper
