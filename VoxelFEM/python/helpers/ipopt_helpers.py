"""Helper used by the reference drivers (VoxelFEM/python/helpers/ipopt_helpers.py:7-57): simulator set-up and the
problem wrapper with its optimisation history.  The IPOPT branch itself is out of scope (cyipopt absent)."""
import pyVoxelFEM


class optimizationHistory:
    def __init__(self):
        self.objective = []
        self.density = []
        self.nondiscreteness = []


def initializeTensorProductSimulator(orderFEM, domainCorners, numberElements, uniformDensity, E0, Emin, SIMPExp,
                                     materialPath, bcsPath):
    TPS = pyVoxelFEM.TensorProductSimulator(orderFEM, domainCorners, numberElements)
    TPS.readMaterial(materialPath)
    TPS.setUniformDensities(uniformDensity)
    TPS.applyDisplacementsAndLoadsFromFile(bcsPath)
    TPS.E_0 = E0
    TPS.E_min = Emin
    TPS.gamma = SIMPExp
    return TPS


class problemObjectWrapper:
    def __init__(self, problem, previousHistory=[]):
        self.history = optimizationHistory() if previousHistory == [] else previousHistory
        self.problem = problem
        self.recordingHistory = True

    def setRecording(self, recording):
        self.recordingHistory = recording

    def objective(self, x):
        self.problem.setVars(x)
        return self.problem.evaluateObjective()

    def gradient(self, x):
        self.problem.setVars(x)
        return self.problem.evaluateObjectiveGradient()

    def constraints(self, x):
        self.problem.setVars(x)
        return self.problem.evaluateConstraints()

    def jacobian(self, x):
        self.problem.setVars(x)
        return self.problem.evaluateConstraintsJacobian()


def initializeIpoptProblem(TOP, previousHistory=[], recording=True):
    """Returns (nlp, problemObj).  Only problemObj (history, adaptive-filter attributes) is used on the OC path
    (fem.py:47,55,80); nlp is None because cyipopt is not available."""
    problemObj = problemObjectWrapper(TOP, previousHistory)
    problemObj.setRecording(recording)
    return None, problemObj
