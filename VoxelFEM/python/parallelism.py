"""Drop-in for MeshFEM's ``parallelism`` python module (parallelism.cc): the reference caps its TBB worker pools here
(train_voxelfem.py:38-39).  The accelerated path has no host worker pool on the hot path, so the calls only record
the request."""
_state = {"max_num_tbb_threads": None, "gradient_assembly_num_threads": None, "hessian_assembly_num_threads": None}


def set_max_num_tbb_threads(n):
    if int(n) < 1:
        raise RuntimeError("number of threads must be positive")
    _state["max_num_tbb_threads"] = int(n)


def unset_max_num_tbb_threads():
    _state["max_num_tbb_threads"] = None


def set_gradient_assembly_num_threads(n):
    _state["gradient_assembly_num_threads"] = int(n)


def set_hessian_assembly_num_threads(n):
    _state["hessian_assembly_num_threads"] = int(n)
