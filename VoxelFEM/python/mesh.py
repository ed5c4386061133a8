"""Drop-in for the subset of MeshFEM's ``mesh`` python module the reference drivers use on this path
(utils.py:315-316, 411-412): ``MSHFieldWriter`` and ``MSHFieldParser3``."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ndr_amd.io import MSHFieldParser3, MSHFieldWriter  # noqa: E402,F401
