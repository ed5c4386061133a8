"""Drop-in module at the path the reference drivers add to sys.path (fem.py:10-11): forwards to the
MI355X-native implementation."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ndr_amd.pyVoxelFEM import *            # noqa: F401,F403,E402
from ndr_amd.pyVoxelFEM import detail       # noqa: F401,E402
