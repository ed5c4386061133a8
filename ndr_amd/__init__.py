"""MI355X-native voxel-FEM topology-optimization hot path (drop-in for the `pyVoxelFEM` surface of
Nikronic/ndr).  See DESIGN.md for scope and INTEGRATION.md for the reference-side binding."""
__version__ = "0.1.0"
