"""mvsdet_amd: MI355X-native (gfx950) plane-sweep cost-volume hot path of MVSDet.

Scope: homography warp + variance cost volume, depth soft-max / top-k
plane selection, depth-weighted lifting of 2D features into the voxel grid -- as HIP kernels
behind a C ABI (include/mvsdet_hip.h), mirrored here with the reference's Python signatures.
The HIP library is loaded on first use; nothing in this package falls back to a CPU path.
"""
__version__ = "0.1.0"
