"""pose2room_amd -- MI355X-native (gfx950) hot path of Pose2Room's P2RNet.

Sub-packages mirror the reference layout for the hot path only:

  pointnet2_ops   -> external/pointnet2_ops_lib/pointnet2_ops  (_ext, pointnet2_utils, pointnet2_modules)
  net_utils       -> net_utils/{nn_distance,nms}.py
  p2rnet          -> models/p2rnet (modules, loss, trainer)

All device compute of the ported ops goes through the C-ABI shared library
`libp2r_hip.so` (include/p2r_hip.h), loaded by `pose2room_amd._lib`.  There is no
CPU fallback: calling an op with the library missing, or with CPU tensors,
raises.
"""
__version__ = "0.1.0"
