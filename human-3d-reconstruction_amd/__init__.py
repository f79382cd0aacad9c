"""h3d_amd -- MI355X (gfx950) native implementation of the multi_pose inference hot path of
Aaron20127/human-3d-reconstruction: DLA-34 + DCNv2 forward, heat-map decode, SMPL/LBS.

Host side is Python on PyTorch-ROCm (device memory, streams, torch.distributed only); all
compute goes through the C-ABI library `csrc/libh3d_hip.so` (include/h3d.h).  There is no CPU
fallback: importing the compute entry points without the built library raises.
"""
from . import synth  # noqa: F401
