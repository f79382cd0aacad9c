"""h3d_amd -- MI355X (gfx950) native implementation of the multi_pose inference hot path of
Aaron20127/human-3d-reconstruction: DLA-34 + DCNv2 forward, heat-map decode, SMPL/LBS.

Host side is Python on PyTorch-ROCm (device memory, streams, torch.distributed only); all
compute goes through the C-ABI library `csrc/libh3d_hip.so` (include/h3d.h).  There is no CPU
fallback: using a compute entry point without the built library raises.

Reference-named entry points (same names / argument meaning as the reference's src/lib):
    models.model.dla_net                      -> h3d_amd.model.dla_net
    models.DCNv2.dcn_v2.{dcn_v2_conv,DCNv2,DCN}, _ext.dcn_v2_forward -> h3d_amd.dcn_v2.*
    models.decode.{_nms,_topk,_topk_channel,multi_pose_decode,ctdet_decode} -> h3d_amd.decode.*
    models.utils.{_sigmoid,_gather_feat,_transpose_and_gather_feat} -> h3d_amd.utils.*
    utils.post_process.multi_pose_post_process -> h3d_amd.detector.multi_pose_post_process
"""
from . import synth  # noqa: F401

__all__ = ["synth", "arch", "model", "engine", "decode", "utils", "dcn_v2", "smpl", "detector"]
