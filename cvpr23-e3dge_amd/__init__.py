"""e3dge_amd -- MI355X-native volume-rendering hot path of E3DGE.

Host side is Python on PyTorch-ROCm (device memory, streams, torch.distributed); every kernel on the path
is hand-written HIP for gfx950 behind the C-ABI in include/e3dge_hip.h (cvpr23-e3dge_amd/csrc), loaded with
ctypes by `_lib`.  The public surface mirrors the reference's:

    e3dge_amd.op                 <-> project/models/op            (fused_leaky_relu, FusedLeakyReLU, upfirdn2d)
    e3dge_amd.volume_renderer    <-> project/utils/volume_renderer (VolumeFeatureRenderer, SirenGenerator, ...)
    e3dge_amd.stylesdf_model     <-> project/models/stylesdf_model (Decoder, Generator, G_pred_latents, ...)
    e3dge_amd.camera_utils       <-> project/utils/camera_utils    (generate_camera_params)

There is no CPU fallback on the product path: ops raise if the HIP library is missing or a tensor is not on
the GPU.  The CPU restatement used for checking lives in `oracle/` and is never imported from here.
"""
__version__ = "0.1.0"
