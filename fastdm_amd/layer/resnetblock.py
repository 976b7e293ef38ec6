"""SDXL's ResNet block (reference fastdm/layer/resnetblock.py:9-50).

Convolutions are library calls in the reference too (`F.conv2d`); here they go to MIOpen through PyTorch-ROCm.  The
add / GroupNorm / SiLU chain around them is one fused HIP op (`fdm_group_norm`).  The time-embedding projection (M = batch rows) is the bf16 QLinear, i.e. the HBM-streaming GEMV
kernel.
"""
import torch
import torch.nn.functional as F

from fastdm_amd.kernel import fused
from fastdm_amd.layer.qlinear import QLinear

BF16 = torch.bfloat16


class ResnetBlock2D:
    def __init__(self, in_channels, out_channels, conv_shortcut=True, data_type=BF16, temb_channels=1280):
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1_gamma = self.norm1_beta = None
        self.conv1_weight = self.conv1_bias = None
        self.time_emb_proj = QLinear(temb_channels, out_channels, data_type=data_type)
        self.norm2_gamma = self.norm2_beta = None
        self.conv2_weight = self.conv2_bias = None
        self.conv_shortcut = conv_shortcut
        self.convshortcut_weight = self.convshortcut_bias = None

    @staticmethod
    def _gn_silu_conv3(x, gamma, beta, weight, bias, addend=None, res=None):
        """[x + addend[:, :, None, None] ->] GroupNorm(32, eps 1e-5) -> SiLU -> 3x3 convolution (stride 1, padding 1).  The element-wise
        part is one fused HIP op (kernel/fused.py:group_norm, NCHW or NHWC); the convolution is the HIP implicit GEMM for channels-last
        activations (kernel/fused.py:conv2d), MIOpen otherwise.  res: added behind the convolution (the block's skip connection)."""
        return fused.conv2d(fused.group_norm(x, 32, gamma, beta, 1e-05, addend=addend, silu=True), weight, bias, 1, 1, res=res)

    def forward(self, input_tensor, temb):
        h = self._gn_silu_conv3(input_tensor, self.norm1_gamma, self.norm1_beta, self.conv1_weight, self.conv1_bias)
        shift = self.time_emb_proj.forward(F.silu(temb))                   # [B, C_out]: per-channel offset from the time embedding
        skip = input_tensor if not self.conv_shortcut else fused.conv2d(input_tensor, self.convshortcut_weight, self.convshortcut_bias, 1)
        # skip + h: the add rides in the convolution's epilogue (channels-last + HIP convolution), same roundings as the separate add
        return self._gn_silu_conv3(h, self.norm2_gamma, self.norm2_beta, self.conv2_weight, self.conv2_bias, addend=shift.contiguous(), res=skip)
