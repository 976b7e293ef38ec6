"""SDXL's ResNet block (reference fastdm/layer/resnetblock.py:9-50).

Convolutions and group-norm are library calls in the reference too (`F.conv2d`, `F.group_norm`); here they go to MIOpen
through PyTorch-ROCm.  The time-embedding projection (M = batch rows) is the bf16 QLinear, i.e. the HBM-streaming GEMV
kernel.
"""
import torch
import torch.nn.functional as F

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

    def forward(self, input_tensor, temb):
        h = F.silu(F.group_norm(input_tensor, 32, self.norm1_gamma, self.norm1_beta, 1e-05))
        h = F.conv2d(h, self.conv1_weight, self.conv1_bias, 1, 1)
        h = h + self.time_emb_proj.forward(F.silu(temb))[:, :, None, None]
        h = F.silu(F.group_norm(h, 32, self.norm2_gamma, self.norm2_beta, 1e-05))
        h = F.conv2d(h, self.conv2_weight, self.conv2_bias, 1, 1)
        if self.conv_shortcut:
            input_tensor = F.conv2d(input_tensor, self.convshortcut_weight, self.convshortcut_bias, 1)
        return input_tensor + h
