"""Attentional scale-sequence fusion blocks: Zoom_cat, ScalSeq, Add (drop-in for reference
nn/extra_modules/block.py:3402-3484)."""
from __future__ import annotations

import torch
import torch.nn as nn

from ...hip import DY_ACT_LEAKY
from ...hip.engine import BN3D_EPS, BN3D_MOM
from ...hip.runtime import HipModule
from ..modules.conv import Conv

__all__ = ("Zoom_cat", "ScalSeq", "Add")


class Zoom_cat(HipModule):
    """cat(maxpool+avgpool of the fine map, middle map, nearest-upsampled coarse map) (reference :3402-3412)."""

    def forward_act(self, xs, out=None):
        return self.rt.eng.zoom_cat(list(xs), out)


class ScalSeq(HipModule):
    """Scale-sequence feature fusion (reference :3414-3443): 1x1 Convs on the two coarser levels, nearest resize to the
    finest, Conv3d(1x1x1)+BatchNorm3d+LeakyReLU(0.1) over the 3-deep stack, max over depth."""

    def __init__(self, inc, channel):
        super().__init__()
        if channel != inc[0]:
            self.conv0 = Conv(inc[0], channel, 1)
        self.conv1 = Conv(inc[1], channel, 1)
        self.conv2 = Conv(inc[2], channel, 1)
        self.conv3d = nn.Conv3d(channel, channel, kernel_size=(1, 1, 1))
        self.bn = nn.BatchNorm3d(channel)
        self.act = nn.LeakyReLU(0.1)
        self.pool_3d = nn.MaxPool3d(kernel_size=(3, 1, 1))

    def _build_specs(self, rt):
        sp = rt.make_spec((id(self), "conv3d"), self.conv3d, None, DY_ACT_LEAKY, 1, 1, name="ScalSeq.conv3d")
        sp.bn3d = dict(weight=self.bn.weight.data, bias=self.bn.bias.data, running_mean=self.bn.running_mean,
                       running_var=self.bn.running_var)
        sp.coef3d = rt.eng.f32(4 * sp.cout)
        sp.bwdcoef3d = rt.eng.f32(2 * sp.cout)
        sp.gbn3d = (rt.gviews[id(self.bn.weight)], rt.gviews[id(self.bn.bias)])

    def forward_act(self, xs, out=None, res=None):
        p3, p4, p5 = xs
        if hasattr(self, "conv0"):
            p3 = self.conv0.forward_act(p3)
        p4 = self.conv1.forward_act(p4)
        p5 = self.conv2.forward_act(p5)
        sp = self.rt.specs[(id(self), "conv3d")]
        return self.rt.eng.scalseq(sp, sp.bn3d, sp.coef3d, sp.bwdcoef3d, sp.gbn3d, [p3, p4, p5], out, res=res)


class Add(HipModule):
    """Element-wise sum of the inputs (reference :3479-3484)."""

    def forward_act(self, xs, out=None):
        return self.rt.eng.add(list(xs), out)
