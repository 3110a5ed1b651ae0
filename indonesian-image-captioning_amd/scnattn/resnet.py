"""ResNet-152 trunk (v1.5: stride on the 3x3 conv) built from plain torch.nn layers, laid out so that
the module tree -- and therefore every state_dict key -- equals
``nn.Sequential(*list(torchvision.models.resnet152().children())[:-2])`` as used by the reference
(models/encoders/caption.py:17-22): children 0..7 = conv1, bn1, relu, maxpool, layer1..layer4.

torchvision is not installed in this image (nor on the GPU box), and the reference's
``pretrained=True`` needs a download, so weights are random-init by torchvision's own recipe
(kaiming-normal fan-out convs, BN gamma=1 beta=0) unless a state_dict is loaded.  On MI355X the
convolutions run on MIOpen's MFMA kernels through PyTorch-ROCm; channels-last memory format is used so
the 1x1 convolutions are plain GEMMs over contiguous channels."""
import os

import torch
from torch import nn


def configure_miopen():
    """This image ships no gfx950 MIOpen find/kernel database, so MIOpen's default exhaustive find
    JIT-compiles every applicable solver for each of the ~150 conv configs of ResNet-152 (many minutes on
    a fresh box).  FAST find mode picks one solver per config from the built-in heuristics (first step
    ~3 s); measured on MI355X it is also the fastest steady state with NCHW fp32 (50 ms fwd+bwd at B=32
    vs 320 ms channels-last, 120 ms GEMM-only).  Respect anything the user has already exported."""
    os.environ.setdefault("MIOPEN_FIND_MODE", "2")
    os.environ.setdefault("MIOPEN_LOG_LEVEL", "3")


class FusedBatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d (same parameters, buffers and state_dict keys) whose CUDA forward runs the fused
    HIP kernels of csrc/batchnorm.hip: batch statistics + normalise + optional residual add + optional
    ReLU in two passes (torch/MIOpen: 3 BN kernels + add + relu), and the matching two-pass backward.
    CPU tensors (structure tests) take torch's own ops."""

    counter_managed = False   # True: the trunk bumps all num_batches_tracked counters with one add
    use_fused = True          # class-wide switch (tests compare the fused kernels with torch's own ops)

    def forward(self, x, residual=None, relu=False):
        if not x.is_cuda or not FusedBatchNorm2d.use_fused or self.weight is None or not self.track_running_stats or self.momentum is None:
            y = super().forward(x)
            if residual is not None:
                y = y + residual
            return torch.relu(y) if relu else y
        from . import functional as SF
        if self.training and not self.counter_managed and self.num_batches_tracked is not None:
            self.num_batches_tracked.add_(1)
        return SF.bn_act(x, residual, self.weight, self.bias, self.running_mean, self.running_var, self.training,
                         self.momentum, self.eps, relu)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = FusedBatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = FusedBatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, kernel_size=1, bias=False)
        self.bn3 = FusedBatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)   # kept for module-tree parity; the BN kernels apply it
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        # fp32 training-mode maps on the GPU: the whole block as ONE autograd node on the hand-written 1x1-convolution
        # kernels with the BatchNorm statistics / normalisation fused into them (scnattn/conv.py)
        from . import conv as _conv
        if _conv.usable(self, x):
            return _conv.bottleneck(self, x)
        if x.dtype == torch.bfloat16:        # the mixed-precision trunk (bf16 maps from scnattn/stem.py under bf16 autocast)
            from . import conv16 as _c16
            if _c16.usable(self, x):
                return _c16.bottleneck(self, x)
        identity = x if self.downsample is None else self.downsample(x)
        out = self.bn1(self.conv1(x), relu=True)
        out = self.bn2(self.conv2(out), relu=True)
        return self.bn3(self.conv3(out), residual=identity, relu=True)


def manage_bn_counters(trunk):
    """Re-home every BatchNorm's ``num_batches_tracked`` into one flat int64 tensor (state_dict keys and
    values unchanged) so that a training forward bumps all of them with ONE kernel instead of one tiny
    launch per layer (157 launches, 0.7 ms per step on MI355X).  Returns the flat tensor."""
    bns = [m for m in trunk.modules() if isinstance(m, FusedBatchNorm2d) and m.num_batches_tracked is not None]
    if not bns:
        return None
    dev = bns[0].num_batches_tracked.device
    flat = torch.zeros(len(bns), dtype=torch.long, device=dev)
    for i, m in enumerate(bns):
        flat[i] = m.num_batches_tracked.to(dev)
        m._buffers["num_batches_tracked"] = flat[i]
        m.counter_managed = True
    return flat


def _make_layer(inplanes, planes, blocks, stride):
    downsample = None
    if stride != 1 or inplanes != planes * Bottleneck.expansion:
        downsample = nn.Sequential(
            nn.Conv2d(inplanes, planes * Bottleneck.expansion, kernel_size=1, stride=stride, bias=False),
            FusedBatchNorm2d(planes * Bottleneck.expansion))
    layers = [Bottleneck(inplanes, planes, stride, downsample)]
    inplanes = planes * Bottleneck.expansion
    for _ in range(1, blocks):
        layers.append(Bottleneck(inplanes, planes))
    return nn.Sequential(*layers), inplanes


def resnet152_trunk(depths=(3, 8, 36, 3), keep_avgpool=False):
    """conv1 .. layer4 (optionally + global average pool, for the tagger) as one nn.Sequential."""
    mods = [nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False), FusedBatchNorm2d(64),
            nn.ReLU(inplace=True), nn.MaxPool2d(kernel_size=3, stride=2, padding=1)]
    inplanes = 64
    for planes, blocks, stride in zip((64, 128, 256, 512), depths, (1, 2, 2, 2)):
        layer, inplanes = _make_layer(inplanes, planes, blocks, stride)
        mods.append(layer)
    if keep_avgpool:
        mods.append(nn.AdaptiveAvgPool2d((1, 1)))
    trunk = nn.Sequential(*mods)
    for m in trunk.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)
    return trunk
