"""EncoderCaption on MI355X: drop-in for the reference's models/encoders/caption.py.

(B,3,H,W) images -> ResNet-152 trunk -> (B,2048,H/32,W/32) -> AdaptiveAvgPool2d(14) + permute ->
(B,14,14,2048).  The trunk is scnattn.resnet.resnet152_trunk (same state_dict keys as the reference's
``resnet.<idx>...``); the pooling + NHWC permute tail (reference :41-43) is one fused HIP kernel that
reads the trunk output in whatever memory format it has and writes a contiguous NHWC tensor, so the
decoder's flatten is free.  ``fine_tune`` freezes the stem and layer1 and toggles layer2-4 (:46-57)."""
import torch
from torch import nn

from scnattn import functional as SF
from scnattn.resnet import resnet152_trunk, configure_miopen, manage_bn_counters
from scnattn.stem import run_trunk, usable as stem_usable

configure_miopen()


class EncoderCaption(nn.Module):
    def __init__(self, encoded_image_size=14, state_dict_path=None, channels_last=False):
        super().__init__()
        self.enc_image_size = encoded_image_size
        self.resnet = resnet152_trunk()
        # kept so the module tree matches the reference; forward uses the fused HIP kernel instead
        self.adaptive_pool = nn.AdaptiveAvgPool2d((encoded_image_size, encoded_image_size))
        self.channels_last = channels_last
        if state_dict_path is not None:  # e.g. torchvision's resnet152 weights saved as a state_dict
            sd = torch.load(state_dict_path, map_location="cpu", weights_only=True)
            self.resnet.load_state_dict({k: v for k, v in sd.items() if not k.startswith("fc.")}, strict=False)
        self.fine_tune()

    def forward(self, images, pooled=True):
        """(B, 3, H, W) -> (B, enc_image_size, enc_image_size, 2048) as the reference (:34-44).  The returned
        tensor carries the un-pooled trunk map as `_scn_prepool` so that this build's attention decoders can
        work on its 8x8 source pixels (models/decoders/_common.py::attached_prepool); `pooled=False` returns
        that (B, h, w, 2048) map itself and skips the pooling."""
        if self.channels_last and images.is_cuda and not stem_usable(self.resnet, images):      # the fused stem reads any strides
            images = images.contiguous(memory_format=torch.channels_last)
        if images.is_cuda and self.training:
            flat = getattr(self, "_bn_counters", None)
            if flat is None or flat.device != images.device:
                self._bn_counters = flat = manage_bn_counters(self.resnet)
            if flat is not None:
                flat.add_(1)     # all BatchNorm num_batches_tracked counters, one launch
        out = run_trunk(self.resnet, images)      # stem on csrc/stem.hip, Bottlenecks on scnattn/conv.py (conv16.py in bf16)
        if out.dtype == torch.bfloat16:           # mixed-precision trunk: the decoder takes (and returns gradients for) fp32
            out = out.float()
        pre = out.permute(0, 2, 3, 1)             # a contiguous (B, h, w, C) view when the trunk is channels-last
        if not pooled:
            return pre
        y = SF.pool_permute(out, self.enc_image_size)
        if y.is_cuda:
            y._scn_prepool = (pre, y._version)
        return y

    def fine_tune(self, fine_tune=True):
        for p in self.resnet.parameters():
            p.requires_grad = False
        for child in list(self.resnet.children())[5:]:
            for p in child.parameters():
                p.requires_grad = fine_tune

    def to(self, *args, **kwargs):
        m = super().to(*args, **kwargs)
        if self.channels_last and any(p.is_cuda for p in m.parameters()):
            m.resnet.to(memory_format=torch.channels_last)
        return m


class Encoder(EncoderCaption):
    """Alias kept because old checkpoints pickle the class under this name (reference :60-62)."""
    pass
