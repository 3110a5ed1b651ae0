"""EncoderTagger on MI355X: drop-in for the reference's models/encoders/tagger.py.

(B,3,H,W) images -> ResNet-152 trunk INCLUDING the global average pool -> (B,2048) -> Dropout(0.15) ->
Linear(2048, semantic_size) -> Sigmoid = tag probabilities (B, semantic_size), which the train step feeds
to the decoder as `semantic_input` (trains/attention_scn.py:214).  Same module tree / state_dict keys as
the reference (`resnet.<idx>...`, `linear.{weight,bias}`); the trunk is scnattn.resnet (MIOpen convolutions +
the fused BatchNorm kernels), the Linear runs on the MFMA sgemm."""
import torch
from torch import nn

from scnattn import functional as SF
from scnattn.resnet import resnet152_trunk, configure_miopen, manage_bn_counters
from scnattn.stem import run_trunk, usable as stem_usable

configure_miopen()


class EncoderTagger(nn.Module):
    def __init__(self, semantic_size=1000, dropout=0.15, channels_last=False):
        super().__init__()
        self.semantic_size = semantic_size
        self.resnet = resnet152_trunk(keep_avgpool=True)   # children()[:-1] of torchvision's resnet152
        self.dropout = nn.Dropout(dropout)
        self.linear = nn.Linear(2048, semantic_size)
        self.sigmoid = nn.Sigmoid()
        self.channels_last = channels_last
        self.fine_tune()

    def forward(self, images):
        if self.channels_last and images.is_cuda and not stem_usable(self.resnet, images):      # the fused stem reads any strides
            images = images.contiguous(memory_format=torch.channels_last)
        if images.is_cuda and self.training:
            flat = getattr(self, "_bn_counters", None)
            if flat is None or flat.device != images.device:
                self._bn_counters = flat = manage_bn_counters(self.resnet)
            if flat is not None:
                flat.add_(1)
        out = run_trunk(self.resnet, images)      # stem on csrc/stem.hip, Bottlenecks on scnattn/conv.py (conv16.py in bf16)
        out = out.float()
        out = out.reshape(out.size(0), -1)
        out = self.dropout(out)
        out = SF.linear(out, self.linear.weight, self.linear.bias)
        return self.sigmoid(out)

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
