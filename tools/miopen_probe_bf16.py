"""Probe: ResNet-152 fwd+bwd under bf16 autocast (MIOpen bf16 convolutions), torch BN path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "indonesian-image-captioning_amd")); sys.path.insert(0, ROOT)
import torch
from scnattn.resnet import resnet152_trunk, FusedBatchNorm2d, configure_miopen
configure_miopen()
torch.backends.cudnn.benchmark = True
FusedBatchNorm2d.use_fused = False
dev = torch.device("cuda:0")
m = resnet152_trunk().to(dev).to(memory_format=torch.channels_last).train()
for i, ch in enumerate(m.children()):
    for p in ch.parameters():
        p.requires_grad = i >= 5
x = torch.randn(32, 3, 256, 256, device=dev).contiguous(memory_format=torch.channels_last)
for mode in ("bf16", "fp32"):
    for i in range(4):
        t = time.time()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=(mode == "bf16")):
            y = m(x)
        torch.cuda.synchronize(); tf = time.time() - t
        y.float().sum().backward()
        torch.cuda.synchronize()
        print("%s step %d fwd %.3fs total %.3fs" % (mode, i, tf, time.time() - t), flush=True)
