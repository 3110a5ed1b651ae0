"""Probe: how long does the first ResNet-152 fwd+bwd take on a fresh box (MIOpen has no gfx950
kernel database in this image, so every conv config is JIT-compiled on first use)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "indonesian-image-captioning_amd")); sys.path.insert(0, ROOT)
import torch
from models.encoders.caption import EncoderCaption
cl = os.environ.get("PROBE_CL", "1") == "1"
B = int(os.environ.get("PROBE_B", "32"))
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = os.environ.get("PROBE_BENCH", "0") == "1"
t0 = time.time()
enc = EncoderCaption(channels_last=cl).to(dev)
enc.fine_tune(True); enc.train()
x = torch.randn(B, 3, 256, 256, device=dev)
print("built %.1fs cl=%s" % (time.time() - t0, cl), flush=True)
for i in range(4):
    t = time.time()
    y = enc(x)
    torch.cuda.synchronize(); tf = time.time() - t
    y.sum().backward()
    torch.cuda.synchronize()
    print("step %d fwd %.2fs total %.2fs" % (i, tf, time.time() - t), flush=True)
