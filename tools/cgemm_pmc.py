"""A few launches of csrc/cgemm.hip on chosen shapes, for rocprofv3 --pmc passes (see tools/README.md)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "indonesian-image-captioning_amd")); sys.path.insert(0, ROOT)
import torch
from scnattn._lib import call, ptr, stream_of

dev = torch.device("cuda:0")
WS = torch.empty(16 << 20, device=dev)
shapes = [(8192, 1024, 256, 0, 1), (131072, 256, 64, 0, 1), (32768, 128, 512, 0, 1), (4096, 4096, 4096, 0, 1),
          (8192, 256, 1024, 0, 0), (1024, 256, 8192, 1, 0)]
for (M, N, K, ta, tb) in shapes:
    a = torch.randn((K, M) if ta else (M, K), device=dev); b = torch.randn((N, K) if tb else (K, N), device=dev)
    o = torch.empty(M, N, device=dev)
    for _ in range(4):
        call("scnattn_cgemm", stream_of(a), ta, tb, M, N, K, 1.0, ptr(a), a.stride(0), ptr(b), b.stride(0), 0.0, ptr(o),
             o.stride(0), None, None, 1, 0, 0, 0, ptr(WS), WS.numel(), None)
    torch.cuda.synchronize()
print("done")
