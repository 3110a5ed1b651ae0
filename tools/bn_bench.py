"""Achieved HBM bandwidth of the fused BatchNorm kernels on the trunk's shapes (B=32, 256x256 input)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "indonesian-image-captioning_amd"), ROOT]
import torch  # noqa: E402
from scnattn import functional as SF  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


print("%-28s %10s %10s %10s %10s" % ("map [N,C,H,W]", "fwd us", "fwd TB/s", "bwd us", "bwd TB/s"))
for (C, HW, res) in [(64, 128, False), (64, 64, False), (256, 64, True), (128, 32, False), (512, 32, True),
                     (256, 16, False), (1024, 16, True), (512, 8, False), (2048, 8, True)]:
    z = torch.randn(32, C, HW, HW, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    r = torch.randn_like(z).contiguous(memory_format=torch.channels_last).requires_grad_(True) if res else None
    g, b = torch.ones(C, device=dev, requires_grad=True), torch.zeros(C, device=dev, requires_grad=True)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    y = SF.bn_act(z, r, g, b, rm, rv, True, 0.1, 1e-5, True)
    dy = torch.randn_like(y)
    nbytes = z.numel() * 4
    tf = timeit(lambda: SF.bn_act(z, r, g, b, rm, rv, True, 0.1, 1e-5, True))
    tb = timeit(lambda: torch.autograd.grad(y, [z] + ([r] if res else []) + [g, b], dy, retain_graph=True))
    # forward: stats read + apply read (+res) + write; backward: reduce (dy, z[, y]) + dx (dy, z[, y]) + writes
    fwd_b = nbytes * (3 + (1 if res else 0))
    bwd_b = nbytes * ((3 if res else 2) + (3 if res else 2) + 1 + (1 if res else 0))
    print("%-28s %10.1f %10.2f %10.1f %10.2f" % ("[32,%d,%d,%d]%s" % (C, HW, HW, "+res" if res else ""), tf,
                                                 fwd_b / tf / 1e6, tb, bwd_b / tb / 1e6))
