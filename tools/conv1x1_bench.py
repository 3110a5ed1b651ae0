"""1x1 convolutions of ResNet-152 (B=32, 256x256 input) as GEMMs on libscnattn's sgemm vs MIOpen through
torch.nn.functional.conv2d (channels-last, cudnn.benchmark + FAST find): forward, d-input, d-weight."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "indonesian-image-captioning_amd")); sys.path.insert(0, ROOT)
os.environ.setdefault("MIOPEN_FIND_MODE", "2"); os.environ.setdefault("MIOPEN_LOG_LEVEL", "3")
import torch
import torch.nn.functional as F
from scnattn import functional as SF
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")


def t_us(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / it


shapes = [("l1.conv1", 64, 64, 256, 64), ("l1.conv3", 64, 64, 64, 256), ("l2.conv1", 32, 32, 512, 128),
          ("l2.conv3", 32, 32, 128, 512), ("l3.conv1", 16, 16, 1024, 256), ("l3.conv3", 16, 16, 256, 1024),
          ("l4.conv1", 8, 8, 2048, 512), ("l4.conv3", 8, 8, 512, 2048)]
B = 32
print("%-10s %6s %6s %6s | %22s | %22s | %22s" % ("layer", "R", "Cin", "Cout", "fwd us mine/miopen", "dgrad us mine/miopen", "wgrad us mine/miopen"), flush=True)
for name, H, W, Cin, Cout in shapes:
    R = B * H * W
    x = torch.randn(B, Cin, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, Cin, 1, 1, device=dev).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, Cout, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    x2 = x.permute(0, 2, 3, 1).reshape(R, Cin); w2 = w.view(Cout, Cin); dy2 = dy.permute(0, 2, 3, 1).reshape(R, Cout)
    y2 = torch.empty(R, Cout, device=dev); dx2 = torch.empty(R, Cin, device=dev); dw2 = torch.empty(Cout, Cin, device=dev)
    f1 = t_us(lambda: SF.gemm(x2, w2, tb=True, out=y2))
    f2 = t_us(lambda: F.conv2d(x, w))
    g1 = t_us(lambda: SF.gemm(dy2, w2, out=dx2))
    g2 = t_us(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False]))
    h1 = t_us(lambda: SF.gemm(dy2, x2, ta=True, out=dw2))
    h2 = t_us(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False]))
    fl = 2.0 * R * Cin * Cout
    print("%-10s %6d %6d %6d | %8.1f /%8.1f (%4.0f TF) | %8.1f /%8.1f | %8.1f /%8.1f" % (name, R, Cin, Cout, f1, f2, fl / f1 / 1e6, g1, g2, h1, h2), flush=True)
    ref = F.conv2d(x, w).permute(0, 2, 3, 1).reshape(R, Cout)
    err = (y2 - ref).abs().max().item() / ref.abs().max().item()
    assert err < 1e-4, err
