"""A few launches of the bf16 trunk kernels (csrc/cgemm16.hip, csrc/wgrad16.hip) on layer3's shapes (B = 32, 16x16 maps,
1024 / 256 channels) for rocprofv3 --pmc passes: VERDICT r02 item 4 asks for SQ_VALU_MFMA_BUSY_CYCLES and FETCH / WRITE
bytes of these kernels.  Algorithmic bytes / flops of each launch are printed so that the counters can be put beside them.
usage: cgemm16_pmc.py   (under rocprofv3 --pmc ...; tools/pmc_agg.py aggregates)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "indonesian-image-captioning_amd")); sys.path.insert(0, ROOT)
import torch
from scnattn._lib import call, ptr, stream_of, ConvExtra, lib

dev = torch.device("cuda:0")
BF = torch.bfloat16
WS = torch.empty(16 << 20, device=dev)
N, H, Cin, p = 32, 16, 1024, 256
R = N * H * H
x = torch.randn(R, Cin, device=dev).to(BF); a1 = torch.randn(R, p, device=dev).to(BF); a2 = torch.randn(R, p, device=dev).to(BF)
w1 = torch.randn(p, Cin, device=dev).to(BF); w3 = torch.randn(4 * p, p, device=dev).to(BF); w3t = torch.randn(p, 4 * p, device=dev).to(BF)
w2 = torch.randn(p, 9 * p, device=dev).to(BF)
z1 = torch.empty(R, p, device=dev, dtype=BF); z3 = torch.empty(R, 4 * p, device=dev, dtype=BF); z2 = torch.empty(R, p, device=dev, dtype=BF)
dz3 = torch.randn(R, 4 * p, device=dev).to(BF); dz2 = torch.randn(R, p, device=dev).to(BF); da2 = torch.empty(R, p, device=dev, dtype=BF)
da1 = torch.empty(R, p, device=dev, dtype=BF)
dw3 = torch.empty(4 * p, p, device=dev); dw2 = torch.empty(p, 9 * p, device=dev)
part = torch.empty(2, 4 * p, lib().scnattn_cgemm_stat_ld(R), device=dev)
ex = ConvExtra(epi=1, stat_partial=part.data_ptr())
st = stream_of(x)
jobs = [
    ("conv1 fwd  8192x256x1024  (cgemm16 + statistics)", 2.0 * R * p * Cin, 2 * (R * Cin + p * Cin + R * p),
     lambda: call("scnattn_cgemm16", st, R, p, Cin, ptr(x), Cin, ptr(w1), Cin, 0.0, ptr(z1), p, 1, ptr(WS), WS.numel(), C.byref(ex))),
    ("conv3 fwd  8192x1024x256  (cgemm16 + statistics)", 2.0 * R * 4 * p * p, 2 * (R * p + 4 * p * p + R * 4 * p),
     lambda: call("scnattn_cgemm16", st, R, 4 * p, p, ptr(a2), p, ptr(w3), p, 0.0, ptr(z3), 4 * p, 1, ptr(WS), WS.numel(), C.byref(ex))),
    ("conv3 dgrad 8192x256x1024 (cgemm16)", 2.0 * R * p * 4 * p, 2 * (R * 4 * p + 4 * p * p + R * p),
     lambda: call("scnattn_cgemm16", st, R, p, 4 * p, ptr(dz3), 4 * p, ptr(w3t), 4 * p, 0.0, ptr(da2), p, 1, ptr(WS), WS.numel(), None)),
    ("conv2 fwd  3x3 256->256   (cgemm16 implicit GEMM + statistics)", 2.0 * R * p * 9 * p, 2 * (R * p + 9 * p * p + R * p),
     lambda: call("scnattn_conv3x3_fwd16", st, N, H, H, p, p, 1, ptr(a1), ptr(w2), ptr(z2), C.byref(ex), ptr(WS), WS.numel())),
    ("conv2 dgrad 3x3           (cgemm16 implicit GEMM)", 2.0 * R * p * 9 * p, 2 * (R * p + 9 * p * p + R * p),
     lambda: call("scnattn_conv3x3_dgrad16", st, N, H, H, p, p, 1, ptr(dz2), ptr(w2), ptr(da1), None, ptr(WS), WS.numel())),
    ("conv3 wgrad 1024x256x8192 (wgrad16_w1)", 2.0 * R * p * 4 * p, 2 * (R * 4 * p + R * p) + 4 * 4 * p * p,
     lambda: call("scnattn_wgrad16_rows", st, R, p, 4 * p, ptr(dz3), ptr(a2), R, ptr(dw3), p, 0, 0, 0, 0, 0, 0, 0, ptr(WS), WS.numel(), 0)),
    ("conv2 wgrad 3x3           (wgrad16_w9)", 2.0 * R * p * 9 * p, 2 * (R * p + R * p) + 4 * 9 * p * p,
     lambda: call("scnattn_wgrad16_3x3", st, N, H, H, p, p, ptr(dz2), ptr(a1), ptr(dw2), ptr(WS), WS.numel(), 0)),
]
for name, fl, by, fn in jobs:
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    print("%-66s %8.3f GFLOP  %8.2f MB algorithmic" % (name, fl / 1e9, by / 1e6))
print("done")
