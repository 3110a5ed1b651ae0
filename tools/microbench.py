"""Per-kernel timings at BASELINE shapes (B=32, P=196, E=2048, A=D=F=M=512), back-to-back launches
bracketed by events on the launch stream.  Usage: python tools/microbench.py [filter]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "indonesian-image-captioning_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from scnattn._lib import call, ptr, stream_of  # noqa: E402
from scnattn import functional as SF  # noqa: E402

dev = torch.device("cuda:0")
if os.environ.get("ATTN_DEPTH"):
    SF.set_option("attn_depth", int(os.environ["ATTN_DEPTH"]))
flt = sys.argv[1] if len(sys.argv) > 1 else ""
B, P, E, A, D, F = 32, 196, 2048, 512, 512, 512
F4 = 4 * F


def R(*s):
    return torch.randn(*s, device=dev)


def timeit(name, fn, bytes_=0, flops=0, iters=50):
    if flt and flt not in name:
        return
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / iters
    extra = ""
    if bytes_:
        extra += "  %.2f TB/s" % (bytes_ / us / 1e6)
    if flops:
        extra += "  %.1f TFLOP/s" % (flops / us / 1e6)
    print("%-46s %9.2f us%s" % (name, us, extra), flush=True)


def skinny(name, rows, N, K, groups=1, ks=0):
    X = R(rows, groups * K)
    W = R(groups, K, N)
    Y = torch.empty(16, groups, rows, N, device=dev)
    used = C.c_int(0)
    st = stream_of(X)

    def fn():
        call("scnattn_skinny_gemm", st, rows, N, K, groups, ptr(X), groups * K, K, ptr(W), N, K * N, ptr(Y), N,
             rows * N, groups * rows * N, ks, C.byref(used))
    fn()
    timeit("%s ks=%d" % (name, used.value), fn, bytes_=4 * groups * K * N, flops=2 * rows * groups * K * N)


def gemm(name, M, N, K, ta=False, tb=False):
    a = R(K, M) if ta else R(M, K)
    b = R(N, K) if tb else R(K, N)
    out = torch.empty(M, N, device=dev)
    timeit("sgemm %s %dx%dx%d %s%s" % (name, M, N, K, "T" if ta else "N", "T" if tb else "N"),
           lambda: SF.gemm(a, b, ta=ta, tb=tb, out=out), flops=2 * M * N * K, iters=10)


for ks in (0, 2, 4, 8):
    skinny("skinny A h->[att2|gpre|ph] 32x4608x512", 32, 4608, 512, 1, ks)
for ks in (0, 4, 8, 16):
    skinny("skinny C z.Wa 32x2048x2048", 32, 2048, 2048, 1, ks)
for ks in (0, 2, 4, 8):
    skinny("skinny D 4x(32x512x1024)", 32, 512, 1024, 4, ks)
skinny("skinny Db 4x(32x1024x512)", 32, 1024, 512, 4)
skinny("skinny H 32x512x4608", 32, 512, 4608, 1)

enc = torch.rand(B, P, E, device=dev)
att1, att2, w, b0, bd = R(B, P, A), R(4, B, A), R(A), R(1), R(A)
e = torch.empty(B, P, device=dev)
st = stream_of(enc)
timeit("attn_scores", lambda: call("scnattn_attn_scores", st, B, P, A, ptr(att1), ptr(att2), 2, B * A, A, ptr(bd),
                                   ptr(w), ptr(b0), ptr(e), None), bytes_=4 * B * P * A)
gpre, bb = R(4, B, E), R(E)
alpha, awe = torch.empty(B, P, device=dev), torch.empty(B, E, device=dev)
gate, z = torch.empty(B, E, device=dev), torch.empty(B, E, device=dev)
timeit("attn_context", lambda: call("scnattn_attn_context", st, B, P, E, ptr(enc), ptr(e), ptr(gpre), 2, B * E, E,
                                    ptr(bb), ptr(alpha), P, None, ptr(awe), ptr(gate), ptr(z)), bytes_=4 * B * P * E)
dawe, dal = R(B, E), torch.empty(B, P, device=dev)
timeit("attn_dalpha", lambda: call("scnattn_attn_dalpha", st, B, P, E, ptr(enc), ptr(dawe), None, P, ptr(dal)),
       bytes_=4 * B * P * E)
de, datt2, a2 = torch.empty(B, P, device=dev), torch.empty(B, A, device=dev), R(B, A)
timeit("attn_softmax_bwd", lambda: call("scnattn_attn_softmax_bwd", st, B, P, A, ptr(att1), ptr(a2), ptr(w),
                                        ptr(alpha), ptr(dal), ptr(de), ptr(datt2), A), bytes_=4 * B * P * A)
r, c0, bi = R(8, 4, B, D), R(B, D), R(4 * D)
gates = torch.empty(B, 4 * D, device=dev)
cn, hn, tc = torch.empty(B, D, device=dev), torch.empty(B, D, device=dev), torch.empty(B, D, device=dev)
for ns in (1, 4, 8):
    timeit("lstm_fwd nslab=%d" % ns,
           lambda: call("scnattn_lstm_fwd", st, B, D, ptr(r), ns, 4 * B * D, D, B * D, ptr(bi), ptr(bi), ptr(c0),
                        ptr(gates), ptr(cn), ptr(hn), ptr(tc)))
pz, ex, ph, qx = R(8, B, F4), R(B, F4), R(4, B, 4608), R(B, F4)
pa, phs = torch.empty(B, F4, device=dev), torch.empty(B, F4, device=dev)
xcat = torch.empty(B, 4, 2 * F, device=dev)
timeit("scn_mix_fwd", lambda: call("scnattn_scn_mix_fwd", st, B, F4, ptr(pz), 8, B * F4, F4, ptr(ex), ptr(ph), 2,
                                   B * 4608, 4608, ptr(qx), ptr(qx), ptr(pa), ptr(phs), ptr(xcat)))
gemm("att1", 6272, 512, 2048, tb=True)
gemm("fc", 1632, 10000, 512, tb=True)
gemm("dHd", 1632, 512, 10000)
gemm("dWfc", 10000, 512, 1632, ta=True)
gemm("dWe", 512, 2048, 6272, ta=True)
gemm("denc", 6272, 2048, 512)
gemm("dWa", 2048, 2048, 1632, ta=True)
gemm("ex", 1632, 2048, 512)
gemm("dWaM", 512, 2048, 1632, ta=True)
gemm("dHa", 512, 2048, 1632, ta=True)
gemm("dWbeta", 2048, 512, 1632, ta=True)
gemm("dWd", 512, 512, 1632, ta=True)
gemm("demb", 1632, 512, 2048, tb=True)
gemm("dWc_g", 512, 512, 1632, ta=True)
gemm("init_h", 32, 512, 2048, tb=True)
gemm("qx", 32, 2048, 1000)
gemm("sq4096", 4096, 4096, 4096)
gemm("sq4096", 4096, 4096, 4096, tb=True)
gemm("sq4096", 4096, 4096, 4096, ta=True)
x = R(6272, 512)
timeit("colsum 6272x512", lambda: SF.colsum(x), bytes_=4 * 6272 * 512, iters=10)
x2 = R(1632, 2048)
timeit("colsum 1632x2048", lambda: SF.colsum(x2), bytes_=4 * 1632 * 2048, iters=10)

# library baseline for the same products (torch.mm -> hipBLASLt / rocBLAS): what a "plain library GEMM" would give
if not flt or "lib" in flt:
    def lib(name, M, N, K, ta=False, tb=False):
        a = R(K, M) if ta else R(M, K)
        b = R(N, K) if tb else R(K, N)
        out = torch.empty(M, N, device=dev)
        timeit("lib   %s %dx%dx%d %s%s" % (name, M, N, K, "T" if ta else "N", "T" if tb else "N"),
               lambda: torch.mm(a.t() if ta else a, b.t() if tb else b, out=out), flops=2 * M * N * K, iters=10)
    lib("att1", 6272, 512, 2048, tb=True)
    lib("y", 2048, 512, 2048, tb=True)
    lib("fc", 1632, 10000, 512, tb=True)
    lib("dHd", 1632, 512, 10000)
    lib("dWfc", 10000, 512, 1632, ta=True)
    lib("dWe", 512, 2048, 6272, ta=True)
    lib("dWa", 2048, 2048, 1632, ta=True)
    lib("ex", 1632, 2048, 512)
    lib("dWaM", 512, 2048, 1632, ta=True)
    lib("demb", 1632, 512, 2048, tb=True)
    lib("sq4096", 4096, 4096, 4096)
