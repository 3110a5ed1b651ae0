"""pytest configuration: registers the `gpu` marker and puts the repo root + the package overlay
(`indonesian-image-captioning_amd/`, which mirrors the reference's `models/` / `utils/` dotted paths) on sys.path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "indonesian-image-captioning_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


# The product default IS the hand-written 3x3 path (scnattn/conv.py, SCNATTN_CONV3=hip; the per-shape stopwatch of round 2
# is gone).  Pinned here only against a stray value in the environment; the MIOpen side (an A/B switch) is covered by the
# tests that request it explicitly (test_fused_bottleneck_vs_fp64[...-miopen]).
os.environ.setdefault("SCNATTN_CONV3", "hip")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
