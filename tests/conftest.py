"""pytest configuration: registers the `gpu` marker and puts the repo root + the package overlay
(`indonesian-image-captioning_amd/`, which mirrors the reference's `models/` / `utils/` dotted paths) on sys.path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "indonesian-image-captioning_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
