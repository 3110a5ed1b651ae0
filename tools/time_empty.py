"""Where does torch.empty spend its time inside a bf16 train step?  Wraps torch.empty with a timer, by size class."""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "indonesian-image-captioning_amd"), ROOT]
import torch
from trains.harness import TrainStep, synthetic_batch
dev = torch.device("cuda:0")
ts = TrainStep(kind="attention_scn", fine_tune_encoder=True, device=dev, encoder_dtype="bf16")
cfg = ts.cfg
imgs, tags, caps, caplens = synthetic_batch(32, cfg["vocab_size"], cfg["max_len"], cfg["image_size"], cfg["semantic_dim"], dev, 1)
for _ in range(6):
    ts.step(imgs, tags, caps, caplens)
torch.cuda.synchronize()
real = torch.empty
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
def timed(*a, **k):
    t0 = time.perf_counter()
    r = real(*a, **k)
    dt = time.perf_counter() - t0
    mb = r.numel() * r.element_size() / 1e6
    key = "<1MB" if mb < 1 else "<20MB" if mb < 20 else "<100MB" if mb < 100 else ">=100MB"
    e = acc[key]; e[0] += 1; e[1] += dt; e[2] = max(e[2], dt)
    return r
torch.empty = timed
for _ in range(4):
    ts.step(imgs, tags, caps, caplens)
torch.cuda.synchronize()
torch.empty = real
for k, (n, t, mx) in sorted(acc.items()):
    print("%-8s calls %5d  total %8.2f ms  mean %7.1f us  max %8.1f us" % (k, n, t * 1e3, t / n * 1e6, mx * 1e6))
