"""Which torch ops issue the device memcpy / fill calls of a decoder-only train step?  (torch.profiler)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "indonesian-image-captioning_amd"), ROOT]
import torch  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402
from trains.harness import TrainStep, synthetic_batch  # noqa: E402

dev = torch.device("cuda:0")
ts = TrainStep(kind="attention_scn", fine_tune_encoder=False, device=dev, encoder=False)
cfg = ts.cfg
_, tags, caps, caplens = synthetic_batch(32, cfg["vocab_size"], cfg["max_len"], cfg["image_size"], cfg["semantic_dim"], dev, 1)
enc = torch.rand(32, 14, 14, 2048, device=dev)
for _ in range(3):
    ts.step(None, tags, caps, caplens, enc)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False, record_shapes=True) as prof:
    ts.step(None, tags, caps, caplens, enc)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cpu_time_total", row_limit=60, max_name_column_width=60))
ev = [e for e in prof.events() if "emcpy" in e.name or "emset" in e.name]
print("memcpy/memset events:", len(ev))
from collections import Counter
print(Counter(e.name for e in ev).most_common(10))
