"""Host-side cost of one full train step by op (torch.profiler, CPU self/total time): which eager ops keep the
launching threads busy while the GPU waits."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "indonesian-image-captioning_amd"), ROOT]
import torch  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402
from trains.harness import TrainStep, synthetic_batch  # noqa: E402

dev = torch.device("cuda:0")
ts = TrainStep(kind="attention_scn", fine_tune_encoder=True, device=dev, **({"encoder_dtype": "bf16"} if "bf16" in sys.argv else {}))
cfg = ts.cfg
imgs, tags, caps, caplens = synthetic_batch(32, cfg["vocab_size"], cfg["max_len"], cfg["image_size"], cfg["semantic_dim"], dev, 1)
for _ in range(6):
    ts.step(imgs, tags, caps, caplens)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(2):
        ts.step(imgs, tags, caps, caplens)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=28, max_name_column_width=58))
print(prof.key_averages().table(sort_by="cpu_time_total", row_limit=16, max_name_column_width=58))
