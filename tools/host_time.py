"""How much of a train step is host enqueue time?  For each step: synchronise, call TrainStep.step(),
stamp when Python returns (everything enqueued), synchronise again.  If `host` is close to `total` the
step is launch-bound on the host and GPU-side gains will not show."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "indonesian-image-captioning_amd"), ROOT]
import torch  # noqa: E402
from trains.harness import TrainStep, synthetic_batch  # noqa: E402

dev = torch.device("cuda:0")
fine = "--no-finetune" not in sys.argv
bf = "bf16" in sys.argv
ts = TrainStep(kind="attention_scn", fine_tune_encoder=fine, device=dev, **({"encoder_dtype": "bf16"} if bf else {}))
cfg = ts.cfg
imgs, tags, caps, caplens = synthetic_batch(32, cfg["vocab_size"], cfg["max_len"], cfg["image_size"],
                                            cfg["semantic_dim"], dev, 1234)
for _ in range(6):
    ts.step(imgs, tags, caps, caplens)
torch.cuda.synchronize()
host, total = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ts.step(imgs, tags, caps, caplens)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append(1e3 * (t1 - t0))
    total.append(1e3 * (t2 - t0))
host.sort(); total.sort()
print("fine_tune=%s host enqueue ms: median %.2f (min %.2f)   total ms: median %.2f (min %.2f)"
      % (fine, host[5], host[0], total[5], total[0]))
# back-to-back (no sync between steps): the steady-state rate the bench sees
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    ts.step(imgs, tags, caps, caplens)
torch.cuda.synchronize()
print("back-to-back ms/step: %.2f" % (1e2 * (time.perf_counter() - t0)))
