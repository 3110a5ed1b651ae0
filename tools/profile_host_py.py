"""Python-level host cost of one train step (cProfile over the main thread AND the autograd thread): which functions of
scnattn/ keep the launching threads busy.  usage: profile_host_py.py [bf16]"""
import cProfile
import os
import pstats
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "indonesian-image-captioning_amd"), ROOT]
import torch  # noqa: E402
from trains.harness import TrainStep, synthetic_batch  # noqa: E402

dev = torch.device("cuda:0")
bf = "bf16" in sys.argv
ts = TrainStep(kind="attention_scn", fine_tune_encoder=True, device=dev, **({"encoder_dtype": "bf16"} if bf else {}))
cfg = ts.cfg
imgs, tags, caps, caplens = synthetic_batch(32, cfg["vocab_size"], cfg["max_len"], cfg["image_size"], cfg["semantic_dim"], dev, 1)
for _ in range(6):
    ts.step(imgs, tags, caps, caplens)
torch.cuda.synchronize()
prof = cProfile.Profile()
threading.setprofile(lambda *a: None)
prof.enable()
for _ in range(4):
    ts.step(imgs, tags, caps, caplens)
torch.cuda.synchronize()
prof.disable()
st = pstats.Stats(prof)
st.sort_stats("tottime").print_stats(22)
st.sort_stats("cumulative").print_stats(22)
