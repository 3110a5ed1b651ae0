"""Where does the HOST spend a train step, and does the GPU wait for it?

For each mode (single GPU; --force-dist = hooks + buckets + 1-rank all-reduce) runs a few steps of the default bench
configuration and prints, per step, the host wall time of each phase WITHOUT synchronising in between, then the time
the final synchronize had to wait.  A final wait near zero means the step is host-bound (the GPU finished as soon as
the host stopped feeding it); a long wait means the host ran ahead.

    python tools/host_timeline.py [--steps 6] [--modes single,dist,dist-cabi]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "indonesian-image-captioning_amd"))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")


WARM = False
BF16 = False


def run(mode, steps):
    import torch.distributed as dist
    from trains.harness import TrainStep, synthetic_batch
    from scnattn import resnet
    resnet.configure_miopen()
    if mode != "single":
        os.environ["SCNATTN_DP_BACKEND"] = "cabi" if mode == "dist-cabi" else "torch"
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            torch.cuda.set_device(0)
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    dev = torch.device("cuda", 0)
    if WARM and mode != "single":         # what bench.py does before building the step under torch.distributed
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
        import bench
        bench.warm_miopen(dev, 32, True)
        dist.barrier()
    ts = TrainStep(device=dev, force_reduce=(mode != "single"), **({"encoder_dtype": "bf16"} if BF16 else {}))
    cfg = ts.cfg
    imgs, tags, caps, caplens = synthetic_batch(cfg["batch_size"], cfg["vocab_size"], cfg["max_len"], cfg["image_size"],
                                                cfg["semantic_dim"], dev, 1234)
    for _ in range(6):
        ts.step(imgs, tags, caps, caplens)
    torch.cuda.synchronize()
    rows = []
    for _ in range(steps):
        t = [time.perf_counter()]
        prepool = ts.encoder(imgs, pooled=False)
        t.append(time.perf_counter())
        scores, caps_sorted, decode_lengths, alphas, sort_ind = ts.decoder(None, tags, caps, caplens, prepool=prepool,
                                                                             pool_size=ts.encoder.enc_image_size)
        dl_dev = (caplens.reshape(-1)[sort_ind] - 1).to(torch.int32)
        loss = ts.loss_fn(scores, caps_sorted, decode_lengths, alphas, dl_dev)
        t.append(time.perf_counter())
        ts.decoder_optimizer.zero_grad()
        ts.encoder_optimizer.zero_grad()
        for r in ts.reducers:
            r.reset()
        loss.backward()
        t.append(time.perf_counter())
        scale = 1.0
        for r in ts.reducers:
            scale = r.finish()
        t.append(time.perf_counter())
        ts.decoder_optimizer.step(scale)
        ts.encoder_optimizer.step(scale)
        t.append(time.perf_counter())
        torch.cuda.synchronize()
        t.append(time.perf_counter())
        rows.append([(b - a) * 1e3 for a, b in zip(t[:-1], t[1:])] + [(t[-1] - t[0]) * 1e3])
    names = ["enc fwd", "dec fwd+loss", "backward", "dp finish", "optimizers", "final sync wait", "step total"]
    print("mode", mode)
    for i, n in enumerate(names):
        v = sorted(r[i] for r in rows)
        print("  %-16s median %7.2f ms   min %7.2f  max %7.2f" % (n, v[len(v) // 2], v[0], v[-1]))
    sys.stdout.flush()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--modes", default="single,dist")
    ap.add_argument("--warm", action="store_true")
    ap.add_argument("--bf16", action="store_true", help="the mixed-precision trunk (bench.py --encoder-dtype bf16)")
    a = ap.parse_args()
    WARM = a.warm
    BF16 = a.bf16
    for m in a.modes.split(","):
        run(m, a.steps)
