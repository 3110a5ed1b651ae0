"""Worker of tests/test_gpu_parity.py::test_two_process_data_parallel_decoder_step (not a test module).
One process per rank, all on cuda:0 (RCCL refuses two ranks on one device, so the collective backend is
gloo; the HIP path, the autograd hooks, the buckets and the 1/world scale are the product's).
usage: dp_gpu_worker.py RANK WORLD PORT OUTDIR"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "indonesian-image-captioning_amd"), ROOT]
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from trains.harness import TrainStep, synthetic_batch
    dev = torch.device("cuda:0")
    ts = TrainStep(kind="attention_scn", fine_tune_encoder=False, device=dev, encoder=False, bucket_mb=0, seed=77 + rank,
                   emb_dim=32, attention_dim=24, decoder_dim=32, factored_dim=40, semantic_dim=12, vocab_size=60,
                   dropout=0.0, max_len=6)                 # different seeds: the broadcast must align the ranks
    G = 8
    _, tags, caps, caplens = synthetic_batch(G, 60, 6, 8, 12, dev, 5)
    enc = torch.rand(G, 14, 14, 2048, generator=torch.Generator().manual_seed(3)).to(dev)
    n = G // world
    sl = slice(rank * n, (rank + 1) * n)
    enc, tags, caps, caplens = enc[sl], tags[sl], caps[sl], caplens[sl]
    red = ts.reducers[0]
    flat = ts.decoder_optimizer.flat
    p0 = flat.flat_p.clone()
    scores, caps_sorted, dl, alphas, _ = ts.decoder(enc, tags, caps, caplens)
    loss = ts.loss_fn(scores, caps_sorted, dl, alphas)
    ts.decoder_optimizer.zero_grad()
    red.reset()
    loss.backward()
    fired = sum(red.launched) if red.enabled else 0
    scale = red.finish()
    flat.gather()
    grad = (flat.flat_g * scale).cpu()
    loss2 = ts.step(None, tags, caps, caplens, enc)               # the whole step once, through the optimizer
    torch.cuda.synchronize()
    names = [k for k, p in ts.decoder.named_parameters() if p.requires_grad]
    torch.save({"grad": grad, "p0": p0.cpu(), "loss": float(loss), "loss2": float(loss2), "fired": fired,
                "buckets": len(red.buckets), "offsets": flat.offsets, "names": names,
                "numels": [p.numel() for p in flat.params], "p1": flat.flat_p.cpu()},
               os.path.join(out, "w%d_r%d.pt" % (world, rank)))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
