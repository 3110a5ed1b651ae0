"""Data-parallel path on CPU: world_size 2, gloo backend, 127.0.0.1 rendezvous.  Checks that the
flat-buffer + bucketed all-reduce (scnattn/dp.py) gives every rank exactly the gradient of the
single-process run on the concatenated batch, that buckets fire from the autograd hooks (overlap path)
and that parameters stay views of the flat buffer.  No GPU, no libscnattn compute calls."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(5)
    return torch.nn.Sequential(torch.nn.Linear(12, 40), torch.nn.Tanh(), torch.nn.Linear(40, 33), torch.nn.Tanh(),
                               torch.nn.Linear(33, 7))


def _data():
    g = torch.Generator().manual_seed(9)
    return torch.randn(16, 12, generator=g), torch.randn(16, 7, generator=g)


def _worker(rank, world, port, q):
    for p in (os.path.join(ROOT, "indonesian-image-captioning_amd"), ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from scnattn.flat import FlatBuffer
        from scnattn.dp import GradReducer, broadcast_parameters
        m = _model()
        if rank == 1:   # start from different weights: the broadcast must fix it
            with torch.no_grad():
                for p in m.parameters():
                    p.add_(1.0)
        flat = FlatBuffer(m.parameters())
        red = GradReducer(flat, bucket_bytes=2048)      # tiny buckets -> several all-reduces
        broadcast_parameters(flat)
        assert len(red.buckets) >= 3
        x, y = _data()
        n = x.shape[0] // world
        xs, ys = x[rank * n:(rank + 1) * n], y[rank * n:(rank + 1) * n]
        for it in range(2):                              # twice: reset() must re-arm the hooks
            flat.zero_grad()
            red.reset()
            loss = ((m(xs) - ys) ** 2).mean()
            loss.backward()
            fired_in_backward = sum(red.launched)
            scale = red.finish()
        for p, gv in zip(flat.params, flat.gviews):
            assert p.grad.data_ptr() == gv.data_ptr()      # after finish() every .grad is the flat view
            assert p.data_ptr() >= flat.flat_p.data_ptr()
        q.put((rank, flat.flat_g.clone() * scale, flat.flat_p.clone(), fired_in_backward, len(red.buckets)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gradients_equal_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda r: r[0])
    # single-process reference on the concatenated batch
    for p in (os.path.join(ROOT, "indonesian-image-captioning_amd"), ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    from scnattn.flat import FlatBuffer
    m = _model()
    flat = FlatBuffer(m.parameters())
    x, y = _data()
    flat.zero_grad()
    ((m(x) - y) ** 2).mean().backward()
    flat.gather()
    for rank, g, p, fired, nb in res:
        assert torch.allclose(g, flat.flat_g, atol=1e-6, rtol=1e-5), "rank %d gradient differs" % rank
        assert torch.equal(p, flat.flat_p), "rank %d parameters differ after broadcast" % rank
        assert fired == nb, "every bucket should have been launched from a hook during backward"
    assert torch.equal(res[0][1], res[1][1])


def test_single_process_reducer_is_a_noop():
    for p in (os.path.join(ROOT, "indonesian-image-captioning_amd"), ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    from scnattn.flat import FlatBuffer
    from scnattn.dp import GradReducer
    m = _model()
    flat = FlatBuffer(m.parameters())
    red = GradReducer(flat)
    x, y = _data()
    flat.zero_grad(); red.reset()
    ((m(x) - y) ** 2).mean().backward()
    assert red.finish() == 1.0
    flat.gather()
    assert flat.flat_g.abs().sum().item() > 0
    keep = flat.flat_g.clone()
    # a second backward without zero_grad accumulates into the flat views (autograd's in-place add)
    ((m(x) - y) ** 2).mean().backward()
    flat.gather()
    assert torch.allclose(flat.flat_g, 2 * keep, atol=1e-6)
    # parameters that got no gradient are gathered as zeros
    flat.zero_grad()
    flat.gather()
    assert flat.flat_g.abs().sum().item() == 0.0
