"""CPU, world_size 2 over gloo: the gradient synchroniser used for the N>1 bench (RCCL on GPUs) averages
gradients exactly, both when armed before backward (bucket all-reduces launched from autograd hooks) and
when flushed entirely inside sync(); un-armed accumulation micro-steps do not communicate."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from image_segmentation_amd.parallel import GradSync
        torch.manual_seed(0)
        model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
        gs = GradSync(model, bucket_mb=0.0003)          # tiny buckets -> several collectives
        assert len(gs.buckets) >= 2
        x = torch.full((5, 8), float(rank + 1))
        res = {}
        for mode in ("armed", "flush"):
            model.zero_grad()
            if mode == "armed":
                gs.arm()
            model(x).sum().backward()
            local = [p.grad.clone() for p in model.parameters()]
            gs.sync()
            gathered = [[torch.zeros_like(g) for _ in range(world)] for g in local]
            for g, out in zip(local, gathered):
                dist.all_gather(out, g)
            for p, out in zip(model.parameters(), gathered):
                assert torch.allclose(p.grad, sum(out) / world, atol=1e-6), mode
            res[mode] = [p.grad.clone() for p in model.parameters()]
        for a, b in zip(res["armed"], res["flush"]):
            assert torch.equal(a, b)
        # accumulation micro-step without arm(): hooks must not communicate nor touch .grad
        model.zero_grad()
        model(x).sum().backward()
        assert not gs._pending
        gs.broadcast_buffers(model)
        q.put((rank, "ok"))
    except Exception as e:          # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gradsync_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert sorted(out) == [(0, "ok"), (1, "ok")], out
