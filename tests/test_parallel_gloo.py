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
        # reference values: an un-armed backward must not communicate nor touch .grad (accumulation micro-step)
        model.zero_grad()
        model(x).sum().backward()
        assert gs.collectives == 0 and not gs._pending
        local = [p.grad.clone() for p in model.parameters()]
        gathered = [[torch.zeros_like(g) for _ in range(world)] for g in local]
        for g, out in zip(local, gathered):
            dist.all_gather(out, g)
        want = [sum(out) / world for out in gathered]
        res = {}
        for mode in ("armed", "flush", "accumulated"):
            model.zero_grad()
            if mode == "accumulated":          # two micro-batches, only the second one armed: (g + g) averaged
                model(x).sum().backward()
            if mode != "flush":
                gs.arm()                       # bucket all-reduces are launched from the autograd hooks
            model(x).sum().backward()
            gs.sync()
            k = 2.0 if mode == "accumulated" else 1.0
            for p, w in zip(model.parameters(), want):
                assert torch.allclose(p.grad, k * w, atol=1e-6), mode
                assert gs._in_place(p), mode   # .grad lives in its bucket slice after sync
            res[mode] = [p.grad.clone() for p in model.parameters()]
        for a, b in zip(res["armed"], res["flush"]):
            assert torch.equal(a, b)
        assert gs.collectives == 3 * len(gs.buckets)
        # a parameter that receives no gradient on ONE rank only (data-dependent branch): hook-launched buckets go out
        # strictly in index order, so both ranks issue the same sequence of collectives (the middle layer's bucket and the
        # ones after it wait for sync() on rank 1) -- no hang, no mixed buckets; the idle rank contributes zeros
        gs.remove()
        torch.manual_seed(1)
        m2 = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.Linear(8, 8), torch.nn.Linear(8, 4))
        gs2 = GradSync(m2, bucket_mb=0.0003)
        assert len(gs2.buckets) >= 3

        def fwd2(r):
            h = m2[0](torch.full((5, 8), float(r + 1)))          # rank r's batch
            if r == 0:
                h = m2[1](h)
            return m2[2](h).sum()
        singles = []
        for r in range(world):
            m2.zero_grad()
            fwd2(r).backward()
            singles.append([None if p.grad is None else p.grad.clone() for p in m2.parameters()])
        m2.zero_grad()
        gs2.arm()
        fwd2(rank).backward()
        gs2.sync()
        for i, p in enumerate(m2.parameters()):
            w2 = sum(torch.zeros_like(p) if s_[i] is None else s_[i] for s_ in singles) / world
            assert torch.allclose(p.grad, w2, atol=1e-6), ("uneven", i)
        gs2.remove()
        gs = GradSync(model, bucket_mb=0.0003)
        # replicas start identical: rank 1 perturbs its parameters, a new synchroniser restores rank 0's
        if rank == 1:
            with torch.no_grad():
                for p in model.parameters():
                    p.add_(1.0)
        gs.remove()
        gs = GradSync(model, bucket_mb=0.0003)
        ref = [p.detach().clone() for p in model.parameters()]
        for t in ref:
            dist.broadcast(t, src=0)
        for p, t in zip(model.parameters(), ref):
            assert torch.equal(p.detach(), t)
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


from image_segmentation_amd.metrics import MetricsHistory as _MH      # noqa: E402  (module level: checkpoints pickle it)


class _OracleAgg(_MH):
    """CPU stand-in for the device confusion kernel (host formulas unchanged)."""
    def accumulate(self, pred, label):
        from oracle import losses_ref
        c = losses_ref.confusion_counts(pred, label, self.num_classes)
        self.total_tp += c[0]; self.total_fp += c[1]; self.total_fn += c[2]; self.total_tn += c[3]


class _OracleCE(torch.nn.Module):
    def forward(self, pred, y):
        from oracle import losses_ref
        return losses_ref.cross_entropy(pred, y)


def _start_worker(rank, world, port, tmp, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import losses_ref
        from oracle.fill import fill, labels
        from image_segmentation_amd import training
        from image_segmentation_amd.metrics import MetricsHistory
        from image_segmentation_amd.parallel import GradSync
        training.VERBOSE = False

        Agg, CE = _OracleAgg, _OracleCE
        torch.manual_seed(100 + rank)       # replicas start DIFFERENT: GradSync must make them rank 0's
        model = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.BatchNorm2d(8), torch.nn.ReLU(),
                                    torch.nn.Conv2d(8, 3, 1))
        gs = GradSync(model)
        opt = torch.optim.AdamW(model.parameters(), lr=1e-2)
        train = [(fill((2, 3, 16, 16), 10 + 7 * rank + i, 0, 1), labels((2, 1, 16, 16), 20 + 7 * rank + i, 3)) for i in range(3)]
        val_all = [([fill((3, 12 + i, 16), 50 + i, 0, 1)], [labels((12 + i, 16), 60 + i, 3)]) for i in range(4)]
        val = val_all[rank::world]          # sharded validation set
        d = os.path.join(tmp, f"rank{rank}")
        best = training.start(d, "m.pt", model, opt, train, val, 2, "cpu", CE(), CE(), 16, agg=Agg(3), load=False,
                              save=True, num_classes=3, ignore_index=None, epochs=2, grad_sync=gs)
        files = sorted(os.listdir(d)) + sorted(os.listdir(os.path.join(d, "metrics")))
        state = torch.cat([p.detach().reshape(-1) for p in model.parameters()] +
                          [b.reshape(-1).double().float() for b in model.buffers()])
        q.put((rank, best, files, state.tolist()))
    except Exception as e:          # pragma: no cover
        import traceback
        q.put((rank, repr(e) + traceback.format_exc(), None, None))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_start_is_rank_aware_world2_gloo(tmp_path):
    """training.start under data parallelism: replicas made identical at construction, BatchNorm buffers broadcast
    before evaluation, evaluation counts summed over the ranks (sharded validation set) so both ranks see the same
    metrics, and only rank 0 writes the checkpoint / weights-only / metrics files."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_start_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted((q.get(timeout=150) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=30)
    (r0, best0, files0, state0), (r1, best1, files1, state1) = out
    assert files0 is not None and files1 is not None, (best0, best1)
    assert best0 == best1                                  # same decision, same metrics on both ranks
    assert "m.pt" in files0 and "MO_m.pt" in files0 and files0.count("m.pt") == 2    # checkpoint + metrics/m.pt
    assert files1 == ["metrics"]                           # rank 1 wrote nothing
    assert state0 == state1                                # parameters AND BatchNorm buffers identical after training
