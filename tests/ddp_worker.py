"""Worker of the data-parallel equivalence check (tests/test_gpu_ddp.py, tools/ddp_check.py): N ranks, one process
each, every rank on the SAME GPU (cuda:0) with the gloo backend -- the one-GPU box has no second card for RCCL, but
everything above the transport is the code the N-GPU bench runs (parallel.GradSync: bucket slices as gradient
destinations, hook-launched in-place all-reduces on a side stream, sync before the optimizer step).

What is checked (semantics of reference utils/training.py:46-56 extended over ranks): with identical replicas,
each rank's synchronised `.grad` equals the MEAN over ranks of the gradients single-process steps produce on each
rank's data -- for accumulation 1 and 2 (only the stepping micro-batch is armed; un-armed micro-steps issue no
collective) -- on the real unet(3,3) through the HIP kernels."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _u01(n, seed):
    import numpy as np
    with np.errstate(over="ignore"):
        z = np.arange(n, dtype=np.uint64) + np.uint64((seed * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
        return (z >> np.uint64(40)).astype(np.float64) / float(1 << 24)


def _data(rank, k, B, S, dev):
    import numpy as np
    import torch
    x = torch.from_numpy(_u01(B * 3 * S * S, 100 + 10 * rank + k).astype(np.float32).reshape(B, 3, S, S)).to(dev)
    y = torch.from_numpy(np.clip(np.floor(3 * _u01(B * S * S, 200 + 10 * rank + k)), 0, 2).astype(np.int64)
                         .reshape(B, S, S)).to(dev)
    return x, y


def run(rank, world, port, dtype_name, out_path, B=2, S=64):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))   # a peer that
    # died before the rendezvous must not leave this rank waiting forever
    res = {"rank": rank, "dtype": dtype_name, "ok": False}
    try:
        import image_segmentation_amd as seg
        from image_segmentation_amd.parallel import GradSync
        dtype = {"f32": torch.float32, "bf16": torch.bfloat16}[dtype_name]
        seg.set_compute_dtype(dtype)
        torch.manual_seed(7 + rank)                      # replicas start DIFFERENT; GradSync makes them rank 0's
        model = seg.unet(3, 3).to(dev).train()
        loss_fn = seg.CrossEntropyLoss()
        gs = GradSync(model, bucket_mb=8.0)
        params = list(model.parameters())
        names = [n for n, _ in model.named_parameters()]

        def local_grads(r, ks, scale):
            """single-process gradients of sum_k loss(X_{r,k}) * scale (what rank r would hold without any sync)"""
            model.zero_grad(set_to_none=True)
            for k in ks:
                x, y = _data(r, k, B, S, dev)
                (loss_fn(model(x), y) * scale).backward()
            return [p.grad.detach().clone() for p in params]

        worst = {}
        for acc in (1, 2):
            ks = list(range(acc))
            before, direct0 = gs.collectives, gs.direct_grads
            per_rank = [local_grads(r, ks, 1.0 / acc) for r in range(world)]     # no arm(): must not communicate
            assert gs.collectives == before, "un-armed micro-steps issued a collective"
            want = [sum(g[i] for g in per_rank) / world for i in range(len(params))]
            # the data-parallel step: accumulation micro-batches un-armed, the last one armed, sync, compare
            model.zero_grad(set_to_none=True)
            for k in ks:
                x, y = _data(rank, k, B, S, dev)
                if k == ks[-1]:
                    gs.arm()
                (loss_fn(model(x), y) / acc).backward()
            gs.sync()
            torch.cuda.synchronize()
            assert gs.collectives == before + len(gs.buckets)
            rel = {}
            for n, p, w in zip(names, params, want):
                den = float(w.norm())
                rel[n] = float((p.grad - w).norm()) / den if den > 0 else float((p.grad - w).norm())
            worst[f"acc{acc}"] = max(rel.items(), key=lambda kv: kv[1])
            res[f"acc{acc}_in_bucket"] = [sum(1 for p in params if gs._in_place(p)), len(params)]
            res[f"acc{acc}_direct"] = gs.direct_grads - direct0
        res["worst_rel_l2"] = {k: [v[0], v[1]] for k, v in worst.items()}
        # replicas identical after construction
        chk = torch.cat([p.detach().reshape(-1)[:4] for p in params]).cpu()
        lst = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(lst, chk)
        res["replicas_identical"] = bool(all(torch.equal(lst[0], t) for t in lst))
        res["buckets"] = len(gs.buckets)
        res["ok"] = bool(res["replicas_identical"] and all(v[1] <= 1e-5 for v in worst.values()))
    except Exception as e:          # pragma: no cover
        import traceback
        res["error"] = repr(e) + "\n" + traceback.format_exc()
    finally:
        with open(f"{out_path}.rank{rank}.json", "w") as f:
            json.dump(res, f)
        dist.destroy_process_group()


if __name__ == "__main__":
    run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5])
