"""Training / evaluation driver -- drop-in for the reference's utils/training.py (train_loop :18-64,
eval_loop :67-121, start :453-618): same signatures, same accumulation/step/zero_grad order, same returned
averages, same checkpoint dictionary keys.  Pure host logic: the model, loss and metrics it drives are the
HIP-backed modules of this package (or anything honouring the same nn.Module protocol).

Differences from the reference, all deliberate:
  * progress bars use tqdm.auto when available (tqdm.notebook needs ipywidgets) and can be silenced;
  * eval_loop prints per-class IoU for agg.get_num_classes() classes instead of a hard-coded 4
    (training.py:81 raises IndexError with a 3-class aggregator);
  * an optional `grad_sync` hook (parallel.GradSync) all-reduces gradients over RCCL right before
    optimizer.step() -- absent in the single-process reference.
"""
import os

import numpy as np
import torch

from .metrics import MetricsHistory
from .utils import process_batch_forward, process_batch_reverse, NEAREST

try:                                    # plain tqdm; the reference's tqdm.notebook needs ipywidgets
    from tqdm.auto import tqdm as _tqdm
except Exception:                       # pragma: no cover
    _tqdm = None

VERBOSE = True


def _bar(it, **kw):
    if _tqdm is None or not VERBOSE:
        return it
    return _tqdm(it, **kw)


def _accumulate(agg, pred, label):
    """MetricsHistory.accumulate, through the sync-free device path when the aggregator offers one."""
    fn = getattr(agg, "accumulate_deferred", None)
    (fn or agg.accumulate)(pred, label)


def _say(*a):
    if VERBOSE and _rank_world()[0] == 0:
        print(*a)


def _rank_world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _reduce_eval(total_loss, num_images, agg, device):
    """Data-parallel evaluation: sum the loss, the image count and the aggregator's TP/FP/FN/TN over the ranks, so that
    every rank holds the same epoch metrics (and takes the same "improved?" decision).  Correct both when the validation
    set is sharded over the ranks and when every rank walks all of it (every count is then multiplied by the world
    size, which cancels in the loss average and in IoU / Dice / accuracy)."""
    import torch.distributed as dist
    flush = getattr(agg, "flush", None)
    if flush is not None:
        flush()
    C = agg.get_num_classes()
    dev = torch.device(device) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.zeros(2 + 4 * C, dtype=torch.float64, device=dev)
    t[0], t[1] = total_loss, num_images
    t[2:] = torch.cat([agg.total_tp, agg.total_fp, agg.total_fn, agg.total_tn]).to(dev)
    dist.all_reduce(t)
    t = t.cpu()
    for i, name in enumerate(("total_tp", "total_fp", "total_fn", "total_tn")):
        getattr(agg, name).copy_(t[2 + i * C:2 + (i + 1) * C])
    return t[0].item(), int(round(t[1].item()))


def train_loop(dataloader, model, loss_fn, optimizer, accumulation_steps, device, scheduler=None, target_size=None,
               grad_sync=None):
    """One epoch (training.py:18-64).  Returns the mean, over optimizer steps, of the UNSCALED loss of the
    last micro-batch of each accumulation window (training.py:58,62)."""
    model.train()
    total_loss = 0.0
    processed_batches = 0

    optimizer.zero_grad()

    n = len(dataloader)
    pbar = _bar(enumerate(dataloader), total=n, desc="Training")
    for batch_idx, (X, y) in pbar:
        if target_size is not None:
            X, _ = process_batch_forward(X, target_size=target_size, device=device)
            y, _ = process_batch_forward(y, target_size=target_size, interpolation=NEAREST, device=device)

        X, y = X.to(device), y.to(device).long()
        pred = model(X)
        loss = loss_fn(pred, y.squeeze(1))

        scaled_loss = loss / accumulation_steps
        stepping = (batch_idx + 1) % accumulation_steps == 0 or (batch_idx + 1) == n
        if grad_sync is not None and stepping:
            grad_sync.arm()               # overlap the RCCL all-reduce with this backward
        scaled_loss.backward()

        if stepping:
            if grad_sync is not None:
                grad_sync.sync()
            optimizer.step()
            if scheduler:
                scheduler.step()
            optimizer.zero_grad()

            total_loss += loss.item()
            processed_batches += 1
            if hasattr(pbar, "set_postfix"):
                pbar.set_postfix({'loss': loss.item(), 'lr': optimizer.param_groups[0]['lr']})

    avg_loss = total_loss / processed_batches if processed_batches > 0 else 0
    _say(f"Training Avg loss (per effective batch): {avg_loss:>8f}")
    return avg_loss


def eval_loop(dataloader, model, loss_fn, device, target_size, agg, grad_sync=None):
    """training.py:67-121: resize+pad -> model (eval mode, no_grad) -> reverse resize -> per-image loss at
    the ORIGINAL size and confusion counts.  Returns (avg_loss, mean_dice, mean_iou)."""
    model.eval()
    num_images_processed = 0
    total_loss = 0.0
    total_dev = None
    num_classes = agg.get_num_classes()
    agg.reset()

    with torch.no_grad():
        for X, y in _bar(dataloader, desc="Eval"):
            X, meta_list = process_batch_forward(X, target_size=target_size, device=device)
            X = X.to(device)
            preds = model(X)

            preds = process_batch_reverse(preds, meta_list, interpolation='bilinear')

            for pred, label in zip(preds, y):
                pred = pred.to(device)
                label = label.to(device).long()

                loss = loss_fn(pred.unsqueeze(0), label.unsqueeze(0).squeeze(1))
                if loss.is_cuda:        # device path: per-image losses and confusion counts stay on the GPU until the end
                    total_dev = loss.detach().double() if total_dev is None else total_dev + loss.detach().double()
                    _accumulate(agg, pred, label)
                else:
                    total_loss += loss.item()
                    agg.accumulate(pred, label)

                num_images_processed += 1

    if total_dev is not None:
        total_loss += total_dev.item()  # float64 sum of the float32 losses, as the reference's `+= loss.item()` builds
    if grad_sync is not None and _rank_world()[1] > 1:
        total_loss, num_images_processed = _reduce_eval(total_loss, num_images_processed, agg, device)
    avg_loss = total_loss / num_images_processed

    mean_dice, mean_iou, mean_acc = agg.compute_epoch_metrics()
    per_class_iou = agg.get_last_per_class_iou()
    ignore_index = agg.get_ignore_index()

    _say(f"\n--- Evaluation Complete ---")
    _say(f"  Images Processed: {num_images_processed}")
    _say(f"  Average Loss (Original Size): {avg_loss:>8f}")
    _say(f"  Ignored Class : {ignore_index}")
    _say(f"  Macro Avg Acc score: {mean_acc:>8f}")
    _say(f"  Macro Avg Dice Score: {mean_dice:>8f}")
    _say(f"  Mean IoU (mIoU): {mean_iou:>8f}")
    _say(f"  --- Per-Class IoU ---")
    for c in range(num_classes):
        _say(f"    Class {c}: {per_class_iou[c].item():>8f}")
    _say("-" * 25)

    return avg_loss, mean_dice, mean_iou


def train_loop_prompt(dataloader, model, loss_fn, optimizer, accumulation_steps, device, scheduler=None, target_size=None,
                      grad_sync=None):
    """One epoch of the prompt model (training.py:153-199): batches are (image, heat-map, label) triples; otherwise
    the accumulation / step / averaging protocol of train_loop."""
    model.train()
    total_loss = 0.0
    processed_batches = 0

    optimizer.zero_grad()

    n = len(dataloader)
    pbar = _bar(enumerate(dataloader), total=n, desc="Training")
    for batch_idx, (X, p, y) in pbar:
        if target_size is not None:
            X, _ = process_batch_forward(X, target_size=target_size, device=device)
            p, _ = process_batch_forward(p, target_size=target_size, device=device)
            y, _ = process_batch_forward(y, target_size=target_size, interpolation=NEAREST, device=device)

        X, p, y = X.to(device), p.to(device), y.to(device).long()
        pred = model(X, p)
        loss = loss_fn(pred, y.squeeze(1))

        scaled_loss = loss / accumulation_steps
        stepping = (batch_idx + 1) % accumulation_steps == 0 or (batch_idx + 1) == n
        if grad_sync is not None and stepping:
            grad_sync.arm()
        scaled_loss.backward()

        if stepping:
            if grad_sync is not None:
                grad_sync.sync()
            optimizer.step()
            if scheduler:
                scheduler.step()
            optimizer.zero_grad()

            total_loss += loss.item()
            processed_batches += 1
            if hasattr(pbar, "set_postfix"):
                pbar.set_postfix({'loss': loss.item(), 'lr': optimizer.param_groups[0]['lr']})

    avg_loss = total_loss / processed_batches if processed_batches > 0 else 0
    _say(f"Training Avg loss (per effective batch): {avg_loss:>8f}")
    return avg_loss


def eval_loop_prompt(dataloader, model, loss_fn, device, target_size, agg, grad_sync=None):
    """training.py:242-296: eval_loop for (image, heat-map, label) batches; the heat-map takes the image's
    resize + pad.  Returns (avg_loss, mean_dice, mean_iou)."""
    model.eval()
    num_images_processed = 0
    total_loss = 0.0
    total_dev = None
    num_classes = agg.get_num_classes()

    with torch.no_grad():
        for X, p, y in _bar(dataloader, desc="Eval"):
            X, meta_list = process_batch_forward(X, target_size=target_size, device=device)
            p, _ = process_batch_forward(p, target_size=target_size, device=device)
            X, p = X.to(device), p.to(device)
            preds = model(X, p)

            preds = process_batch_reverse(preds, meta_list, interpolation='bilinear')

            for pred, label in zip(preds, y):
                pred = pred.to(device)
                label = label.to(device).long()

                loss = loss_fn(pred.unsqueeze(0), label.unsqueeze(0).squeeze(1))
                if loss.is_cuda:        # device path: per-image losses and confusion counts stay on the GPU until the end
                    total_dev = loss.detach().double() if total_dev is None else total_dev + loss.detach().double()
                    _accumulate(agg, pred, label)
                else:
                    total_loss += loss.item()
                    agg.accumulate(pred, label)

                num_images_processed += 1

    if total_dev is not None:
        total_loss += total_dev.item()  # float64 sum of the float32 losses, as the reference's `+= loss.item()` builds
    if grad_sync is not None and _rank_world()[1] > 1:
        total_loss, num_images_processed = _reduce_eval(total_loss, num_images_processed, agg, device)
    avg_loss = total_loss / num_images_processed

    mean_dice, mean_iou, mean_acc = agg.compute_epoch_metrics()
    per_class_iou = agg.get_last_per_class_iou()
    ignore_index = agg.get_ignore_index()

    _say(f"\n--- Evaluation Complete ---")
    _say(f"  Images Processed: {num_images_processed}")
    _say(f"  Average Loss (Original Size): {avg_loss:>8f}")
    _say(f"  Ignored Class : {ignore_index}")
    _say(f"  Macro Avg Acc score: {mean_acc:>8f}")
    _say(f"  Macro Avg Dice Score: {mean_dice:>8f}")
    _say(f"  Mean IoU (mIoU): {mean_iou:>8f}")
    _say(f"  --- Per-Class IoU ---")
    for c in range(num_classes):
        _say(f"    Class {c}: {per_class_iou[c].item():>8f}")
    _say("-" * 25)

    return avg_loss, mean_dice, mean_iou


def start(*args, **kwargs):
    """training.py:453-618: optional resume, epoch loop, per-epoch metrics file, best-mIoU checkpoint
    (+ weights-only "MO_<name>").  Checkpoint keys are the reference's."""
    return _start(False, *args, **kwargs)


def start_prompt(*args, **kwargs):
    """training.py:299-450: `start` for the prompt model -- the prompt loops, a full (pickled) checkpoint load, the
    metrics history stored inside the checkpoint and no weights-only "MO_" file."""
    return _start(True, *args, **kwargs)


def _start(
        prompt: bool,
        model_save_dir: str,
        model_save_name: str,
        model,
        optimizer,
        train_dataloader,
        val_dataloader,
        accumulation_steps: int,
        device,
        train_loss_fn,
        val_loss_fn,
        target_size: int,
        scheduler=None,
        agg: MetricsHistory = None,
        load: bool = True,
        save: bool = True,
        num_classes: int = 4,
        ignore_index: int = 3,
        epochs: int = 100,
        grad_sync=None,
):
    start_epoch = 0
    best_dev_dice = -np.inf
    best_dev_miou = -np.inf
    best_dev_loss = np.inf
    # data parallel (grad_sync given, one process per GPU): every rank loads the same checkpoint, rank 0's BatchNorm
    # running statistics are broadcast before evaluation (they are per replica during training), the evaluation
    # counts are summed over the ranks so that all ranks take the same decision, and ONLY rank 0 writes files.
    rank, world = _rank_world()

    os.makedirs(model_save_dir, exist_ok=True)
    os.makedirs(f"{model_save_dir}/metrics", exist_ok=True)
    path = f"{model_save_dir}/{model_save_name}"
    if load and os.path.isfile(path):
        _say(f"Loading checkpoint from: {path}")
        checkpoint = torch.load(path, map_location=device, weights_only=not prompt)   # training.py:351 vs :506
        model.load_state_dict(checkpoint["model_state_dict"])
        _say(" -> Model state loaded.")
        try:
            optimizer.load_state_dict(checkpoint["optimizer_state_dict"])
            _say(" -> Optimizer state loaded.")
        except Exception as e:
            _say(f" -> Warning: Could not load optimizer state: {e}. Optimizer will start from scratch.")
        try:
            scheduler.load_state_dict(checkpoint["scheduler_state_dict"])
            _say(" -> Scheduler state loaded.")
        except Exception as e:
            _say(f" -> Warning: Could not load scheduler state: {e}. Scheduler will start from scratch.")
        try:
            agg = checkpoint.get("history")
            agg.to(device)
            _say(" -> Metrics History loaded.")
        except Exception:
            _say(" -> No metric history saved")
            agg = MetricsHistory(num_classes, ignore_index)
        start_epoch = checkpoint.get("epoch", 0)
        best_dev_dice = checkpoint.get("best_dev_dice", -np.inf)
        best_dev_miou = checkpoint.get("best_dev_miou", -np.inf)
        best_dev_loss = checkpoint.get("best_dev_loss", np.inf)
        _say(f" -> Resuming training from epoch {start_epoch + 1}")
        _say(f" -> Loaded best metrics: Dice={best_dev_dice:.6f}, mIoU={best_dev_miou:.6f}, Loss={best_dev_loss:.6f}")
        _say(f" -> Notes from checkpoint: {checkpoint.get('notes', 'N/A')}")
    else:
        _say(f"Checkpoint file not found at {path}. Starting training from scratch.")
    if agg is None:
        agg = MetricsHistory(num_classes, ignore_index)

    _say("\nStarting Training...")
    for t in range(start_epoch, epochs):
        _say(f"Epoch {t+1}\n-------------------------------")
        tl, el = (train_loop_prompt, eval_loop_prompt) if prompt else (train_loop, eval_loop)
        tl(train_dataloader, model, train_loss_fn, optimizer, accumulation_steps, device, scheduler, target_size,
           grad_sync=grad_sync)
        if grad_sync is not None:
            grad_sync.broadcast_buffers(model)
            val_loss, val_dice, val_miou = el(val_dataloader, model, val_loss_fn, device, target_size, agg,
                                              grad_sync=grad_sync)
        else:
            val_loss, val_dice, val_miou = el(val_dataloader, model, val_loss_fn, device, target_size, agg)
        writer = save and rank == 0

        if writer:
            torch.save({"epoch": t + 1, "history": agg}, f"{model_save_dir}/metrics/{model_save_name}")

        if val_miou > best_dev_miou:
            best_dev_dice, best_dev_miou, best_dev_loss = val_dice, val_miou, val_loss
            if writer:
                _say(f"Validation IoU score improved ({best_dev_miou:.6f}). Saving model...")
                checkpoint = {
                    "epoch": t + 1,
                    "model_state_dict": model.state_dict(),
                    "optimizer_state_dict": optimizer.state_dict(),
                    "best_dev_dice": best_dev_dice,
                    "best_dev_miou": best_dev_miou,
                    "best_dev_loss": best_dev_loss,
                    "notes": f"Model saved based on best Micro Dice. Ignored index for metric: {ignore_index}",
                }
                if scheduler:
                    checkpoint["scheduler_state_dict"] = scheduler.state_dict()
                if prompt:
                    checkpoint["history"] = agg                      # training.py:424
                torch.save(checkpoint, path)
                if not prompt:
                    torch.save({"epoch": t + 1, "model_state_dict": model.state_dict()},
                               f"{model_save_dir}/MO_{model_save_name}")
        else:
            _say(f"Validation IoU score did not improve from {best_dev_miou:.6f}")
        if world > 1 and save:
            import torch.distributed as dist
            dist.barrier()                # nobody reads or overwrites a file rank 0 is still writing

    _say("\n--- Training Finished! ---")
    _say(f"Best validation IoU score achieved: {best_dev_miou:.6f}")
    _say(f"Corresponding validation dice: {best_dev_dice:.6f}")
    _say(f"Corresponding validation loss: {best_dev_loss:.6f}")
    return best_dev_miou, best_dev_dice, best_dev_loss
