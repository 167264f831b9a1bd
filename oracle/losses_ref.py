"""Oracle (TEST INFRASTRUCTURE): CPU restatement of the reference losses and metrics.

Follows /root/reference/utils/weighted_loss.py:6-98 (soft Dice), :102-166 (Dice + CE) and
/root/reference/utils/MetricsHistory.py:55-128.  Written as plain functions on fp32 tensors.
"""
import torch
import torch.nn.functional as F


def cross_entropy(logits, target, weight=None, ignore_index=None):
    """nn.CrossEntropyLoss(mean) as used at training.py:47 / weighted_loss.py:138,163:
    sum(w[y]*nll)/sum(w[y]) over non-ignored pixels."""
    lsm = F.log_softmax(logits.float(), dim=1)
    t = target.long()
    valid = torch.ones_like(t, dtype=torch.bool) if ignore_index is None else (t != ignore_index)
    tt = torch.where(valid, t, torch.zeros_like(t))
    nll = -lsm.gather(1, tt.unsqueeze(1)).squeeze(1)
    w = torch.ones(logits.shape[1]) if weight is None else weight.float()
    wy = w[tt] * valid
    return (wy * nll).sum() / wy.sum()


def soft_dice(logits, target, smooth=1e-5, class_weights=None, ignore_index=None):
    """weighted_loss.py:31-98. target [N,H,W] or [N,1,H,W]; returns -dice."""
    c = logits.shape[1]
    p = F.softmax(logits.float(), dim=1)                               # :36
    t = target.long()
    if t.dim() != 4 or t.shape[1] != 1:
        # :42-46 -- a [N,H,W] target never satisfies `shp_y == probs.shape[2:]`, so the
        # reference raises for it; only [N,1,H,W] is accepted by the bare Dice loss.
        raise ValueError(f"Shape mismatch: probs {tuple(p.shape)}, y {tuple(target.shape)}")
    onehot = torch.zeros_like(p).scatter_(1, t, 1.0)                   # :57-58
    inter = (p * onehot).sum(dim=(2, 3)).sum(0)                        # :66,71
    sp = p.sum(dim=(2, 3)).sum(0)                                      # :67,72
    sg = onehot.sum(dim=(2, 3)).sum(0)                                 # :61,73
    dc = (2.0 * inter + smooth) / torch.clip(sp + sg + smooth, 1e-8)   # :76-77
    valid = torch.ones(c, dtype=torch.bool)
    if ignore_index is not None and 0 <= ignore_index < c:
        valid[ignore_index] = False                                    # :79-81
    dcv = dc[valid]
    if class_weights is not None:
        wv = class_weights.float()[valid]
        return -((dcv * wv).sum() / wv.sum().clamp(min=1e-8))          # :87-94
    return -dcv.mean()                                                 # :96


def dice_ce(logits, target, dice_weight=1.0, ce_weight=1.0, ignore_index=None,
            class_weights=None, smooth_dice=1e-5):
    """weighted_loss.py:140-166."""
    t = target.squeeze(1) if target.dim() == 4 else target                 # :141-161
    d = soft_dice(logits, t.unsqueeze(1), smooth_dice, class_weights, ignore_index)
    ce = cross_entropy(logits, t, class_weights, ignore_index)
    return dice_weight * d + ce_weight * ce


def confusion_counts(pred, label, num_classes):
    """MetricsHistory.py:65-75: argmax over C then per-class TP/FP/FN/TN (float64)."""
    hard = torch.argmax(pred, dim=0)
    lab = label.long()
    out = torch.zeros(4, num_classes, dtype=torch.float64)
    for k in range(num_classes):
        ph, lh = hard == k, lab == k
        out[0, k] = (ph & lh).sum()
        out[1, k] = (ph & ~lh).sum()
        out[2, k] = (~ph & lh).sum()
        out[3, k] = (~ph & ~lh).sum()
    return out


def epoch_metrics(counts, ignore_index=None):
    """MetricsHistory.py:100-113: IoU tp/(tp+fp+fn), Dice 2tp/(2tp+fp+fn), Acc; macro mean over
    non-ignored classes; no epsilon (NaN when a class is absent)."""
    tp, fp, fn, tn = counts
    iou = tp / (tp + fp + fn)
    dice = 2 * tp / (2 * tp + fp + fn)
    acc = (tp + tn) / (tp + tn + fp + fn)
    m = torch.ones(tp.numel(), dtype=torch.bool)
    if ignore_index is not None and 0 <= ignore_index < tp.numel():
        m[ignore_index] = False
    return dice[m].mean().item(), iou[m].mean().item(), acc[m].mean().item(), iou
