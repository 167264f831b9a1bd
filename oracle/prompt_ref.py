"""Oracle (TEST INFRASTRUCTURE): CPU restatement of the reference prompt model's probability remix and of its
Dice + NLL loss on probabilities.

Follows /root/reference:
  prompt_mix          prompt_based/prompt.py:33-56   softmax(clip logits), sigmoid(mask logit);
                      final[0] = 1-m; final[1:4] = m*p[0:3]; final[1] += m*p[3]
  dice_prompt         utils/weighted_loss.py:201-265 soft Dice on probs (softmax optional), one-hot by scatter,
                      sums over H,W then N, dc = (2I+s)/clip(Sp+Sg+s,1e-8), (weighted) mean over non-ignored classes
  dice_nll            utils/weighted_loss.py:311-343 dice_weight*dice + nll_weight*NLLLoss(weight, ignore_index)(nll_nonlin(x))
Pinned by tests/golden/prompt_small.npz (tools/gen_golden.py drives the imported reference's PromptModel.forward and
WeightedDiceNLLLoss).
"""
import torch
import torch.nn.functional as F


def prompt_mix(clip_logit, mask_logit):
    p = torch.softmax(clip_logit, dim=1)
    m = torch.sigmoid(mask_logit)
    sel = m * p
    return torch.cat([1.0 - m, sel[:, 0:1] + sel[:, 3:4], sel[:, 1:2], sel[:, 2:3]], dim=1)


def dice_prompt(x, y, apply_softmax=True, ignore_index=None, class_weights=None, smooth=1e-5):
    probs = torch.softmax(x, 1) if apply_softmax else x
    C = probs.shape[1]
    if y.ndim == probs.ndim - 1:
        y = y.unsqueeze(1)
    onehot = torch.zeros_like(probs).scatter_(1, y.long(), 1)
    inter = (probs * onehot).sum(dim=(2, 3)).sum(0)
    sp = probs.sum(dim=(2, 3)).sum(0)
    sg = onehot.sum(dim=(2, 3)).sum(0)
    dc = (2.0 * inter + smooth) / torch.clip(sp + sg + smooth, 1e-8)
    valid = torch.ones(C, dtype=torch.bool)
    if ignore_index is not None and 0 <= ignore_index < C:
        valid[ignore_index] = False
    if class_weights is not None:
        w = class_weights[valid]
        return -((dc[valid] * w).sum() / w.sum().clamp(min=1e-8))
    return -dc[valid].mean()


def dice_nll(outputs, targets, dice_weight=1.0, nll_weight=1.0, ignore_index=None, class_weights=None, smooth_dice=1e-5,
             apply_softmax=True, nll_nonlin=None):
    t = targets.squeeze(1) if targets.ndim == 4 else targets
    d = dice_prompt(outputs, t.unsqueeze(1), apply_softmax, ignore_index, class_weights, smooth_dice)
    z = nll_nonlin(outputs) if nll_nonlin is not None else outputs
    n = F.nll_loss(z, t.long(), weight=class_weights, ignore_index=-100 if ignore_index is None else ignore_index)
    return dice_weight * d + nll_weight * n
