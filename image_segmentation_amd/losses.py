"""Per-pixel segmentation losses on the fused HIP loss kernel -- drop-in for the reference's
utils/weighted_loss.py:6-98 (WeightedMemoryEfficientDiceLoss), :102-166 (WeightedDiceCELoss) and for
torch.nn.CrossEntropyLoss as the reference constructs it (unet/unet.ipynb cell 0; weighted_loss.py:132-138).
Constructor signatures, accepted target shapes and error behaviour follow the reference."""
from typing import Optional

import torch
from torch import nn

from . import ops


def _check_targets(outputs, targets):
    # weighted_loss.py:141-161 -- [N,H,W] or [N,1,H,W] class indices
    if targets.ndim == 3 or (targets.ndim == 4 and targets.shape[1] == 1):
        return
    if targets.ndim == outputs.ndim and targets.shape[1] != 1:
        raise ValueError(f"Target shape {targets.shape} has multiple channels but expected class indices "
                         f"[N, H, W] or [N, 1, H, W] for CE.")
    raise ValueError(f"Unsupported target shape {targets.shape} for CE. Expected [N, H, W] or [N, 1, H, W].")


class CrossEntropyLoss(nn.Module):
    """nn.CrossEntropyLoss(weight=None, ignore_index=-100, reduction='mean') for [N,C,H,W] logits."""

    def __init__(self, weight: Optional[torch.Tensor] = None, ignore_index: int = -100, reduction: str = "mean"):
        super().__init__()
        if reduction != "mean":
            raise NotImplementedError("only reduction='mean' (the reference's setting) is implemented")
        self.weight = weight
        self.ignore_index = ignore_index

    def forward(self, outputs, targets):
        ign = self.ignore_index if self.ignore_index is not None and self.ignore_index >= 0 else None
        return ops.SegLossFn.apply(outputs, targets, self.weight, ign, 0.0, 0.0, 1.0)


class WeightedMemoryEfficientDiceLoss(nn.Module):
    def __init__(self, apply_softmax: bool = True, ignore_index: Optional[int] = None,
                 class_weights: Optional[torch.Tensor] = None, smooth: float = 1e-5):
        super().__init__()
        if not apply_softmax:
            raise NotImplementedError("apply_softmax=False (prompt-model variant) is out of scope")
        self.apply_softmax = apply_softmax
        self.ignore_index = ignore_index
        self.smooth = smooth
        self.class_weights = class_weights

    def forward(self, x, y):
        # weighted_loss.py:41-46: only [N,1,H,W] targets pass the reference's shape check
        if not (y.ndim == x.ndim and y.shape[1] == 1):
            raise ValueError(f"Shape mismatch: probs {x.shape}, y {y.shape}")
        return ops.SegLossFn.apply(x, y, self.class_weights, self.ignore_index, self.smooth, 1.0, 0.0)


class WeightedDiceCELoss(nn.Module):
    def __init__(self, dice_weight: float = 1.0, ce_weight: float = 1.0, ignore_index: Optional[int] = None,
                 class_weights: Optional[torch.Tensor] = None, smooth_dice: float = 1e-5, ce_kwargs={}):
        super().__init__()
        if ce_kwargs:
            raise NotImplementedError("extra nn.CrossEntropyLoss kwargs are not supported by the fused kernel")
        self.dice_weight = dice_weight
        self.ce_weight = ce_weight
        self.ignore_index = ignore_index
        self.class_weights = class_weights
        self.smooth_dice = smooth_dice

    def forward(self, outputs, targets):
        _check_targets(outputs, targets)
        return ops.SegLossFn.apply(outputs, targets, self.class_weights, self.ignore_index, self.smooth_dice,
                                   self.dice_weight, self.ce_weight)
