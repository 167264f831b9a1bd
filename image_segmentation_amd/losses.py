"""Per-pixel segmentation losses on the fused HIP loss kernel -- drop-in for the reference's
utils/weighted_loss.py:6-98 (WeightedMemoryEfficientDiceLoss), :102-166 (WeightedDiceCELoss), :170-265
(WeightedMemoryEfficientDiceLossPrompt), :268-343 (WeightedDiceNLLLoss) and for torch.nn.CrossEntropyLoss as the reference constructs it (unet/unet.ipynb cell 0; weighted_loss.py:132-138).
Constructor signatures, accepted target shapes and error behaviour follow the reference."""
from typing import Callable, Optional

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
            raise NotImplementedError("apply_softmax=False: use WeightedMemoryEfficientDiceLossPrompt (weighted_loss.py:170)")
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


def _log_eps(fn):
    """Map an `nll_nonlin` callable onto the kernel's NLL input: None -> (0, 0.0) (NLL of the values themselves);
    x -> log(x + eps) -> (1, eps), recognised by probing the callable (prompt.ipynb: `lambda x: torch.log(x + 1e-9)`)."""
    if fn is None:
        return 0, 0.0
    with torch.no_grad():
        x = torch.tensor([0.0, 0.25, 1.0], dtype=torch.float64)
        y = fn(x)
        eps = float(torch.exp(y[0]))
        ok = bool(torch.isfinite(y[1:]).all()) and torch.allclose(y[1:], torch.log(x[1:] + eps), rtol=1e-9, atol=1e-12)
        if eps > 0 and not torch.isfinite(y[0]):
            ok = False
    if not ok or eps < 0 or eps > 1e-2:
        raise NotImplementedError("nll_nonlin must be None or x -> log(x + eps): other non-linearities have no HIP kernel")
    return 1, eps


class WeightedMemoryEfficientDiceLossPrompt(nn.Module):
    def __init__(self, dice_nonlin: Callable = None, apply_softmax: bool = True, ignore_index: Optional[int] = None,
                 class_weights: Optional[torch.Tensor] = None, smooth: float = 1e-5):
        super().__init__()
        if dice_nonlin is not None:
            raise NotImplementedError("dice_nonlin has no HIP kernel (the reference never sets it)")
        self.apply_softmax = apply_softmax
        self.ignore_index = ignore_index
        self.smooth = smooth
        self.dice_nonlin = dice_nonlin
        if class_weights is not None:
            assert isinstance(class_weights, torch.Tensor), "class_weights must be a torch.Tensor"
        self.class_weights = class_weights

    def forward(self, x, y):
        # weighted_loss.py:213-217: only [N,1,H,W] passes (the [N,H,W] branch compares y.shape with probs.shape[2:],
        # which can never match, so the reference raises for it)
        if y.ndim != x.ndim:
            raise ValueError(f"Shape mismatch: probs {x.shape}, y {y.shape}")
        elif y.shape[1] != 1:
            raise NotImplementedError("one-hot targets are not supported by the fused kernel")
        if self.apply_softmax:
            return ops.SegLossFn.apply(x, y, self.class_weights, self.ignore_index, self.smooth, 1.0, 0.0)
        return ops.ProbLossFn.apply(x, y, self.class_weights, self.ignore_index, self.smooth, 1.0, 0.0, 0, 0.0)


class WeightedDiceNLLLoss(nn.Module):
    def __init__(self, dice_weight: float = 1.0, nll_weight: float = 1.0, ignore_index: Optional[int] = None,
                 class_weights: Optional[torch.Tensor] = None, smooth_dice: float = 1e-5, apply_softmax: bool = True,
                 dice_nonlin: Callable = None, nll_nonlin: Callable = None, nll_kwargs={}):
        super().__init__()
        if nll_kwargs:
            raise NotImplementedError("extra nn.NLLLoss kwargs are not supported by the fused kernel")
        # like the reference (weighted_loss.py:296-301) dice_nonlin is stored but never reaches the Dice term
        self.dice_weight = dice_weight
        self.nll_weight = nll_weight
        self.ignore_index = ignore_index
        self.class_weights = class_weights
        self.smooth_dice = smooth_dice
        self.apply_softmax = apply_softmax
        self.dice_nonlin = dice_nonlin
        self.nll_nonlin = nll_nonlin
        self._nll_log, self._eps = _log_eps(nll_nonlin)

    def forward(self, outputs, targets):
        _check_targets(outputs, targets)
        if not self.apply_softmax:      # one pass: Dice and NLL of the same probabilities
            return ops.ProbLossFn.apply(outputs, targets, self.class_weights, self.ignore_index, self.smooth_dice,
                                        self.dice_weight, self.nll_weight, self._nll_log, self._eps)
        # reference default: Dice of softmax(outputs) + NLL of nll_nonlin(outputs)
        dice = ops.SegLossFn.apply(outputs, targets, self.class_weights, self.ignore_index, self.smooth_dice, 1.0, 0.0)
        nll = ops.ProbLossFn.apply(outputs, targets, self.class_weights, self.ignore_index, self.smooth_dice, 0.0, 1.0,
                                   self._nll_log, self._eps)
        return self.dice_weight * dice + self.nll_weight * nll
