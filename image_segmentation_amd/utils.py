"""Eval-time pre/post-processing helpers -- host-side restatement of the reference's utils/utils.py:13-115
(aspect-preserving resize + zero pad to a square, and its inverse).  These sit either side of the hot path
(a "next" row): they run stock torch ops on whatever device the tensors live on.
PARITY UNPINNED: the reference calls torchvision.transforms.functional.resize, which is not installed here,
so there is no oracle for the resize arithmetic; the geometry (sizes, padding, metadata) follows the source."""
from typing import List

import torch
import torch.nn.functional as F

BILINEAR, NEAREST = "bilinear", "nearest"


def resize_with_padding(image, target_size=512, interpolation=BILINEAR):
    """utils.py:13-49 -- (C,H,W) -> (C,target,target) plus metadata."""
    _, orig_h, orig_w = image.shape
    scale = min(target_size / orig_w, target_size / orig_h)
    new_w = int(round(orig_w * scale))
    new_h = int(round(orig_h * scale))
    img = image.unsqueeze(0)
    if interpolation == NEAREST or not torch.is_floating_point(image):
        resized = F.interpolate(img.float(), size=(new_h, new_w), mode="nearest").to(image.dtype)
    else:
        # torchvision's tensor resize antialiases bilinear down-scaling by default
        resized = F.interpolate(img, size=(new_h, new_w), mode="bilinear", align_corners=False, antialias=True)
    resized = resized.squeeze(0)
    pad_w, pad_h = target_size - new_w, target_size - new_h
    pad_left, pad_top = pad_w // 2, pad_h // 2
    pad_right, pad_bottom = pad_w - pad_left, pad_h - pad_top
    padded = F.pad(resized, (pad_left, pad_right, pad_top, pad_bottom), value=0)
    meta = {"original_size": (orig_h, orig_w), "new_size": (new_h, new_w),
            "pad": (pad_left, pad_top, pad_right, pad_bottom), "scale": scale}
    return padded, meta


def reverse_resize_and_padding(image, meta, interpolation="bilinear"):
    """utils.py:51-75 -- crop the padding, resize back to the original size."""
    pad_left, pad_top, _, _ = meta["pad"]
    new_h, new_w = meta["new_size"]
    cropped = image[..., pad_top: pad_top + new_h, pad_left: pad_left + new_w]
    orig_h, orig_w = meta["original_size"]
    out = F.interpolate(cropped.unsqueeze(0).float(), size=(orig_h, orig_w), mode=interpolation,
                        align_corners=False if interpolation != "nearest" else None)
    return out.squeeze(0)


def process_batch_forward(batch_images, target_size=512, interpolation=BILINEAR):
    """utils.py:77-97."""
    resized_batch, meta_list = [], []
    for image in batch_images:
        if image.ndim == 3 and image.shape[0] == 4:
            image = image[:3, ...]
        r, meta = resize_with_padding(image, target_size, interpolation)
        resized_batch.append(r)
        meta_list.append(meta)
    return torch.stack(resized_batch), meta_list


def process_batch_reverse(batch_outputs, meta_list: List[dict], interpolation="bilinear"):
    """utils.py:99-115."""
    return [reverse_resize_and_padding(o, m, interpolation=interpolation) for o, m in zip(batch_outputs, meta_list)]
