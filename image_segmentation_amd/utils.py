"""Eval-time pre/post-processing -- drop-in for the reference's utils/utils.py:13-115 (aspect-preserving resize +
zero pad to a square, and its inverse), the steps either side of the eval forward.

Two paths, chosen by where the caller wants the result:
  * `device=` a CUDA device (what train_loop / eval_loop pass): each image is uploaded as it is and resized +
    padded straight into its slot of the network batch by the HIP kernel segk_resize_pad; the reverse step
    (segk_crop_resize) works on the CUDA logits without leaving the device.  No stock-torch resize runs.
  * no device / CPU tensors (a data loader preparing batches on the host, as the reference does): the stock torch
    ops below, byte-for-byte what the reference's helpers call.
Arithmetic: the reference resizes with torchvision's TF.resize, which is not installed here; its tensor branch is
F.interpolate(mode, align_corners=False, antialias=True) (ATen), and that is what both paths compute
(oracle/resize_ref.py; PARITY against torchvision itself stays UNPINNED, against ATen it is pinned by
tests/test_gpu_evalpipe.py).  The geometry (sizes, rounding, padding, metadata) follows the source."""
from typing import List

import torch
import torch.nn.functional as F

from . import _lib

BILINEAR, NEAREST = "bilinear", "nearest"
# Bilinear down-scaling with or without the anti-aliasing (triangle) filter.  The reference calls torchvision's
# TF.resize(image, size, interpolation=BILINEAR) on tensors (utils/utils.py:28) and pins no torchvision version:
# from 0.17 on that call anti-aliases by default (antialias=True), before it did not (antialias=None -> False on tensors,
# with a deprecation warning).  True reproduces a current torchvision; pass antialias=False (or set this default) to
# reproduce an older one.  Up-scaling and nearest are unaffected.
ANTIALIAS = True


def _geometry(orig_h, orig_w, target_size):
    """utils.py:25-40: new size, padding and the metadata dictionary."""
    scale = min(target_size / orig_w, target_size / orig_h)
    new_w = int(round(orig_w * scale))
    new_h = int(round(orig_h * scale))
    pad_w, pad_h = target_size - new_w, target_size - new_h
    pad_left, pad_top = pad_w // 2, pad_h // 2
    pad_right, pad_bottom = pad_w - pad_left, pad_h - pad_top
    meta = {"original_size": (orig_h, orig_w), "new_size": (new_h, new_w),
            "pad": (pad_left, pad_top, pad_right, pad_bottom), "scale": scale}
    return new_h, new_w, pad_top, pad_left, meta


def _is_cuda_device(device):
    return device is not None and torch.device(device).type == "cuda"


def _resize_pad_into(image, slot, target_size, interpolation, antialias=None):
    """image (C,H,W) on the slot's device -> slot (C,T,T) (HIP kernel); returns the metadata."""
    antialias = ANTIALIAS if antialias is None else antialias
    C, H, W = image.shape
    nh, nw, pt, pl, meta = _geometry(H, W, target_size)
    integer = not torch.is_floating_point(image)
    src = image.contiguous() if (image.dtype in (torch.float32, torch.int64)) else \
        (image.long().contiguous() if integer else image.float().contiguous())
    mode = 1 if (interpolation == NEAREST or integer) else (0 if antialias else 2)
    _lib.call("segk_resize_pad", src.data_ptr(), slot.data_ptr(), C, H, W, nh, nw, target_size, pt, pl, mode,
              1 if integer else 0, torch.cuda.current_stream().cuda_stream)
    return meta


def resize_with_padding(image, target_size=512, interpolation=BILINEAR, antialias=None):
    """utils.py:13-49 -- (C,H,W) -> (C,target,target) plus metadata.  antialias: see ANTIALIAS."""
    antialias = ANTIALIAS if antialias is None else antialias
    _, orig_h, orig_w = image.shape
    scale = min(target_size / orig_w, target_size / orig_h)
    new_w = int(round(orig_w * scale))
    new_h = int(round(orig_h * scale))
    img = image.unsqueeze(0)
    if interpolation == NEAREST or not torch.is_floating_point(image):
        resized = F.interpolate(img.float(), size=(new_h, new_w), mode="nearest").to(image.dtype)
    else:
        resized = F.interpolate(img, size=(new_h, new_w), mode="bilinear", align_corners=False, antialias=bool(antialias))
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


def process_batch_forward(batch_images, target_size=512, interpolation=BILINEAR, device=None, antialias=None):
    """utils.py:77-97.  With a CUDA `device` the batch is produced on it by the HIP kernel (float images stay
    float32, integer label maps come back int64); otherwise the host path."""
    if _is_cuda_device(device) or (len(batch_images) and isinstance(batch_images[0], torch.Tensor) and batch_images[0].is_cuda):
        dev = torch.device(device) if device is not None else batch_images[0].device
        imgs = []
        for image in batch_images:
            if image.ndim == 3 and image.shape[0] == 4:
                image = image[:3, ...]
            imgs.append(image.to(dev, non_blocking=True))
        integer = not torch.is_floating_point(imgs[0])
        batch = torch.empty((len(imgs), imgs[0].shape[0], target_size, target_size),
                            dtype=torch.int64 if integer else torch.float32, device=dev)
        with torch.cuda.device(dev):
            meta_list = [_resize_pad_into(im, batch[i], target_size, interpolation, antialias) for i, im in enumerate(imgs)]
        return batch, meta_list
    resized_batch, meta_list = [], []
    for image in batch_images:
        if image.ndim == 3 and image.shape[0] == 4:
            image = image[:3, ...]
        r, meta = resize_with_padding(image, target_size, interpolation, antialias)
        resized_batch.append(r)
        meta_list.append(meta)
    return torch.stack(resized_batch), meta_list


def process_batch_reverse(batch_outputs, meta_list: List[dict], interpolation="bilinear"):
    """utils.py:99-115.  CUDA outputs are cropped and resized by the HIP kernel."""
    if isinstance(batch_outputs, torch.Tensor) and batch_outputs.is_cuda:
        outs = batch_outputs.detach()
        if outs.dtype != torch.float32 or not outs.is_contiguous():
            outs = outs.float().contiguous()
        N, C, T, T2 = outs.shape
        if T != T2:
            raise ValueError(f"process_batch_reverse expects square network outputs, got {tuple(outs.shape)}")
        mode = 1 if interpolation == "nearest" else 0
        res = []
        with torch.cuda.device(outs.device):
            s = torch.cuda.current_stream().cuda_stream
            for n, meta in enumerate(meta_list):
                pl, pt, _, _ = meta["pad"]
                nh, nw = meta["new_size"]
                oh, ow = meta["original_size"]
                o = torch.empty((C, oh, ow), dtype=torch.float32, device=outs.device)
                _lib.call("segk_crop_resize", outs[n].data_ptr(), o.data_ptr(), C, T, pt, pl, nh, nw, oh, ow, mode, s)
                res.append(o)
        return res
    return [reverse_resize_and_padding(o, m, interpolation=interpolation) for o, m in zip(batch_outputs, meta_list)]
