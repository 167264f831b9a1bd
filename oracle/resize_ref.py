"""Oracle (TEST INFRASTRUCTURE): CPU restatement of the reference's eval-time pre/post-processing.

Follows /root/reference/utils/utils.py:
  resize_with_padding          :13-49   scale = min(T/w, T/h); new size by round(); resize; centred zero padding
  reverse_resize_and_padding   :51-75   crop the window, F.interpolate back to the original size
  process_batch_forward/reverse :77-115 per-image loops (4-channel images keep their first 3 channels)
The resize itself is torchvision.transforms.functional.resize in the reference -- a third-party dependency that is
not installed in this image (no wheel, no network) and that the reference does not pin.  Its published tensor branch
(torchvision/transforms/_functional_tensor.py: resize) is torch.nn.functional.interpolate(img, size, mode,
align_corners=False for bilinear, antialias=True for bilinear from torchvision 0.17 on, antialias=False before) with
integer images cast to float and back; that call is restated here with the switch exposed (`antialias=`).  PARITY against torchvision itself is UNPINNED (nothing to run or capture); the geometry and the
ATen arithmetic are what the GPU tests hold the HIP kernels to.
"""
import torch
import torch.nn.functional as F


def resize_with_padding(image, target_size=512, nearest=False, antialias=True):
    _, h, w = image.shape
    scale = min(target_size / w, target_size / h)
    nw, nh = int(round(w * scale)), int(round(h * scale))
    img = image.unsqueeze(0)
    if nearest or not torch.is_floating_point(image):
        r = F.interpolate(img.float(), size=(nh, nw), mode="nearest").to(image.dtype)
    else:
        r = F.interpolate(img, size=(nh, nw), mode="bilinear", align_corners=False, antialias=antialias)
    pw, ph = target_size - nw, target_size - nh
    pl, pt = pw // 2, ph // 2
    out = F.pad(r.squeeze(0), (pl, pw - pl, pt, ph - pt), value=0)
    return out, {"original_size": (h, w), "new_size": (nh, nw), "pad": (pl, pt, pw - pl, ph - pt), "scale": scale}


def reverse_resize_and_padding(image, meta, interpolation="bilinear"):
    pl, pt, _, _ = meta["pad"]
    nh, nw = meta["new_size"]
    crop = image[..., pt:pt + nh, pl:pl + nw]
    return F.interpolate(crop.unsqueeze(0), size=meta["original_size"], mode=interpolation,
                         align_corners=False if interpolation != "nearest" else None).squeeze(0)


def process_batch_forward(images, target_size=512, nearest=False, antialias=True):
    outs, metas = [], []
    for im in images:
        if im.ndim == 3 and im.shape[0] == 4:
            im = im[:3]
        o, m = resize_with_padding(im, target_size, nearest, antialias)
        outs.append(o); metas.append(m)
    return torch.stack(outs), metas
