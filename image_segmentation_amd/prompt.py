"""Prompt-based segmentation model on the MI355X kernels -- drop-in for the reference's prompt_based/prompt.py:6-56.

PromptModel(path=None): a frozen 4-class ClipUNet (`self.clip`) and a trainable mask U-Net `unet(4, 1)` (`self.mask`)
fed the image concatenated with a click heat-map; the class probabilities are the CLIP softmax remixed with the mask
sigmoid (segk_prompt_mix_fwd/bwd).  Child names (`clip.*`, `mask.*`) and the checkpoint behaviour
(`checkpoint["model_state_dict"]` loaded into `self.clip`) follow the reference.  `clip=` injects an already built
ClipUNet (offline use: the default constructor needs the hub, like the reference's)."""
import torch
from torch import nn

from . import ops
from .clipunet import ClipUNet
from .unet import unet


class PromptModel(nn.Module):
    def __init__(self, path=None, clip=None):
        super().__init__()
        self.clip = clip if clip is not None else ClipUNet()
        self.mask = unet(4, 1)
        self.softmax = nn.Softmax(dim=1)        # parameter-free children kept for module-tree parity (prompt.py:17-18)
        self.sigmoid = nn.Sigmoid()
        if path is not None:
            try:
                checkpoint = torch.load(path, weights_only=False, map_location=lambda storage, loc: storage)
                self.clip.load_state_dict(checkpoint["model_state_dict"])
            except Exception as e:
                print(f"Error loading checkpoint: {str(e)[:200]}")
                raise
        for param in self.clip.parameters():
            param.requires_grad = False

    def forward(self, x, heatmap):
        clip_logit = self.clip(x)                                  # frozen branch: no autograd graph is recorded
        mask_logit = self.mask(torch.concat([x, heatmap], dim=1))
        return ops.PromptMixFn.apply(clip_logit, mask_logit)
