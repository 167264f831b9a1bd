"""Oracle (TEST INFRASTRUCTURE): CPU restatement of the reference CLIP-UNet decoder.

Follows /root/reference/clip/clipunet.py:
  DecoderBlock  clipunet.py:68-105   ConvT2x2s2(in->in/2); Conv1x1(skip 768->in/2);
                                     bilinear(skip -> x.size, align_corners=False);
                                     cat([x, skip]) (upsampled FIRST); Conv3x3(no bias)->BN->ReLU x2
  UNetDecoder   clipunet.py:108-144  init_conv 1x1 then blocks over reversed(skips)
The ViT encoder (clipunet.py:7-65) is third-party (transformers.CLIPVisionModel): its restatement is
oracle/clip_vit_ref.py; pretrained weights cannot be fetched offline, so pretrained parity is UNPINNED and the
decoder is pinned with portable-fill features (SURVEY.md 8c answers C, D).
"""
import torch
from torch import nn
import torch.nn.functional as F
from .unet_ref import _bn


class DecoderBlock(nn.Module):
    def __init__(self, in_channels, in_channels_skip, out_channels):
        super().__init__()
        self.upsample = nn.ConvTranspose2d(in_channels, in_channels // 2, 2, stride=2)
        self.skip_conv = nn.Conv2d(in_channels_skip, in_channels // 2, 1)
        self.conv_block = nn.Sequential(
            nn.Conv2d(in_channels, out_channels, 3, padding=1, bias=False),
            nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True),
            nn.Conv2d(out_channels, out_channels, 3, padding=1, bias=False),
            nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True))

    def forward(self, x, skip):
        x = F.conv_transpose2d(x, self.upsample.weight, self.upsample.bias, stride=2)
        skip = F.conv2d(skip, self.skip_conv.weight, self.skip_conv.bias)
        if skip.shape[2:] != x.shape[2:]:
            skip = F.interpolate(skip, size=x.shape[2:], mode="bilinear", align_corners=False)
        x = torch.cat([x, skip], dim=1)
        s = self.conv_block
        x = F.relu(_bn(F.conv2d(x, s[0].weight, None, padding=1), s[1]))
        return F.relu(_bn(F.conv2d(x, s[3].weight, None, padding=1), s[4]))


class UNetDecoder(nn.Module):
    def __init__(self, encoder_hidden_dim, decoder_channels):
        super().__init__()
        self.init_conv = nn.Conv2d(encoder_hidden_dim, decoder_channels[0], 1)
        self.decoder_blocks = nn.ModuleList()
        c = decoder_channels[0]
        for out_ch in decoder_channels[1:]:
            self.decoder_blocks.append(DecoderBlock(c, encoder_hidden_dim, out_ch))
            c = out_ch

    def forward(self, x, skips):
        x = F.conv2d(x, self.init_conv.weight, self.init_conv.bias)
        for block, skip in zip(self.decoder_blocks, reversed(skips)):
            x = block(x, skip)
        return x


def tokens_to_grid(hidden_state, grid):
    """clipunet.py:48-51,54-63: drop CLS, [B,196,768] -> [B,768,14,14]."""
    b, _, d = hidden_state.shape
    return hidden_state[:, 1:, :].reshape(b, grid, grid, d).permute(0, 3, 1, 2).contiguous()
