"""Oracle (TEST INFRASTRUCTURE): CPU restatement of the reference autoencoder family, fp32, stock torch ops.

Follows /root/reference/autoencoder/autoencoder.py:
  EncoderBlock            :6-33     Conv3x3(bias=False) -> BN -> ReLU, twice; returns (MaxPool2d(2,2)(a), a)
  Encoder                 :35-54    three blocks (din -> base, 2 base, 4 base); returns (bottleneck, skip3, skip2, skip1)
  DecoderBlockWithSkips   :57-93    ConvT2x2s2 -> centre-crop the skip if larger -> cat([up, skip]) -> DoubleConv(bias=False)
  DecoderWithSkips        :96-114
  DecoderBlockNoSkips     :117-146  ConvT2x2s2 -> DoubleConv(bias=False)
  DecoderNoSkips          :149-168
  ReconstructionAutoencoder :171-200  encoder -> DecoderNoSkips -> Conv3x3(bias) -> Sigmoid
  SegmentationEncoder     :203-268  Encoder wrapper (optional checkpoint loading / freezing)
  SegmentationAutoencoder :271-305  encoder -> DecoderWithSkips -> Conv1x1 logits

Child names are kept so that state_dicts interchange with the reference and with the product modules.  Pinned
by tests/golden/autoencoder_*.npz (generated from the imported reference by tools/gen_golden.py).
"""
import torch
from torch import nn
import torch.nn.functional as F

from .unet_ref import _bn


def _double(x, c1, b1, c2, b2):
    a1 = F.relu(_bn(F.conv2d(x, c1.weight, c1.bias, padding=1), b1))
    return F.relu(_bn(F.conv2d(a1, c2.weight, c2.bias, padding=1), b2))


class EncoderBlock(nn.Module):
    def __init__(self, din, dout):
        super().__init__()
        self.conv1 = nn.Conv2d(din, dout, 3, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(dout)
        self.relu1 = nn.ReLU()
        self.conv2 = nn.Conv2d(dout, dout, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(dout)
        self.relu2 = nn.ReLU(inplace=True)
        self.pool = nn.MaxPool2d(2, 2)

    def forward(self, x):
        skip = _double(x, self.conv1, self.bn1, self.conv2, self.bn2)
        return F.max_pool2d(skip, 2, 2), skip


class Encoder(nn.Module):
    def __init__(self, din, base_channels):
        super().__init__()
        self.encoderPart1 = EncoderBlock(din, base_channels)
        self.encoderPart2 = EncoderBlock(base_channels, base_channels * 2)
        self.encoderPart3 = EncoderBlock(base_channels * 2, base_channels * 4)

    def forward(self, x):
        p1, s1 = self.encoderPart1(x)
        p2, s2 = self.encoderPart2(p1)
        b, s3 = self.encoderPart3(p2)
        return b, s3, s2, s1


def _convs(dcin, dout):
    return nn.Sequential(nn.Conv2d(dcin, dout, 3, padding=1, bias=False), nn.BatchNorm2d(dout), nn.ReLU(inplace=True),
                         nn.Conv2d(dout, dout, 3, padding=1, bias=False), nn.BatchNorm2d(dout), nn.ReLU(inplace=True))


class DecoderBlockWithSkips(nn.Module):
    def __init__(self, din_up, din_skip, dout):
        super().__init__()
        self.up = nn.ConvTranspose2d(din_up, dout, 2, stride=2)
        self.convs = _convs(dout + din_skip, dout)

    def forward(self, x, skip):
        u = F.conv_transpose2d(x, self.up.weight, self.up.bias, stride=2)
        if skip.shape[2:] != u.shape[2:]:
            dy, dx = skip.size(2) - u.size(2), skip.size(3) - u.size(3)
            if dy < 0 or dx < 0:
                raise ValueError("Upsampled larger than skip")
            skip = skip[:, :, dy // 2: dy // 2 + u.size(2), dx // 2: dx // 2 + u.size(3)]
        s = self.convs
        return _double(torch.cat([u, skip], 1), s[0], s[1], s[3], s[4])


class DecoderWithSkips(nn.Module):
    def __init__(self, base_channels):
        super().__init__()
        c = base_channels
        self.decoderBlock1 = DecoderBlockWithSkips(c * 4, c * 4, c * 2)
        self.decoderBlock2 = DecoderBlockWithSkips(c * 2, c * 2, c)
        self.decoderBlock3 = DecoderBlockWithSkips(c, c, c)

    def forward(self, b, s3, s2, s1):
        return self.decoderBlock3(self.decoderBlock2(self.decoderBlock1(b, s3), s2), s1)


class DecoderBlockNoSkips(nn.Module):
    def __init__(self, din_up, dout):
        super().__init__()
        self.up = nn.ConvTranspose2d(din_up, dout, 2, stride=2)
        self.convs = _convs(dout, dout)

    def forward(self, x):
        s = self.convs
        return _double(F.conv_transpose2d(x, self.up.weight, self.up.bias, stride=2), s[0], s[1], s[3], s[4])


class DecoderNoSkips(nn.Module):
    def __init__(self, base_channels):
        super().__init__()
        c = base_channels
        self.decoderBlock1 = DecoderBlockNoSkips(c * 4, c * 2)
        self.decoderBlock2 = DecoderBlockNoSkips(c * 2, c)
        self.decoderBlock3 = DecoderBlockNoSkips(c, c)

    def forward(self, b):
        return self.decoderBlock3(self.decoderBlock2(self.decoderBlock1(b)))


class ReconstructionAutoencoder(nn.Module):
    def __init__(self, din, dout=3, base_channels=64):
        super().__init__()
        self.encoder = Encoder(din, base_channels)
        self.decoder = DecoderNoSkips(base_channels)
        self.decoderOut = nn.Sequential(nn.Conv2d(base_channels, dout, 3, padding=1), nn.Sigmoid())

    def forward(self, x):
        b, _, _, _ = self.encoder(x)
        d = self.decoder(b)
        return torch.sigmoid(F.conv2d(d, self.decoderOut[0].weight, self.decoderOut[0].bias, padding=1))


class SegmentationEncoder(nn.Module):
    def __init__(self, din, base_channels, pretrained_encoder_path=None, freeze_encoder=True):
        super().__init__()
        self.encoder = Encoder(din, base_channels)
        if freeze_encoder:                      # autoencoder.py:256-261
            for p in self.encoder.parameters():
                p.requires_grad = False

    def forward(self, x):
        return self.encoder(x)


class SegmentationAutoencoder(nn.Module):
    def __init__(self, din, base_channels=64, num_classes=4, pretrained_encoder_path=None, freeze_encoder=True):
        super().__init__()
        self.num_classes = num_classes
        self.encoder = SegmentationEncoder(din, base_channels, pretrained_encoder_path, freeze_encoder)
        self.decoder = DecoderWithSkips(base_channels)
        self.finalConv = nn.Conv2d(base_channels, num_classes, 1)

    def forward(self, x):
        b, s3, s2, s1 = self.encoder(x)
        return F.conv2d(self.decoder(b, s3, s2, s1), self.finalConv.weight, self.finalConv.bias)
