"""Oracle (TEST INFRASTRUCTURE): CPU restatement of the reference U-Net, fp32, stock torch ops.

Follows /root/reference/unet/unet.py:
  DoubleConvReLU  unet.py:4-25   Conv3x3(pad1,bias) -> BN(eps 1e-5, mom 0.1) -> ReLU, twice
  Down            unet.py:28-45  MaxPool2d(2,2) then DoubleConvReLU
  Up              unet.py:47-64  cat([x1, ConvT2x2s2(x2)], C) then DoubleConvReLU(din, dout)
  unet            unet.py:67-105 5 encoder levels (64..1024), 4 decoder levels, 1x1 head

Child names are kept (doubleConvReLU.{0,1,3,4}, maxpool_doubleConv.1, upsample, doubleConv,
output) so that state_dicts interchange with the reference and with the product modules; the
arithmetic is written with torch.nn.functional calls.  Pinned by tests/golden/*.npz
(generated from the imported reference by tools/gen_golden.py) and SURVEY.md 8c answers A, B.
"""
import torch
from torch import nn
import torch.nn.functional as F


def _bn(x, bn: nn.BatchNorm2d):
    # training: batch statistics (biased var to normalise, unbiased into running_var),
    # num_batches_tracked += 1; eval: running statistics.  unet.py:17,20
    if bn.training:
        bn.num_batches_tracked.add_(1)
    return F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias,
                        bn.training, bn.momentum, bn.eps)


class DoubleConvReLU(nn.Module):
    def __init__(self, din, dout, bias=True):
        super().__init__()
        # containers only; index 2 and 5 are the parameter-free ReLUs (unet.py:15-22)
        self.doubleConvReLU = nn.Sequential(
            nn.Conv2d(din, dout, 3, padding=1, bias=bias), nn.BatchNorm2d(dout), nn.ReLU(),
            nn.Conv2d(dout, dout, 3, padding=1, bias=bias), nn.BatchNorm2d(dout), nn.ReLU())

    def forward(self, x):
        s = self.doubleConvReLU
        z1 = F.conv2d(x, s[0].weight, s[0].bias, padding=1)
        a1 = F.relu(_bn(z1, s[1]))
        z2 = F.conv2d(a1, s[3].weight, s[3].bias, padding=1)
        return F.relu(_bn(z2, s[4]))


class Down(nn.Module):
    def __init__(self, din, dout):
        super().__init__()
        self.maxpool_doubleConv = nn.Sequential(nn.MaxPool2d(2, 2), DoubleConvReLU(din, dout))

    def forward(self, x):
        return self.maxpool_doubleConv[1](F.max_pool2d(x, 2, 2))


class Up(nn.Module):
    def __init__(self, din, dout):
        super().__init__()
        self.upsample = nn.ConvTranspose2d(din, dout, 2, stride=2)
        self.doubleConv = DoubleConvReLU(din, dout)

    def forward(self, x1, x2):
        u = F.conv_transpose2d(x2, self.upsample.weight, self.upsample.bias, stride=2)
        return self.doubleConv(torch.cat([x1, u], dim=1))   # skip first (unet.py:63)


class unet(nn.Module):
    def __init__(self, din, dout):
        super().__init__()
        self.scale = 1
        self.down1 = DoubleConvReLU(din, 64)
        self.down2 = Down(64, 128)
        self.down3 = Down(128, 256)
        self.down4 = Down(256, 512)
        self.down5 = Down(512, 1024)
        self.up1 = Up(1024, 512)
        self.up2 = Up(512, 256)
        self.up3 = Up(256, 128)
        self.up4 = Up(128, 64)
        self.output = nn.Conv2d(64, dout, 1)

    def forward(self, x):
        x1 = self.down1(x)
        x2 = self.down2(x1)
        x3 = self.down3(x2)
        x4 = self.down4(x3)
        x5 = self.down5(x4)
        x = self.up1(x4, x5)
        x = self.up2(x3, x)
        x = self.up3(x2, x)
        x = self.up4(x1, x)
        return F.conv2d(x, self.output.weight, self.output.bias)
