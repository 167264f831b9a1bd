"""U-Net on the MI355X kernels -- drop-in for the reference's unet/unet.py (same constructor signatures,
child-module names and therefore the same state_dict keys/shapes; parameters stay fp32 in the reference
layouts).  The nn.Conv2d / nn.BatchNorm2d / nn.ConvTranspose2d children are PARAMETER CONTAINERS only (they
give identical default initialisation and state_dict layout); forward never calls them -- it drives the
fused HIP kernels through image_segmentation_amd.ops.  CUDA/HIP tensors only: there is no CPU path.

  DoubleConvReLU(din, dout)   reference unet/unet.py:4-25
  Down(din, dout)             reference unet/unet.py:28-45
  Up(din, dout).forward(x1, x2)   reference unet/unet.py:47-64   (skip x1 FIRST in the concat, :63)
  unet(din, dout)             reference unet/unet.py:67-105
"""
import torch
from torch import nn

from . import ops


class _FusedBase(nn.Module):
    """Holds the packed-weight cache and the optional per-module compute dtype override."""

    def __init__(self):
        super().__init__()
        self.cache = ops.PackCache()
        self.compute_dtype = None      # None -> ops.get_compute_dtype()


class DoubleConvReLU(_FusedBase):
    def __init__(self, din, dout, bias=True):
        super().__init__()
        self.doubleConvReLU = nn.Sequential(
            nn.Conv2d(din, dout, kernel_size=3, padding=1, bias=bias),
            nn.BatchNorm2d(dout),
            nn.ReLU(),
            nn.Conv2d(dout, dout, kernel_size=3, padding=1, bias=bias),
            nn.BatchNorm2d(dout),
            nn.ReLU(),
        )

    def bn_modules(self):
        return self.doubleConvReLU[1], self.doubleConvReLU[4]

    def forward(self, x, x_second=None):
        """x_second: optional second operand of a channel concat [x | x_second] (used by Up)."""
        s = self.doubleConvReLU
        y = ops.DoubleConvFn.apply(self, x, x_second, s[0].weight, s[0].bias, s[1].weight, s[1].bias,
                                   s[3].weight, s[3].bias, s[4].weight, s[4].bias)
        y._segk_bn2 = self.__dict__.pop("_bn2_vectors", None)   # for a pooling layer behind this block (Down)
        y._segk_pooled = self.__dict__.pop("_pooled_output", None)
        return y


class Down(nn.Module):
    def __init__(self, din, dout):
        super().__init__()
        self.maxpool_doubleConv = nn.Sequential(
            nn.MaxPool2d(kernel_size=2, stride=2),
            DoubleConvReLU(din, dout),
        )

    def forward(self, x, return_skip=False):
        """return_skip: also return an alias of x to be used as the skip connection, so that the pooling backward
        adds its routed gradient straight into the skip gradient (one kernel instead of pool-backward + add)."""
        dc = self.maxpool_doubleConv[1]
        dtype = dc.compute_dtype or ops.get_compute_dtype()
        if return_skip:
            p, skip = ops.MaxPoolSkipFn.apply(x, dtype, getattr(x, "_segk_bn2", None))
            return dc(p), skip
        return dc(ops.MaxPoolFn.apply(x, dtype))


class Up(_FusedBase):
    def __init__(self, din, dout):
        super().__init__()
        self.upsample = nn.ConvTranspose2d(din, dout, kernel_size=2, stride=2)
        self.doubleConv = DoubleConvReLU(din, dout)

    def forward(self, x1, x2):
        u = ops.ConvT2x2Fn.apply(self, x2, self.upsample.weight, self.upsample.bias)
        if u.shape[2:] != x1.shape[2:]:
            raise RuntimeError(f"Sizes of tensors must match except in dimension 1: {tuple(x1.shape)} vs "
                               f"{tuple(u.shape)} (H and W must be multiples of 16)")
        return self.doubleConv(x1, u)       # concat [x1 | u] consumed in place by the first conv


class unet(_FusedBase):
    def __init__(self, din, dout):
        super().__init__()
        self.scale = 1

        self.down1 = DoubleConvReLU(din, self.scale * 64)
        self.down2 = Down(self.scale * 64, self.scale * 128)
        self.down3 = Down(self.scale * 128, self.scale * 256)
        self.down4 = Down(self.scale * 256, self.scale * 512)
        self.down5 = Down(self.scale * 512, self.scale * 1024)

        self.up1 = Up(self.scale * 1024, self.scale * 512)
        self.up2 = Up(self.scale * 512, self.scale * 256)
        self.up3 = Up(self.scale * 256, self.scale * 128)
        self.up4 = Up(self.scale * 128, self.scale * 64)

        self.output = nn.Conv2d(self.scale * 64, dout, kernel_size=1)
        # the outputs of down1..down4 are pooled by the next Down block (forward below): those blocks emit the pooled
        # tensor in the same pass as their final BN+ReLU
        for blk in (self.down1, self.down2.maxpool_doubleConv[1], self.down3.maxpool_doubleConv[1],
                    self.down4.maxpool_doubleConv[1]):
            blk._emit_pool = True

    def set_compute_dtype(self, dtype):
        """Per-model override of ops.set_compute_dtype (torch.float32 parity mode / torch.bfloat16)."""
        for m in self.modules():
            if isinstance(m, _FusedBase):
                m.compute_dtype = dtype
        return self

    def forward(self, x):
        with ops.defer_batch_counters():     # one fused update of the 18 num_batches_tracked counters
            return self._forward(x)

    def _forward(self, x):
        x1 = self.down1(x)
        x2, s1 = self.down2(x1, return_skip=True)     # s1..s4 alias x1..x4 (the skip connections, unet.py:96-103)
        x3, s2 = self.down3(x2, return_skip=True)
        x4, s3 = self.down4(x3, return_skip=True)
        x5, s4 = self.down5(x4, return_skip=True)

        x = self.up1(s4, x5)
        x = self.up2(s3, x)
        x = self.up3(s2, x)
        x = self.up4(s1, x)

        return ops.HeadFn.apply(self, x, self.output.weight, self.output.bias)
