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

    def forward(self, x, x_second=None, emit_pool=False, head=None, upsample_from=None):
        """The reference call is forward(x).  Extensions used by the models of this package, all explicit:
        x_second: second operand of a channel concat [x | x_second];
        upsample_from=(module, x2): the second concat operand is module.upsample(x2) (ConvTranspose2d), computed inside
        this block's autograd node (Up);
        emit_pool: also return MaxPool2d(2,2) of the output -> (y, pooled) (the next Down block's pooling);
        head: an nn.Conv2d(C, classes, 1) applied to the output -> returns the fp32 logits instead of y."""
        s = self.doubleConvReLU
        params = (s[0].weight, s[0].bias, s[1].weight, s[1].bias, s[3].weight, s[3].bias, s[4].weight, s[4].bias)
        up_mod, up_x = upsample_from if upsample_from is not None else (None, None)
        return ops.double_conv(self, x, x_second, params, emit_pool=emit_pool, head=head, up_mod=up_mod, up_x=up_x)


class Down(nn.Module):
    def __init__(self, din, dout):
        super().__init__()
        self.maxpool_doubleConv = nn.Sequential(
            nn.MaxPool2d(kernel_size=2, stride=2),
            DoubleConvReLU(din, dout),
        )

    def forward(self, x, pooled=None, emit_pool=False):
        """The reference call is forward(x).  pooled: MaxPool2d(2,2)(x) when the block that produced x already emitted it
        (DoubleConvReLU.forward(emit_pool=True): its backward then also routes the pooled gradient into the skip
        gradient and accumulates its BatchNorm reductions in one kernel); emit_pool: see DoubleConvReLU.forward."""
        dc = self.maxpool_doubleConv[1]
        if pooled is None:
            pooled = ops.MaxPoolFn.apply(x, dc.compute_dtype or ops.get_compute_dtype())
        return dc(pooled, emit_pool=emit_pool)


class Up(_FusedBase):
    def __init__(self, din, dout):
        super().__init__()
        self.upsample = nn.ConvTranspose2d(din, dout, kernel_size=2, stride=2)
        self.doubleConv = DoubleConvReLU(din, dout)

    def forward(self, x1, x2, head=None):
        """cat([x1, upsample(x2)]) -> DoubleConv (unet.py:62-64): ConvTranspose2d and DoubleConv run in ONE autograd node
        (the concat is consumed in place by the first conv; the concat gradient's channel sums are the ConvTranspose
        bias gradient).  head: see DoubleConvReLU.forward."""
        return self.doubleConv(x1, head=head, upsample_from=(self, x2))


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

    def set_compute_dtype(self, dtype):
        """Per-model override of ops.set_compute_dtype (torch.float32 parity mode / torch.bfloat16)."""
        for m in self.modules():
            if isinstance(m, _FusedBase):
                m.compute_dtype = dtype
        return self

    def forward(self, x):
        ops.repack_stale(self)               # after an optimizer step: all 18 weight pairs re-packed by one launch
        with ops.defer_batch_counters():     # one fused update of the 18 num_batches_tracked counters
            return self._forward(x)

    def _forward(self, x):
        # the outputs of down1..down4 are pooled by the next Down block (unet.py:40) AND used as skip connections
        # (:96-103): those blocks emit the pooled tensor in the same pass as their final BN+ReLU
        x1, p1 = self.down1(x, emit_pool=True)
        x2, p2 = self.down2(x1, pooled=p1, emit_pool=True)
        x3, p3 = self.down3(x2, pooled=p2, emit_pool=True)
        x4, p4 = self.down4(x3, pooled=p3, emit_pool=True)
        x5 = self.down5(x4, pooled=p4)

        x = self.up1(x4, x5)
        x = self.up2(x3, x)
        x = self.up3(x2, x)
        return self.up4(x1, x, head=self.output)      # output 1x1 conv inside up4's autograd node (unet.py:105)
