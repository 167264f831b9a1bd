"""Autoencoder family on the MI355X kernels -- drop-in for the reference's autoencoder/autoencoder.py (same class
names, constructor signatures, child-module names and therefore state_dict keys; parameters stay fp32 in the
reference layouts).  Like unet.py, the nn.Conv2d / nn.BatchNorm2d / nn.ConvTranspose2d children only hold
parameters; forward drives the fused HIP kernels through image_segmentation_amd.ops.  CUDA/HIP tensors only.

  EncoderBlock(din, dout)                         reference autoencoder.py:6-33    -> (pooled, skip)
  Encoder(din, base_channels)                     :35-54   -> (bottleneck, skip3, skip2, skip1)
  DecoderBlockWithSkips(din_up, din_skip, dout)   :57-93   cat([up, skip]) consumed in place by the first conv
  DecoderWithSkips(base_channels)                 :96-114
  DecoderBlockNoSkips(din_up, dout)               :117-146
  DecoderNoSkips(base_channels)                   :149-168
  ReconstructionAutoencoder(din, dout, base)      :171-200 (3x3 conv + bias head, Sigmoid; fp32 NCHW output)
  SegmentationEncoder / SegmentationAutoencoder   :203-305 (optional checkpoint loading and encoder freezing)
"""
import torch
from torch import nn

from . import ops
from .unet import _FusedBase


class _DoubleConvBlock(_FusedBase):
    """Two bias-free Conv3x3 + BatchNorm + ReLU driven as one fused DoubleConv; subclasses name the children."""

    def _convs(self):
        raise NotImplementedError

    def bn_modules(self):
        c = self._convs()
        return c[1], c[3]

    def _double(self, xa, xb=None, emit_pool=False, head=None):
        c1, b1, c2, b2 = self._convs()
        params = (c1.weight, c1.bias, b1.weight, b1.bias, c2.weight, c2.bias, b2.weight, b2.bias)
        return ops.double_conv(self, xa, xb, params, emit_pool=emit_pool, head=head)


class EncoderBlock(_DoubleConvBlock):
    def __init__(self, din, dout):
        super().__init__()
        self.conv1 = nn.Conv2d(din, dout, kernel_size=3, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(dout)
        self.relu1 = nn.ReLU()
        self.conv2 = nn.Conv2d(dout, dout, kernel_size=3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(dout)
        self.relu2 = nn.ReLU(inplace=True)
        self.pool = nn.MaxPool2d(kernel_size=2, stride=2)

    def _convs(self):
        return self.conv1, self.bn1, self.conv2, self.bn2

    def forward(self, x):
        skip, pooled = self._double(x, emit_pool=True)     # BN+ReLU and the pooling in one pass, one autograd node
        return pooled, skip


class Encoder(nn.Module):
    def __init__(self, din, base_channels):
        super().__init__()
        self.encoderPart1 = EncoderBlock(din, base_channels)
        self.encoderPart2 = EncoderBlock(base_channels, base_channels * 2)
        self.encoderPart3 = EncoderBlock(base_channels * 2, base_channels * 4)

    def forward(self, x):
        x1_pooled, skip1 = self.encoderPart1(x)
        x2_pooled, skip2 = self.encoderPart2(x1_pooled)
        bottleneck, skip3 = self.encoderPart3(x2_pooled)
        return bottleneck, skip3, skip2, skip1


def _conv_seq(cin, dout):
    return nn.Sequential(
        nn.Conv2d(cin, dout, kernel_size=3, padding=1, bias=False), nn.BatchNorm2d(dout), nn.ReLU(inplace=True),
        nn.Conv2d(dout, dout, kernel_size=3, padding=1, bias=False), nn.BatchNorm2d(dout), nn.ReLU(inplace=True))


class DecoderBlockWithSkips(_DoubleConvBlock):
    def __init__(self, din_up, din_skip, dout):
        super().__init__()
        self.up = nn.ConvTranspose2d(din_up, dout, kernel_size=2, stride=2)
        self.convs = _conv_seq(dout + din_skip, dout)

    def _convs(self):
        s = self.convs
        return s[0], s[1], s[3], s[4]

    def forward(self, x, skip_features, head=None):
        x_upsampled = ops.ConvT2x2Fn.apply(self, x, self.up.weight, self.up.bias)
        if skip_features.shape[2:] != x_upsampled.shape[2:]:
            diffY = skip_features.size()[2] - x_upsampled.size()[2]
            diffX = skip_features.size()[3] - x_upsampled.size()[3]
            if diffY < 0 or diffX < 0:
                raise ValueError("Upsampled larger than skip")
            skip_features = skip_features[:, :, diffY // 2: diffY // 2 + x_upsampled.size()[2],
                                          diffX // 2: diffX // 2 + x_upsampled.size()[3]]
        return self._double(x_upsampled, skip_features, head=head)     # concat [up | skip] never materialised


class DecoderWithSkips(nn.Module):
    def __init__(self, base_channels):
        super().__init__()
        self.decoderBlock1 = DecoderBlockWithSkips(din_up=base_channels * 4, din_skip=base_channels * 4, dout=base_channels * 2)
        self.decoderBlock2 = DecoderBlockWithSkips(din_up=base_channels * 2, din_skip=base_channels * 2, dout=base_channels)
        self.decoderBlock3 = DecoderBlockWithSkips(din_up=base_channels, din_skip=base_channels, dout=base_channels)

    def forward(self, bottleneck, skip3, skip2, skip1, head=None):
        """head: an nn.Conv2d(C, classes, 1) run inside the last block's autograd node (returns its logits)."""
        d1 = self.decoderBlock1(bottleneck, skip3)
        d2 = self.decoderBlock2(d1, skip2)
        return self.decoderBlock3(d2, skip1, head=head)


class DecoderBlockNoSkips(_DoubleConvBlock):
    def __init__(self, din_up, dout):
        super().__init__()
        self.up = nn.ConvTranspose2d(din_up, dout, kernel_size=2, stride=2)
        self.convs = _conv_seq(dout, dout)

    def _convs(self):
        s = self.convs
        return s[0], s[1], s[3], s[4]

    def forward(self, x):
        return self._double(ops.ConvT2x2Fn.apply(self, x, self.up.weight, self.up.bias))


class DecoderNoSkips(nn.Module):
    def __init__(self, base_channels):
        super().__init__()
        self.decoderBlock1 = DecoderBlockNoSkips(din_up=base_channels * 4, dout=base_channels * 2)
        self.decoderBlock2 = DecoderBlockNoSkips(din_up=base_channels * 2, dout=base_channels)
        self.decoderBlock3 = DecoderBlockNoSkips(din_up=base_channels, dout=base_channels)

    def forward(self, bottleneck):
        return self.decoderBlock3(self.decoderBlock2(self.decoderBlock1(bottleneck)))


def _set_dtype(root, dtype):
    for m in root.modules():
        if isinstance(m, _FusedBase):
            m.compute_dtype = dtype
    return root


class ReconstructionAutoencoder(_FusedBase):
    def __init__(self, din, dout=3, base_channels=64):
        super().__init__()
        self.encoder = Encoder(din, base_channels)
        self.decoder = DecoderNoSkips(base_channels)
        self.decoderOut = nn.Sequential(
            nn.Conv2d(base_channels, dout, kernel_size=3, padding=1),
            nn.Sigmoid()
        )

    def set_compute_dtype(self, dtype):
        return _set_dtype(self, dtype)

    def forward(self, x):
        ops.repack_stale(self)
        with ops.defer_batch_counters():
            bottleneck, _s3, _s2, _s1 = self.encoder(x)
            decoded = self.decoder(bottleneck)
            z = ops.Conv3x3Fn.apply(self, decoded, self.decoderOut[0].weight, self.decoderOut[0].bias)
        # the 3-channel head output leaves the NHWC act layout through stock (differentiable) torch ops
        return torch.sigmoid(z.float()).contiguous()


class SegmentationEncoder(nn.Module):
    def __init__(self, din, base_channels, pretrained_encoder_path=None, freeze_encoder=True):
        super().__init__()
        self.encoder = Encoder(din, base_channels)

        if pretrained_encoder_path:
            try:
                full_state_dict = torch.load(pretrained_encoder_path, weights_only=False,
                                             map_location=lambda storage, loc: storage)
                if "model_state_dict" in full_state_dict:
                    model_state_dict = full_state_dict["model_state_dict"]
                elif "state_dict" in full_state_dict:
                    model_state_dict = full_state_dict["state_dict"]
                else:
                    model_state_dict = full_state_dict
                encoder_state_dict = {}
                has_encoder_prefix = any(k.startswith('encoder.') for k in model_state_dict.keys())
                for key, value in model_state_dict.items():
                    if has_encoder_prefix and key.startswith('encoder.'):
                        encoder_state_dict[key[len('encoder.'):]] = value
                if not encoder_state_dict:
                    print("Warning: Could not extract encoder state dict. Checkpoint might be empty or incompatible.")
                else:
                    load_result = self.encoder.load_state_dict(encoder_state_dict, strict=True)
                    print(f"Loaded encoder weights. Load result:")
                    if load_result.missing_keys:
                        print("  Missing keys:", load_result.missing_keys)
                    if load_result.unexpected_keys:
                        print("  Unexpected keys:", load_result.unexpected_keys)
                    if not load_result.missing_keys and not load_result.unexpected_keys:
                        print("  All keys matched successfully.")
            except FileNotFoundError:
                print(f"Warning: Pre-trained encoder file not found: {pretrained_encoder_path}. Using random weights.")
            except Exception as e:
                print(f"Warning: Error loading weights: {e}. Check compatibility. Using random weights.")

        if freeze_encoder:
            if not pretrained_encoder_path:
                print("Warning: Freezing encoder, but no pre-trained weights were loaded.")
            for param in self.encoder.parameters():
                param.requires_grad = False
            print("Encoder parameters frozen.")
        else:
            print("Encoder parameters are trainable.")

    def forward(self, x):
        return self.encoder(x)


class SegmentationAutoencoder(_FusedBase):
    def __init__(self, din, base_channels=64, num_classes=4, pretrained_encoder_path=None, freeze_encoder=True):
        super().__init__()
        self.num_classes = num_classes
        self.encoder = SegmentationEncoder(din, base_channels, pretrained_encoder_path=pretrained_encoder_path,
                                           freeze_encoder=freeze_encoder)
        self.decoder = DecoderWithSkips(base_channels)
        self.finalConv = nn.Conv2d(base_channels, num_classes, kernel_size=1)

    def set_compute_dtype(self, dtype):
        return _set_dtype(self, dtype)

    def forward(self, x):
        ops.repack_stale(self)
        with ops.defer_batch_counters():
            bottleneck, skip3, skip2, skip1 = self.encoder(x)
            return self.decoder(bottleneck, skip3, skip2, skip1, head=self.finalConv)
