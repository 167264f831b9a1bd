"""CLIP-UNet on the MI355X kernels -- drop-in for the reference's clip/clipunet.py (same constructor
signatures, child names and state_dict keys: decoder.init_conv, decoder.decoder_blocks.K.{upsample,skip_conv,
conv_block.{0,1,3,4}}, output_layer, encoder.clip_vit.*).

  ClipViTEncoder(model_name, freeze_encoder, skip_indices)   reference clip/clipunet.py:7-65
  DecoderBlock(in_channels, in_channels_skip, out_channels)  reference clip/clipunet.py:68-105
  UNetDecoder(encoder_hidden_dim, decoder_channels)          reference clip/clipunet.py:108-144
  ClipUNet(num_classes, decoder_channels, freeze_encoder, model_name, skip_indices)   :147-188

The ViT-B/16 encoder is third-party code (transformers.CLIPVisionModel) in the reference; here the stock module
is kept as the parameter container (same `encoder.clip_vit.*` keys) and, frozen as in the reference, its forward
runs on the HIP kernels too (vit.py: MFMA GEMMs for the patch projection and every nn.Linear, csrc/vit.hip for
LayerNorm, attention and the token layout).  The decoder -- ConvTranspose up-sampling, 1x1 skip projections (MFMA
GEMMs), bilinear skip resize, concat-free bias-free DoubleConv blocks and the 1x1 head -- runs on the HIP kernels.  `model_name` may be a hub
id (needs network/cache, like the reference) or a local directory; `ClipViTEncoder.from_config` builds the same
architecture from a local CLIPVisionConfig with random weights (offline use, tests, benchmarks).
"""
import torch
import torch.nn as nn

from . import ops
from .unet import _FusedBase, DoubleConvReLU


class ClipViTEncoder(nn.Module):
    def __init__(self, model_name="openai/clip-vit-base-patch16", freeze_encoder=True, skip_indices=[3, 5, 7, 9],
                 _config=None):
        super().__init__()
        from transformers import CLIPVisionModel, CLIPVisionConfig
        self.skip_indices = sorted(skip_indices)
        self.compute_dtype = None      # None -> ops.get_compute_dtype()
        if _config is not None:
            self.config = _config
            self.clip_vit = CLIPVisionModel(_config)
        else:
            self.config = CLIPVisionConfig.from_pretrained(model_name)
            self.clip_vit = CLIPVisionModel.from_pretrained(model_name)
        if freeze_encoder:
            for param in self.clip_vit.parameters():
                param.requires_grad = False
        self.grid_size = self.config.image_size // self.config.patch_size
        self.hidden_dim = self.config.hidden_size

    @classmethod
    def from_config(cls, config=None, freeze_encoder=True, skip_indices=[3, 5, 7, 9]):
        from transformers import CLIPVisionConfig
        return cls(freeze_encoder=freeze_encoder, skip_indices=skip_indices,
                   _config=config or CLIPVisionConfig(patch_size=16))

    @staticmethod
    def remap_clip_keys(state_dict, own, nested):
        """In place: bring every `<own>...` key to the layout of the installed transformers -- `<own>vision_model.<rest>`
        (4.x: CLIPVisionModel wraps a CLIPVisionTransformer; the layout of checkpoints written by the reference,
        clip/clipunet.py:25-26 and prompt_based/segmentation_webapp/app.py:65-79) when `nested`, `<own><rest>` (5.x)
        otherwise.  `position_ids` buffers (persistent in old 4.x releases only) follow the same rule."""
        vm = "vision_model."
        for k in [k for k in state_dict if k.startswith(own)]:
            rest = k[len(own):]
            if nested and not rest.startswith(vm):
                state_dict[own + vm + rest] = state_dict.pop(k)
            elif not nested and rest.startswith(vm):
                state_dict[own + rest[len(vm):]] = state_dict.pop(k)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        # load_state_dict works on its own shallow copy of the caller's dict, and a module's hook runs before its
        # children are loaded: renaming here is what the `clip_vit` child then sees
        own = prefix + "clip_vit."
        self.remap_clip_keys(state_dict, own, hasattr(self.clip_vit, "vision_model"))
        pid = own + ("vision_model." if hasattr(self.clip_vit, "vision_model") else "") + "embeddings.position_ids"
        if pid in state_dict and pid[len(own):] not in self.clip_vit.state_dict():
            del state_dict[pid]                   # saved by a release that kept the index buffer persistent
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    def _grid(self, hidden_state, n):
        # drop CLS, [B,196,768] -> [B,768,14,14] (clipunet.py:48-51,54-63).  The permuted view of the
        # contiguous token tensor IS a channels-last [B,C,H,W] tensor: no copy is needed for the kernels.
        patches = hidden_state[:, 1:, :].reshape(n, self.grid_size, self.grid_size, self.hidden_dim).contiguous()
        return patches.permute(0, 3, 1, 2)

    def forward(self, x):
        if x.shape[2] != self.config.image_size or x.shape[3] != self.config.image_size:
            print(f"Input image size ({x.shape[2]}x{x.shape[3]}) doesn't match "
                  f"CLIP expected size ({self.config.image_size}x{self.config.image_size}). "
                  f"Behavior may be unexpected. Consider resizing input.")
        if not any(p.requires_grad for p in self.clip_vit.parameters()):
            # frozen feature extractor (the reference default, clipunet.py:28-30): forward-only HIP path
            from . import vit
            plan = self.__dict__.get("_plan")
            if plan is None:
                plan = vit.ClipVisionPlan()
                object.__setattr__(self, "_plan", plan)
            dtype = self.compute_dtype or ops.get_compute_dtype()
            return vit.forward_features(self.clip_vit, plan, x, self.skip_indices, dtype)
        # fine-tuning the encoder (freeze_encoder=False): third-party autograd of the stock module on the GPU
        outputs = self.clip_vit(pixel_values=x, output_hidden_states=True)
        n = x.shape[0]
        bottleneck = self._grid(outputs.last_hidden_state, n)
        skips = [self._grid(outputs.hidden_states[i], n) for i in self.skip_indices]
        return bottleneck, skips


class _Conv1x1(_FusedBase):
    """Parameter container + fused call for nn.Conv2d(cin, cout, kernel_size=1)."""

    def __init__(self, conv):
        super().__init__()
        self._conv = [conv]            # not registered twice: the owner registers `conv` under the reference name

    def __call__(self, x):
        c = self._conv[0]
        return ops.Conv1x1Fn.apply(self, x, c.weight, c.bias)


class DecoderBlock(_FusedBase):
    def __init__(self, in_channels, in_channels_skip, out_channels):
        super().__init__()
        self.upsample = nn.ConvTranspose2d(in_channels, in_channels // 2, kernel_size=2, stride=2)
        self.skip_conv = nn.Conv2d(in_channels_skip, in_channels // 2, kernel_size=1)
        dc = DoubleConvReLU(in_channels, out_channels, bias=False)
        self.conv_block = dc.doubleConvReLU          # registered under the reference's name: conv_block.{0,1,3,4}
        object.__setattr__(self, "_dc", dc)          # fused driver (shares the Sequential; not a registered child)
        object.__setattr__(self, "_skip", _Conv1x1(self.skip_conv))

    def train(self, mode=True):
        super().train(mode)
        self._dc.train(mode)
        return self

    def forward(self, x, skip, head=None):
        dtype = self.compute_dtype or ops.get_compute_dtype()
        self._dc.compute_dtype = self.compute_dtype
        self._skip.compute_dtype = self.compute_dtype
        x = ops.ConvT2x2Fn.apply(self, x, self.upsample.weight, self.upsample.bias)
        skip = self._skip(skip)
        if skip.shape[2:] != x.shape[2:]:
            skip = ops.BilinearFn.apply(skip, x.shape[2:], dtype)
        return self._dc(x, skip, head=head)            # concat [x | skip] (upsampled FIRST, clipunet.py:102)


class UNetDecoder(_FusedBase):
    def __init__(self, encoder_hidden_dim, decoder_channels):
        super().__init__()
        self.init_conv = nn.Conv2d(encoder_hidden_dim, decoder_channels[0], kernel_size=1)
        object.__setattr__(self, "_init", _Conv1x1(self.init_conv))
        self.decoder_blocks = nn.ModuleList()
        in_channels = decoder_channels[0]
        for i in range(len(decoder_channels) - 1):
            out_ch = decoder_channels[i + 1]
            self.decoder_blocks.append(DecoderBlock(in_channels, encoder_hidden_dim, out_ch))
            in_channels = out_ch

    def forward(self, x, skips, head=None):
        """head: an nn.Conv2d(C, classes, 1) run inside the last block's autograd node (returns its logits)."""
        self._init.compute_dtype = self.compute_dtype
        x = self._init(x)
        last = len(self.decoder_blocks) - 1
        for i, (block, skip) in enumerate(zip(self.decoder_blocks, reversed(skips))):
            x = block(x, skip, head=head if i == last else None)
        return x


class ClipUNet(_FusedBase):
    def __init__(self, num_classes=4, decoder_channels=[1024, 512, 256, 128, 64], freeze_encoder=True,
                 model_name="openai/clip-vit-base-patch16", skip_indices=[3, 5, 7, 9], encoder=None):
        super().__init__()
        self.encoder = encoder if encoder is not None else ClipViTEncoder(
            model_name=model_name, freeze_encoder=freeze_encoder, skip_indices=skip_indices)
        self.decoder = UNetDecoder(encoder_hidden_dim=self.encoder.hidden_dim, decoder_channels=decoder_channels)
        self.output_layer = nn.Conv2d(decoder_channels[-1], num_classes, kernel_size=1)

    def set_compute_dtype(self, dtype):
        for m in self.modules():
            if isinstance(m, (_FusedBase, ClipViTEncoder)):
                m.compute_dtype = dtype
        return self

    def forward(self, x):
        ops.repack_stale(self)                   # after an optimizer step: the decoder's weight pairs re-packed by one launch
        x, skips = self.encoder(x)
        with ops.defer_batch_counters():         # one fused update of the decoder's num_batches_tracked counters
            return self.decoder(x, skips, head=self.output_layer)
