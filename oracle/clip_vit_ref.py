"""Oracle (TEST INFRASTRUCTURE): CPU restatement of the CLIP vision transformer forward that the reference's
ClipViTEncoder runs (reference clip/clipunet.py:25-26 builds transformers.CLIPVisionModel, :41-46 calls it with
output_hidden_states=True, :48-63 drops CLS and reshapes hidden states 3,5,7,9 and the last one to grids).

The arithmetic lives in a third-party dependency that is not vendored under /root/reference and that the
reference does not pin (no requirements file): `transformers` (5.15.0 in this image), module
transformers/models/clip/modeling_clip.py.  Restated from its published algorithm in plain torch ops:
  CLIPVisionEmbeddings  patch Conv2d(3, D, ps, stride ps, bias=False) -> tokens, prepend class_embedding,
                        add position_embedding
  CLIPVisionTransformer pre_layrnorm, encoder, (post_layernorm only feeds pooler_output: unused by the reference)
  CLIPEncoderLayer      h = h + attn(LN1(h));  h = h + fc2(quick_gelu(fc1(LN2(h))))
  CLIPAttention         softmax(q k^T * head_dim**-0.5) v with biased q/k/v/out projections
  quick_gelu            x * sigmoid(1.702 x)
Pinned by tests/golden/clip_vit_small.npz: tools/gen_golden.py drives the real CLIPVisionModel (local random
config, portable-fill weights) exactly as clipunet.py:41-63 does.  Pretrained-weight parity stays UNPINNED (no
checkpoint is reachable offline).
"""
import torch
import torch.nn.functional as F


def _vision(clip_vit):
    return getattr(clip_vit, "vision_model", clip_vit)


@torch.no_grad()
def hidden_states(clip_vit, x):
    """All L+1 hidden states [B,T,D] (index 0 = pre-LayerNorm output) from the parameters of a CLIPVisionModel."""
    vm, cfg = _vision(clip_vit), clip_vit.config
    emb = vm.embeddings
    D, heads, eps = cfg.hidden_size, cfg.num_attention_heads, cfg.layer_norm_eps
    hd = D // heads
    B = x.shape[0]
    tok = F.conv2d(x, emb.patch_embedding.weight, None, stride=cfg.patch_size).flatten(2).transpose(1, 2)
    e = torch.cat([emb.class_embedding.expand(B, 1, -1), tok], 1) + emb.position_embedding.weight[None]
    h = F.layer_norm(e, (D,), vm.pre_layrnorm.weight, vm.pre_layrnorm.bias, eps)
    out = [h]
    for layer in vm.encoder.layers:
        at = layer.self_attn
        a = F.layer_norm(h, (D,), layer.layer_norm1.weight, layer.layer_norm1.bias, eps)
        T = a.shape[1]
        q = F.linear(a, at.q_proj.weight, at.q_proj.bias).view(B, T, heads, hd).transpose(1, 2)
        k = F.linear(a, at.k_proj.weight, at.k_proj.bias).view(B, T, heads, hd).transpose(1, 2)
        v = F.linear(a, at.v_proj.weight, at.v_proj.bias).view(B, T, heads, hd).transpose(1, 2)
        p = torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, dim=-1)
        c = (p @ v).transpose(1, 2).reshape(B, T, D)
        h = h + F.linear(c, at.out_proj.weight, at.out_proj.bias)
        b = F.layer_norm(h, (D,), layer.layer_norm2.weight, layer.layer_norm2.bias, eps)
        f = F.linear(b, layer.mlp.fc1.weight, layer.mlp.fc1.bias)
        f = f * torch.sigmoid(1.702 * f)
        h = h + F.linear(f, layer.mlp.fc2.weight, layer.mlp.fc2.bias)
        out.append(h)
    return out


def encoder_features(clip_vit, x, skip_indices):
    """(bottleneck, skips) as clipunet.py:41-65 returns them: [B,D,G,G] grids without the CLS token."""
    from .clipunet_ref import tokens_to_grid
    hs = hidden_states(clip_vit, x)
    g = clip_vit.config.image_size // clip_vit.config.patch_size
    return tokens_to_grid(hs[-1], g), [tokens_to_grid(hs[i], g) for i in sorted(skip_indices)]
