"""Frozen CLIP vision transformer forward on the MI355X kernels (host driver).

The reference (clip/clipunet.py:25-46) calls transformers.CLIPVisionModel with output_hidden_states=True and
keeps hidden states `skip_indices` plus the last one.  Here the stock module is only the parameter container
(state_dict keys `encoder.clip_vit.*` unchanged); `forward_features` walks its parameters and drives
libsegk.so: MFMA GEMMs (`segk_linear`) for the patch projection and every nn.Linear, and the kernels of
csrc/vit.hip for embedding + pre-LayerNorm, residual add + LayerNorm, attention and the token->grid layout.
Forward only: the encoder is frozen (clipunet.py:28-30), so no autograd graph is built.

Token tensors are row matrices [Mp][C] with Mp = B*T rounded up to 16 rows (the GEMM's pixel-strip width);
rows past B*T are scratch and never read back.
"""
import torch

from . import _lib, ops
from .ops import _DT, _p, _stream, pad32


def _vision(clip_vit):
    """transformers 4.x nests the transformer under `.vision_model`; 5.x does not."""
    return getattr(clip_vit, "vision_model", clip_vit)


class ClipVisionPlan:
    """Packed weights of one CLIPVisionModel for one compute dtype (re-packed when a parameter's version moves)."""

    def __init__(self):
        self.cache = ops.PackCache()
        self._ws = {}

    def workspace(self, key, shapes, dtype, dev):
        """Row-matrix scratch buffers, allocated (zeroed) once per (batch geometry, dtype, device)."""
        k = (key, dtype, torch.device(dev))
        ws = self._ws.get(k)
        if ws is None:
            ws = [torch.zeros(sh, dtype=(torch.float32 if f32 else dtype), device=dev) for sh, f32 in shapes]
            self._ws = {k: ws}          # one geometry at a time: a new batch shape releases the old scratch
        return ws

    def linear(self, key, weight, dtype):
        def build():
            w = weight.detach().float().reshape(weight.shape[0], -1)
            n, k = w.shape
            if pad32(k) != k:
                w = torch.nn.functional.pad(w, (0, pad32(k) - k))
            return ops.pack_conv(w.reshape(n, -1, 1, 1).contiguous(), w.shape[1], 0, dtype, 0, taps=1)
        return self.cache.get((key, dtype), weight, build)

    def fused_qkv(self, key, attn, dtype):
        # one GEMM for the three projections: rows [q | k | v]
        def build():
            w = torch.cat([attn.q_proj.weight, attn.k_proj.weight, attn.v_proj.weight], 0).detach().float()
            b = torch.cat([attn.q_proj.bias, attn.k_proj.bias, attn.v_proj.bias], 0).detach().float().contiguous()
            return ops.pack_conv(w.reshape(w.shape[0], -1, 1, 1).contiguous(), w.shape[1], 0, dtype, 0, taps=1), b
        # the fused copy depends on six tensors: the cache entry follows q_proj.weight, the other five are part of the key
        others = (attn.k_proj.weight, attn.v_proj.weight, attn.q_proj.bias, attn.k_proj.bias, attn.v_proj.bias)
        sig = tuple((t._version, t.data_ptr()) for t in others)
        full = (key, dtype, sig)
        if full not in self.cache._c:     # drop the copy made for an older state of the five (at most one per layer lives)
            for k in [k for k in self.cache._c if k[:2] == (key, dtype)]:
                del self.cache._c[k]
        return self.cache.get(full, attn.q_proj.weight, build)


def _f32(p):
    p = p.detach()
    return p if (p.dtype == torch.float32 and p.is_contiguous()) else p.float().contiguous()


def forward_features(clip_vit, plan, x, skip_indices, dtype):
    """x [B,3,H,W] fp32 CUDA -> (last hidden state grid, [hidden state grids at skip_indices]) as act tensors
    [B,D,G,G] (NHWC storage, compute dtype), CLS dropped (clipunet.py:48-63)."""
    ops._require_cuda(x, "ClipViTEncoder")
    vm = _vision(clip_vit)
    cfg = clip_vit.config
    emb = vm.embeddings
    D, heads, L = cfg.hidden_size, cfg.num_attention_heads, cfg.num_hidden_layers
    ps, I, eps = cfg.patch_size, cfg.intermediate_size, float(cfg.layer_norm_eps)
    if cfg.hidden_act != "quick_gelu":
        raise RuntimeError(f"ClipViTEncoder: activation {cfg.hidden_act!r} is not implemented on the HIP path (quick_gelu only)")
    if D % 32 or I % 32 or D % heads:
        raise RuntimeError("ClipViTEncoder: hidden and intermediate sizes must be multiples of 32")
    hd = D // heads
    B, C, H, W = x.shape
    if H % ps or W % ps or H != W:
        raise RuntimeError(f"ClipViTEncoder: {H}x{W} input is not a square multiple of the patch size {ps}")
    G = H // ps
    N, T = G * G, G * G + 1
    if emb.position_embedding.weight.shape[0] != T:
        raise RuntimeError(f"ClipViTEncoder: {T} tokens but {emb.position_embedding.weight.shape[0]} position embeddings "
                           "(the reference does not interpolate them either)")
    dev, dt, s = x.device, _DT[dtype], _stream()
    xin = x.detach()
    if xin.dtype != torch.float32 or not xin.is_contiguous():
        xin = xin.float().contiguous()
    M = B * T
    Mp, Np = (M + 15) // 16 * 16, (B * N + 15) // 16 * 16
    Kp = pad32(C * ps * ps)

    # GEMMs with N = D are 78 output tiles at B = 16 (256 CUs): split their K three ways when it divides into whole
    # 64-element stages; the residual add sums the partial products (bf16 path only)
    def ksplit(K):
        return 3 if (dtype == torch.bfloat16 and Mp >= 128 and D % 128 == 0 and K % 192 == 0 and K // 192 >= 4) else 1
    S_out, S_fc2 = ksplit(D), ksplit(I)
    patches, proj, h, a, qkv, ctx, o, f = plan.workspace(
        (B, C, H, W), [((Np, Kp), 0), ((Np, D), 0), ((Mp, D), 1), ((Mp, D), 0), ((Mp, 3 * D), 0), ((Mp, D), 0),
                       ((max(S_out, S_fc2) * Mp, D), 0), ((Mp, I), 0)], dtype, dev)

    def linear_to_o(src, w, bias, K, S):
        if S > 1:
            _lib.call("segk_linear_splitk", src.data_ptr(), w.data_ptr(), bias.data_ptr(), o.data_ptr(), Mp, K, D, S, dt, s)
        else:
            _lib.call("segk_linear", src.data_ptr(), w.data_ptr(), bias.data_ptr(), o.data_ptr(), Mp, K, D, 0, dt, s)

    def add_ln(S, gamma, beta):
        _lib.call("segk_add_layernorm_parts", h.data_ptr(), o.data_ptr(), S, Mp * D, _p(gamma), _p(beta), eps,
                  a.data_ptr() if gamma is not None else 0, M, D, D, dt, s)
    # patch embedding: im2col + GEMM (Conv2d(3, D, ps, stride ps, bias=False))
    _lib.call("segk_vit_patchify", xin.data_ptr(), patches.data_ptr(), B, C, H, W, ps, Kp, dt, s)
    wp = plan.linear("patch", emb.patch_embedding.weight, dtype)
    with ops._span("vit_gemm", 2.0 * B * N * Kp * D, 0.0):
        _lib.call("segk_linear", patches.data_ptr(), wp.data_ptr(), 0, proj.data_ptr(), Np, Kp, D, 0, dt, s)
    _lib.call("segk_vit_embed_ln", proj.data_ptr(), _f32(emb.class_embedding).data_ptr(),
              _f32(emb.position_embedding.weight).data_ptr(), _f32(vm.pre_layrnorm.weight).data_ptr(),
              _f32(vm.pre_layrnorm.bias).data_ptr(), eps, h.data_ptr(), B, T, D, D, dt, s)

    def grid():
        out = torch.empty((B, G, G, D), dtype=dtype, device=dev)
        _lib.call("segk_vit_tokens_to_grid", h.data_ptr(), out.data_ptr(), B, T, D, D, dt, s)
        return ops.act_view(out, D)

    want = set(skip_indices)
    states = {0: grid()} if 0 in want else {}
    layers = vm.encoder.layers
    es = 2 if dtype == torch.bfloat16 else 4
    _lib.call("segk_add_layernorm", h.data_ptr(), 0, _f32(layers[0].layer_norm1.weight).data_ptr(),
              _f32(layers[0].layer_norm1.bias).data_ptr(), eps, a.data_ptr(), M, D, D, dt, s)
    for li, layer in enumerate(layers):
        at = layer.self_attn
        wqkv, bqkv = plan.fused_qkv(("qkv", li), at, dtype)
        with ops._span("vit_gemm", 2.0 * M * D * 3 * D, 0.0):
            _lib.call("segk_linear", a.data_ptr(), wqkv.data_ptr(), bqkv.data_ptr(), qkv.data_ptr(), Mp, D, 3 * D, 0, dt, s)
        with ops._span("vit_attention", 4.0 * B * heads * T * T * hd, M * 4.0 * D * es):
            _lib.call("segk_attention", qkv.data_ptr(), ctx.data_ptr(), B, T, heads, hd, 3 * D, D, float(hd) ** -0.5, dt, s)
        with ops._span("vit_gemm", 2.0 * M * D * D, 0.0):
            linear_to_o(ctx, plan.linear(("out", li), at.out_proj.weight, dtype), _f32(at.out_proj.bias), D, S_out)
        add_ln(S_out, _f32(layer.layer_norm2.weight), _f32(layer.layer_norm2.bias))
        with ops._span("vit_gemm", 2.0 * M * D * I, 0.0):
            _lib.call("segk_linear", a.data_ptr(), plan.linear(("fc1", li), layer.mlp.fc1.weight, dtype).data_ptr(),
                      _f32(layer.mlp.fc1.bias).data_ptr(), f.data_ptr(), Mp, D, I, 1, dt, s)
        with ops._span("vit_gemm", 2.0 * M * D * I, 0.0):
            linear_to_o(f, plan.linear(("fc2", li), layer.mlp.fc2.weight, dtype), _f32(layer.mlp.fc2.bias), I, S_fc2)
        if li + 1 < L:      # residual add fused with the next layer's layer_norm1
            nxt = layers[li + 1]
            add_ln(S_fc2, _f32(nxt.layer_norm1.weight), _f32(nxt.layer_norm1.bias))
        else:               # last layer: add only (last_hidden_state is taken before post_layernorm)
            add_ln(S_fc2, None, None)
        if li + 1 in want:
            states[li + 1] = grid()
    last = states[L] if L in states else grid()
    return last, [states[i] for i in sorted(want)]
