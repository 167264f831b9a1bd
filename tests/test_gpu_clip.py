"""GPU (-m gpu): CLIP-UNet decoder path (reference clip/clipunet.py:68-188) on the HIP kernels against the
reference goldens and the CPU oracle: bilinear skip resize, DecoderBlock, UNetDecoder + head, ClipUNet."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle.fill import fill, labels, fill_module
from oracle import clipunet_ref, losses_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def seg():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import image_segmentation_amd as s
    return s


def cpu(t):
    return t.detach().float().cpu().numpy()


def close(a, b, rtol, atol, msg=""):
    np.testing.assert_allclose(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), rtol=rtol, atol=atol,
                               err_msg=msg)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [(2, 32, 14, 14, 28, 28), (1, 64, 14, 14, 224, 224), (2, 40, 3, 5, 6, 10), (1, 32, 7, 7, 10, 13),
                                  (2, 32, 14, 14, 56, 56), (1, 32, 5, 9, 23, 40), (1, 96, 14, 14, 112, 112)])
def test_bilinear(seg, dtype, case):
    from image_segmentation_amd import ops
    B, C, IH, IW, OH, OW = case
    x0 = fill((B, C, IH, IW), 1, -1, 1).to(dtype).float()
    x = x0.clone().requires_grad_(True)
    y = F.interpolate(x, size=(OH, OW), mode="bilinear", align_corners=False)
    g0 = fill((B, C, OH, OW), 2, -1, 1).to(dtype).float()
    y.backward(g0)
    xa = ops.to_act(x0.cuda(), dtype).requires_grad_(True)
    ya = ops.BilinearFn.apply(xa, (OH, OW), dtype)
    ya.backward(ops.to_act(g0.cuda(), dtype))
    t = 2e-6 if dtype == torch.float32 else 1e-2
    assert np.abs(cpu(ya) - y.detach().numpy()).max() < t
    scale = max(1.0, (OH / IH) * (OW / IW))
    assert np.abs(cpu(xa.grad) - x.grad.numpy()).max() < t * scale


def check_grads(model, g, rtol, atol):
    for n, p in model.named_parameters():
        ref = g["grad." + n]
        close(cpu(p.grad), ref, rtol, atol * max(1.0, np.abs(ref).max()), n)


@pytest.mark.parametrize("tag,cin,cskip,cout,b,h,base", [("decoderblock_16_12_8", 16, 12, 8, 1, 3, 7000),
                                                       ("decoderblock_64_96_32", 64, 96, 32, 2, 7, 7100)])
def test_decoder_block_golden_fp32(seg, golden, tag, cin, cskip, cout, b, h, base):
    g = golden(tag)
    seg.set_compute_dtype(torch.float32)
    m = seg.DecoderBlock(cin, cskip, cout); fill_module(m, base); m.cuda().train()
    x = fill((b, cin, h, h), 31, -1, 1).cuda().requires_grad_(True)
    sk = fill((b, cskip, h, h), 32, -1, 1).cuda().requires_grad_(True)
    y = m(x, sk)
    (y.float() * fill(tuple(y.shape), 5, -1, 1).cuda()).sum().backward()
    close(cpu(y), g["y"], 1e-4, 2e-5)
    close(cpu(x.grad), g["dx"], 1e-3, 5e-5); close(cpu(sk.grad), g["dskip"], 1e-3, 5e-5)
    check_grads(m, g, 1e-3, 1e-4)
    for n, buf in m.named_buffers():
        close(cpu(buf), g["buf." + n], 1e-5, 1e-6, n)


def test_clip_decoder_golden_fp32(seg, golden):
    g = golden("clip_decoder_b2")
    seg.set_compute_dtype(torch.float32)
    dec = seg.UNetDecoder(768, [1024, 512, 256, 128, 64]); head = torch.nn.Conv2d(64, 4, 1)
    both = torch.nn.ModuleDict({"decoder": dec, "output_layer": head}); fill_module(both, 5000); both.cuda().train()
    x = fill((2, 768, 14, 14), 11, -1, 1).cuda()
    skips = [fill((2, 768, 14, 14), 20 + i, -1, 1).cuda() for i in range(4)]
    from image_segmentation_amd import ops
    d = dec(x, skips)
    lg = ops.HeadFn.apply(dec, d, head.weight, head.bias)
    assert abs(d.double().sum().item() - float(g["dec_sum"])) < 20.0
    close(cpu(d[:, :, ::16, ::16]), g["dec_sample"], 1e-3, 1e-3)
    close(cpu(lg[:, :, ::8, ::8]), g["logits_sample"], 1e-3, 1e-3)
    assert (lg.argmax(1)[:, ::4, ::4].cpu().numpy().astype(np.uint8) == g["argmax_sample"]).all()
    Y = labels((2, 224, 224), 3, 4).cuda()
    ce = seg.CrossEntropyLoss()(lg, Y)
    assert abs(ce.item() - float(g["ce"])) < 2e-5
    ce.backward()
    norms = {n: p.grad.double().norm().item() for n, p in both.named_parameters()}
    for n, ref in zip(g["grad_names"], g["grad_norms"]):
        assert abs(norms[str(n)] - ref) <= 3e-3 * ref + 1e-7, (n, norms[str(n)], ref)


def test_clipunet_end_to_end_vs_oracle(seg):
    """ClipUNet with a LOCAL random-weight ViT-B/16 config (no hub access): the stock encoder runs on the
    GPU, the decoder on the HIP kernels; compared with the CPU oracle decoder fed the same encoder features."""
    pytest.importorskip("transformers")
    seg.set_compute_dtype(torch.float32)
    torch.manual_seed(0)
    enc = seg.ClipViTEncoder.from_config()
    m = seg.ClipUNet(num_classes=4, encoder=enc)
    fill_module(m.decoder, 5000); fill_module(m.output_layer, 6000)
    m.cuda().train()
    X = fill((2, 3, 224, 224), 9, -1, 1).cuda()
    lg = m(X)
    assert tuple(lg.shape) == (2, 4, 224, 224) and lg.dtype == torch.float32
    assert not any(p.requires_grad for p in m.encoder.parameters())
    with torch.no_grad():
        bott, skips = m.encoder(X)
    ref = clipunet_ref.UNetDecoder(768, [1024, 512, 256, 128, 64]); fill_module(ref, 5000); ref.train()
    rh = torch.nn.Conv2d(64, 4, 1); fill_module(rh, 6000)
    lr = rh(ref(bott.float().cpu().contiguous(), [s.float().cpu().contiguous() for s in skips]))
    assert np.abs(cpu(lg) - lr.detach().numpy()).max() < 2e-3
    Y = labels((2, 224, 224), 3, 4)
    loss = seg.WeightedDiceCELoss(ignore_index=3)(lg, Y.cuda()); loss.backward()
    lref = losses_ref.dice_ce(lr, Y, ignore_index=3); lref.backward()
    assert abs(loss.item() - lref.item()) < 1e-4
    gn = m.decoder.init_conv.weight.grad.norm().item(); gr = ref.init_conv.weight.grad.norm().item()
    assert abs(gn - gr) < 5e-3 * gr
    keys = list(m.state_dict().keys())
    assert "decoder.decoder_blocks.0.conv_block.0.weight" in keys and "output_layer.bias" in keys
    assert any(k.startswith("encoder.clip_vit.") for k in keys)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_clip_decoder_one_launch_repack_after_optimizer_step(seg, dtype):
    """ops.repack_stale on the CLIP decoder: its fused drivers (`_dc`, `_skip`, `_init`) live outside the module tree, so the
    cache walk has to find them; after an optimizer step every registered copy -- 8 3x3 weight pairs, 4 ConvTranspose pairs and
    bias operands, and the 5 1x1 convs' forward weights (kind 3) and padded biases (kind 2, one repeat) -- equals a fresh
    per-tensor pack of the stepped parameter bit for bit, in the buffers that were there before."""
    from image_segmentation_amd import ops
    seg.set_compute_dtype(dtype)
    dec = seg.UNetDecoder(64, [64, 32, 32, 32, 32]); head = torch.nn.Conv2d(32, 4, 1)

    class Wrap(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.decoder, self.output_layer = dec, head

        def forward(self, x, skips):
            ops.repack_stale(self)
            return self.decoder(x, skips, head=self.output_layer)
    m = Wrap(); fill_module(m, 4100); m.cuda().train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-2, fused=True)
    x = fill((2, 64, 4, 4), 11, -1, 1).cuda()
    skips = [fill((2, 64, 4, 4), 20 + i, -1, 1).cuda() for i in range(4)]
    Y = labels((2, 64, 64), 3, 4).cuda()
    loss_fn = seg.CrossEntropyLoss()
    loss_fn(m(x, skips), Y).backward(); opt.step(); opt.zero_grad(set_to_none=True)
    caches = ops._pack_caches(m)
    entries = [(c, ka) for c in caches for ka in c.multi]
    kinds = sorted(c.multi[ka][0] for c, ka in entries)
    assert kinds == [0] * 8 + [1] * 4 + [2] * 9 + [3] * 5, kinds
    before = {id(c._c[ka][1]): c._c[ka][1].clone() for c, ka in entries}
    ptrs = {(id(c), ka): c._c[ka][1].data_ptr() for c, ka in entries}
    ops.repack_stale(m)
    for c, ka in entries:
        kind, kb, param, d0, d1, dt = c.multi[ka]
        ver = (param._version, param.data_ptr(), param.device, ops._OPT_EPOCH[0])
        assert c._c[ka][0] == ver and (kb is None or c._c[kb][0] == ver), ka
        assert c._c[ka][1].data_ptr() == ptrs[(id(c), ka)]
        d = None
        if kind == 0:
            f, d = ops.pack_conv_both(param, d0, d1, dt)
        elif kind == 1:
            f, d = ops.pack_convt(param, dt, 0), ops.pack_convt(param, dt, 1)
        elif kind == 3:
            f = ops.pack_conv(param, d0, 0, dt, 0, taps=1)
        else:
            reps = d0 if d0 > 0 else 4
            f = torch.zeros((reps, c._c[ka][1].numel() // reps), device="cuda"); f[:, :param.shape[0]] = param.detach()
        assert torch.equal(c._c[ka][1].view_as(f), f), (kind, ka)
        if d is not None:
            assert torch.equal(c._c[kb][1], d), (kind, kb)
        assert not torch.equal(c._c[ka][1], before[id(c._c[ka][1])]), (kind, ka)      # the step really moved it
    l2 = loss_fn(m(x, skips), Y); l2.backward()
    assert torch.isfinite(l2)
    seg.set_compute_dtype(torch.bfloat16)
