"""CPU: the oracle (oracle/) against the golden vectors captured from the imported reference
(tools/gen_golden.py) and against the SURVEY.md 8c known answers.  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from oracle.fill import fill, labels, fill_module, u01
from oracle import unet_ref, clipunet_ref, losses_ref

CW3 = [0.2046795970925636, 1.0271954434416883, 1.2293222812780409]


def close(a, b, rtol=1e-5, atol=1e-6):
    np.testing.assert_allclose(np.asarray(a), np.asarray(b), rtol=rtol, atol=atol)


def test_fill_known_answer():
    close(u01(4, 7), [0.1738678217, 0.8773486614, 0.7263535857, 0.1351458430], rtol=0, atol=1e-9)


def _check_grads(model, g, atol=2e-5):
    for n, p in model.named_parameters():
        close(p.grad.numpy(), g["grad." + n], rtol=1e-4, atol=atol)
    for n, b in model.named_buffers():
        close(b.numpy(), g["buf." + n], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("tag,din,dout,shape", [("doubleconv_3_8", 3, 8, (2, 3, 16, 16)),
                                                ("doubleconv_32_64", 32, 64, (2, 32, 24, 40))])
def test_doubleconv(golden, tag, din, dout, shape):
    g = golden(tag)
    m = unet_ref.DoubleConvReLU(din, dout); fill_module(m, 1000); m.train()
    x = fill(shape, 1, -1, 1).requires_grad_(True)
    y = m(x)
    (y * fill(tuple(y.shape), 5, -1, 1)).sum().backward()
    close(y.detach(), g["y"]); close(x.grad, g["dx"], atol=2e-5)
    _check_grads(m, g)
    m.eval()
    with torch.no_grad():
        close(m(fill(shape, 1, -1, 1)), golden(tag + "_eval")["y"])


def test_doubleconv_survey_answer_A():
    m = unet_ref.DoubleConvReLU(3, 8); fill_module(m, 1000); m.train()
    y = m(fill((2, 3, 16, 16), 1, -1, 1))
    assert abs(y.double().sum().item() - 1660.802708) < 1e-2
    assert abs(y.max().item() - 3.150337) < 1e-4
    close(y[0, 0, 0, :4].detach(), [0.554044, 0, 0, 0], atol=1e-5)
    bn1 = m.doubleConvReLU[1]
    close(bn1.running_mean[:3], [-0.006203, -0.000819, -0.006024], atol=1e-6)
    close(bn1.running_var[:3], [0.909010, 0.910909, 0.909674], atol=1e-6)


def test_down_up(golden):
    g = golden("down_32_64")
    m = unet_ref.Down(32, 64); fill_module(m, 2000); m.train()
    x = fill((2, 32, 32, 32), 1, -1, 1).requires_grad_(True)
    y = m(x); (y * fill(tuple(y.shape), 5, -1, 1)).sum().backward()
    close(y.detach(), g["y"]); close(x.grad, g["dx"], atol=2e-5); _check_grads(m, g)

    g = golden("up_64_32")
    m = unet_ref.Up(64, 32); fill_module(m, 3000); m.train()
    x1 = fill((2, 32, 32, 32), 1, -1, 1).requires_grad_(True)
    x2 = fill((2, 64, 16, 16), 2, -1, 1).requires_grad_(True)
    y = m(x1, x2); (y * fill(tuple(y.shape), 5, -1, 1)).sum().backward()
    close(y.detach(), g["y"]); close(x1.grad, g["dx1"], atol=2e-5); close(x2.grad, g["dx2"], atol=2e-5)
    _check_grads(m, g)


def test_unet_config1(golden):
    """BASELINE config 1: unet(3,3), 4x3x128x128, CE -- logits, losses, grads, argmax, metrics."""
    g = golden("unet_3_3_b4_128")
    m = unet_ref.unet(3, 3); fill_module(m, 1000); m.train()
    X = fill((4, 3, 128, 128), 1, 0, 1); Y = labels((4, 1, 128, 128), 2, 3)
    lg = m(X)
    assert np.abs(lg.detach().numpy() - g["logits"]).max() < 1e-4
    assert abs(lg.double().sum().item() - (-33843.182145)) < 0.5          # SURVEY answer B
    w = torch.tensor(CW3)
    ce = losses_ref.cross_entropy(lg, Y.squeeze(1))
    assert abs(ce.item() - float(g["ce"])) < 1e-5 and abs(ce.item() - 1.137935) < 1e-5
    assert abs(losses_ref.cross_entropy(lg, Y.squeeze(1), w).item() - float(g["wce"])) < 1e-5
    assert abs(losses_ref.soft_dice(lg, Y, 1.0, w).item() - float(g["dice"])) < 1e-5
    assert abs(losses_ref.dice_ce(lg, Y, class_weights=w, smooth_dice=1.0).item() - float(g["dicece"])) < 1e-5
    ce.backward()
    norms = {n: p.grad.double().norm().item() for n, p in m.named_parameters()}
    for n, ref in zip(g["grad_names"], g["grad_norms"]):
        assert abs(norms[str(n)] - ref) <= 1e-3 * ref + 2e-7, n
    assert (lg.argmax(1).numpy().astype(np.uint8) == g["argmax"]).all()
    counts = sum(losses_ref.confusion_counts(lg[i].detach(), Y[i, 0], 3) for i in range(4))
    assert (counts.numpy() == g["counts"]).all()
    d, i, a, pc = losses_ref.epoch_metrics(counts)
    close([d, i, a], g["metrics"], rtol=1e-12, atol=0)
    close([d, i, a], [0.3136445809, 0.1892491490, 0.5557556152], atol=1e-9)
    m.eval()
    with torch.no_grad():
        ev = m(X)
    assert abs(ev.double().sum().item() - float(g["eval_logits_sum"])) < 0.5
    close(ev[:, :, ::16, ::16], g["eval_logits_sample"], atol=1e-4)


def test_unet_dicece_grads(golden):
    g = golden("unet_3_3_b2_32x48_dicece")
    m = unet_ref.unet(3, 3); fill_module(m, 1000); m.train()
    X = fill((2, 3, 32, 48), 3, 0, 1); Y = labels((2, 1, 32, 48), 4, 3)
    lg = m(X); close(lg.detach(), g["logits"], atol=1e-4)
    loss = losses_ref.dice_ce(lg, Y, class_weights=torch.tensor(CW3), smooth_dice=1.0)
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    loss.backward()
    norms = {n: p.grad.double().norm().item() for n, p in m.named_parameters()}
    for n, ref in zip(g["grad_names"], g["grad_norms"]):
        assert abs(norms[str(n)] - ref) <= 2e-3 * ref + 2e-7, n


@pytest.mark.parametrize("tag,cin,cskip,cout,b,h,base", [("decoderblock_16_12_8", 16, 12, 8, 1, 3, 7000),
                                                       ("decoderblock_64_96_32", 64, 96, 32, 2, 7, 7100)])
def test_decoder_block(golden, tag, cin, cskip, cout, b, h, base):
    g = golden(tag)
    m = clipunet_ref.DecoderBlock(cin, cskip, cout); fill_module(m, base); m.train()
    x = fill((b, cin, h, h), 31, -1, 1).requires_grad_(True)
    sk = fill((b, cskip, h, h), 32, -1, 1).requires_grad_(True)
    y = m(x, sk); (y * fill(tuple(y.shape), 5, -1, 1)).sum().backward()
    close(y.detach(), g["y"]); close(x.grad, g["dx"], atol=2e-5); close(sk.grad, g["dskip"], atol=2e-5)
    _check_grads(m, g)


def test_clip_decoder(golden):
    g = golden("clip_decoder_b2")
    dec = clipunet_ref.UNetDecoder(768, [1024, 512, 256, 128, 64]); head = torch.nn.Conv2d(64, 4, 1)
    both = torch.nn.ModuleDict({"decoder": dec, "output_layer": head}); fill_module(both, 5000); both.train()
    x = fill((2, 768, 14, 14), 11, -1, 1)
    skips = [fill((2, 768, 14, 14), 20 + i, -1, 1) for i in range(4)]
    d = dec(x, skips); lg = head(d)
    assert abs(d.double().sum().item() - float(g["dec_sum"])) < 5.0
    close(lg[:, :, ::8, ::8].detach(), g["logits_sample"], atol=1e-4)
    Y = labels((2, 224, 224), 3, 4)
    ce = losses_ref.cross_entropy(lg, Y)
    # (SURVEY answer C used an unstated module enumeration for its fill; the golden generated
    #  from the imported reference with ModuleDict{decoder, output_layer}, base 5000, is the pin.)
    assert abs(ce.item() - float(g["ce"])) < 1e-5
    ce.backward()
    norms = {n: p.grad.double().norm().item() for n, p in both.named_parameters()}
    for n, ref in zip(g["grad_names"], g["grad_norms"]):
        assert abs(norms[str(n)] - ref) <= 1e-3 * ref + 2e-7, n


def test_losses_small(golden):
    g = golden("losses_small")
    lg0 = fill((2, 4, 12, 20), 41, -3, 3); Y = labels((2, 12, 20), 42, 4)
    w4 = torch.tensor([0.3, 1.1, 0.9, 1.7])
    cases = {
        "ce": lambda l: losses_ref.cross_entropy(l, Y),
        "ce_w": lambda l: losses_ref.cross_entropy(l, Y, w4),
        "ce_w_ign3": lambda l: losses_ref.cross_entropy(l, Y, w4, 3),
        "dice": lambda l: losses_ref.soft_dice(l, Y.unsqueeze(1), 1e-5),
        "dice_w_ign3": lambda l: losses_ref.soft_dice(l, Y.unsqueeze(1), 1.0, w4, 3),
        "dicece": lambda l: losses_ref.dice_ce(l, Y),
        "dicece_w_ign3": lambda l: losses_ref.dice_ce(l, Y.unsqueeze(1), 0.7, 1.3, 3, w4, 1.0),
    }
    for k, fn in cases.items():
        l = lg0.clone().requires_grad_(True)
        v = fn(l); v.backward()
        assert abs(v.item() - float(g[k])) < 2e-6, k
        close(l.grad, g[k + "_grad"], rtol=1e-4, atol=1e-8)
    with pytest.raises(ValueError):                       # reference rejects [N,H,W] for bare Dice
        losses_ref.soft_dice(lg0, Y)


def _check_grad_summary(model, g, rtol=2e-4, atol=2e-6):
    norms = dict(zip([str(n) for n in g["gnames"]], g["gnorms"]))
    heads = dict(zip([str(n) for n in g["gnames"]], g["gheads"]))
    for n, p in model.named_parameters():
        gr = p.grad if p.grad is not None else torch.zeros_like(p)
        assert abs(gr.double().norm().item() - norms[n]) <= rtol * norms[n] + atol, n
        h = gr.flatten()[:8].numpy()
        close(h, heads[n][:h.size], rtol=2e-3, atol=2e-5)


def test_autoencoder_family(golden):
    """oracle/autoencoder_ref.py vs goldens from the imported reference autoencoder/autoencoder.py"""
    from oracle import autoencoder_ref as ae
    x = fill((2, 3, 32, 32), 1, 0, 1)
    y = labels((2, 32, 32), 2, 3)
    g = golden("autoencoder_seg_b2_32")
    m = ae.SegmentationAutoencoder(3, base_channels=32, num_classes=3, freeze_encoder=False); fill_module(m, 3000); m.train()
    logits = m(x)
    loss = torch.nn.functional.cross_entropy(logits, y); loss.backward()
    close(logits.detach(), g["logits"], atol=2e-5); assert abs(loss.item() - float(g["loss"])) < 1e-5
    _check_grad_summary(m, g)
    close(m.encoder.encoder.encoderPart1.bn1.running_mean, g["buf.encoder.encoder.encoderPart1.bn1.running_mean"])
    close(m.decoder.decoderBlock3.convs[4].running_var, g["buf.decoder.decoderBlock3.convs.4.running_var"])
    gf = golden("autoencoder_seg_frozen_b2_32")
    mf = ae.SegmentationAutoencoder(3, base_channels=32, num_classes=3, freeze_encoder=True); fill_module(mf, 3000); mf.train()
    lf = torch.nn.functional.cross_entropy(mf(x), y); lf.backward()
    assert abs(lf.item() - float(gf["loss"])) < 1e-5
    assert all(p.grad is None for p in mf.encoder.parameters())
    _check_grad_summary(mf, gf)
    gr = golden("autoencoder_rec_b2_32")
    r = ae.ReconstructionAutoencoder(3, 3, base_channels=32); fill_module(r, 4000); r.train()
    rec = r(x)
    lr_ = torch.nn.functional.mse_loss(rec, x); lr_.backward()
    close(rec.detach(), gr["rec"], atol=2e-6); assert abs(lr_.item() - float(gr["loss"])) < 1e-6
    _check_grad_summary(r, gr)


def small_vit():
    transformers = pytest.importorskip("transformers")
    cfg = transformers.CLIPVisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=4,
                                        num_attention_heads=2, image_size=80, patch_size=16)
    m = transformers.CLIPVisionModel(cfg); fill_module(m, 8000); m.eval()
    return m


def test_clip_vit_restatement(golden):
    """oracle/clip_vit_ref.py (plain-torch restatement of transformers' CLIP ViT) against the fixture captured from
    the real CLIPVisionModel driven as reference clip/clipunet.py:41-63 drives it."""
    from oracle import clip_vit_ref
    g = golden("clip_vit_small")
    m = small_vit()
    x = fill((2, 3, 80, 80), 9, -1, 1)
    hs = clip_vit_ref.hidden_states(m, x)
    assert len(hs) == int(g["n_hidden"]) == 5
    bott, skips = clip_vit_ref.encoder_features(m, x, [3, 1, 2])
    close(bott.numpy(), g["bottleneck"], rtol=1e-4, atol=2e-5)
    for i, sk in zip((1, 2, 3), skips):
        close(sk.numpy(), g[f"skip{i}"], rtol=1e-4, atol=2e-5)
    close(hs[-1][:, 0].numpy(), g["cls_last"], rtol=1e-4, atol=2e-5)


CW4 = [0.2046795970925636, 1.0271954434416883, 1.2293222812780409, 0.5]
PROMPT_VARIANTS = {
    "prob_log": dict(apply_softmax=False, log_eps=1e-9, ignore_index=3, class_weights=CW4, smooth_dice=1),
    "prob_log_plain": dict(apply_softmax=False, log_eps=1e-9),
    "prob_identity": dict(apply_softmax=False, dice_weight=0.7, nll_weight=0.3),
    "default_softmax": dict(class_weights=CW4),
    "softmax_log": dict(log_eps=0.0, ignore_index=0),
}


def _nonlin(eps):
    return None if eps is None else (lambda t: torch.log(t + eps))


def test_prompt_model_and_losses(golden):
    """oracle/prompt_ref.py against the fixture from the reference's PromptModel.forward and WeightedDiceNLLLoss."""
    from oracle import prompt_ref
    g = golden("prompt_small")
    clip = unet_ref.unet(3, 4); mask = unet_ref.unet(4, 1)
    fill_module(clip, 9000); fill_module(mask, 9500)
    for p in clip.parameters():
        p.requires_grad = False
    clip.train(); mask.train()
    x = fill((2, 3, 32, 48), 1, 0, 1); heat = fill((2, 1, 32, 48), 2, 0, 1)
    y = labels((2, 32, 48), 3, 4)
    final = prompt_ref.prompt_mix(clip(x), mask(torch.cat([x, heat], 1)))
    close(final.detach().numpy(), g["final"], rtol=1e-4, atol=2e-6)
    cw = torch.tensor(CW4)
    loss = prompt_ref.dice_nll(final, y, ignore_index=3, class_weights=cw, smooth_dice=1, apply_softmax=False,
                               nll_nonlin=_nonlin(1e-9))
    assert abs(loss.item() - float(g["loss"])) < 2e-6
    loss.backward()
    norms = {n: p.grad.double().norm().item() for n, p in mask.named_parameters()}
    for n, ref in zip(g["gnames"], g["gnorms"]):
        assert abs(norms[str(n)] - ref) <= 2e-4 * ref + 1e-9, n
    pr = torch.softmax(fill((2, 4, 12, 20), 41, -3, 3), 1)
    Y = labels((2, 12, 20), 42, 4)
    for tag, kw in PROMPT_VARIANTS.items():
        kw = dict(kw)
        eps = kw.pop("log_eps", None)
        if "class_weights" in kw:
            kw["class_weights"] = torch.tensor(kw["class_weights"])
        inp = pr.clone().requires_grad_(True)
        l = prompt_ref.dice_nll(inp, Y, nll_nonlin=_nonlin(eps), **kw)
        l.backward()
        assert abs(l.item() - float(g[tag + ".loss"])) < 2e-6, tag
        close(inp.grad.numpy(), g[tag + ".grad"], rtol=1e-4, atol=1e-8)
    for tag, kw in {"dicep_prob": dict(apply_softmax=False, class_weights=cw, ignore_index=3, smooth=1),
                    "dicep_softmax": dict()}.items():
        inp = pr.clone().requires_grad_(True)
        l = prompt_ref.dice_prompt(inp, Y, **kw)
        l.backward()
        assert abs(l.item() - float(g[tag + ".loss"])) < 2e-6, tag
        close(inp.grad.numpy(), g[tag + ".grad"], rtol=1e-4, atol=1e-8)
