#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing the REFERENCE from /root/reference (build container only).

The reference never travels to the GPU box; only the fixtures written here (inputs are NOT stored
-- they are regenerated from the portable splitmix64 fill, oracle/fill.py -- plus expected outputs)
are committed.  Re-run:  python tools/gen_golden.py     (needs /root/reference, CPU only)

Fixture contents are data (numbers), never reference source text.
"""
import os, sys, json
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("SEG_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.path.insert(1, ROOT)

from oracle.fill import fill, labels, fill_module          # noqa: E402
from unet.unet import unet as RefUnet, DoubleConvReLU as RefDC, Down as RefDown, Up as RefUp  # noqa: E402
from clip.clipunet import DecoderBlock as RefBlock, UNetDecoder as RefDecoder               # noqa: E402
from utils.weighted_loss import WeightedDiceCELoss as RefDiceCE, \
    WeightedMemoryEfficientDiceLoss as RefDice                                               # noqa: E402
from utils.MetricsHistory import MetricsHistory as RefMetrics                                # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)
META = {"torch": torch.__version__, "threads": torch.get_num_threads(), "dtype": "float32"}
ONLY = os.environ.get("GOLDEN_ONLY")      # e.g. GOLDEN_ONLY=clip_vit regenerates one family; unset = all


def want(family):
    return ONLY is None or ONLY == family


CW3 = [0.2046795970925636, 1.0271954434416883, 1.2293222812780409]


def npy(t):
    return t.detach().cpu().numpy()


def grad_summary(model):
    """per-parameter grad L2 norm + first 8 values (full grads would be >100 MB for the U-Net)."""
    names, norms, heads = [], [], []
    for n, p in model.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        names.append(n)
        norms.append(g.double().norm().item())
        h = g.flatten()[:8]
        heads.append(np.pad(npy(h), (0, 8 - h.numel())))
    return np.array(names), np.array(norms), np.stack(heads)


def save(name, **kw):
    kw["meta"] = np.array(json.dumps(META))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **kw)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in kw.items() if k != "meta"})


def full_grads(model):
    return {"grad." + n: npy(p.grad) for n, p in model.named_parameters()}


def buffers(model):
    return {"buf." + n: npy(b) for n, b in model.named_buffers()}


# ---- A: DoubleConvReLU(3,8) (SURVEY 8c answer A) and a kernel-friendly (32,64) case ------------
def gen_doubleconv(tag, din, dout, shape, base=1000):
    m = RefDC(din, dout); fill_module(m, base); m.train()
    x = fill(shape, 1, -1, 1).requires_grad_(True)
    y = m(x)
    gy = fill(tuple(y.shape), 5, -1, 1)
    (y * gy).sum().backward()
    save(tag, din=din, dout=dout, shape=np.array(shape), y=npy(y), dx=npy(x.grad),
         **full_grads(m), **buffers(m))
    m.eval()
    with torch.no_grad():
        save(tag + "_eval", y=npy(m(fill(shape, 1, -1, 1))))


if want("doubleconv"):
    gen_doubleconv("doubleconv_3_8", 3, 8, (2, 3, 16, 16))
    gen_doubleconv("doubleconv_32_64", 32, 64, (2, 32, 24, 40))


# ---- Down / Up blocks ---------------------------------------------------------------------------
def gen_down():
    m = RefDown(32, 64); fill_module(m, 2000); m.train()
    x = fill((2, 32, 32, 32), 1, -1, 1).requires_grad_(True)
    y = m(x); gy = fill(tuple(y.shape), 5, -1, 1); (y * gy).sum().backward()
    save("down_32_64", y=npy(y), dx=npy(x.grad), **full_grads(m), **buffers(m))


def gen_up():
    m = RefUp(64, 32); fill_module(m, 3000); m.train()
    x1 = fill((2, 32, 32, 32), 1, -1, 1).requires_grad_(True)
    x2 = fill((2, 64, 16, 16), 2, -1, 1).requires_grad_(True)
    y = m(x1, x2); gy = fill(tuple(y.shape), 5, -1, 1); (y * gy).sum().backward()
    save("up_64_32", y=npy(y), dx1=npy(x1.grad), dx2=npy(x2.grad), **full_grads(m), **buffers(m))


if want("downup"): gen_down(); gen_up()


# ---- B: unet(3,3) on 4x3x128x128 (BASELINE config 1; SURVEY 8c answer B) ------------------------
def gen_unet():
    m = RefUnet(3, 3); fill_module(m, 1000); m.train()
    X = fill((4, 3, 128, 128), 1, 0, 1)
    Y = labels((4, 1, 128, 128), 2, 3)
    logits = m(X)
    ce = torch.nn.CrossEntropyLoss()(logits, Y.squeeze(1))
    w = torch.tensor(CW3)
    wce = torch.nn.CrossEntropyLoss(weight=w)(logits, Y.squeeze(1))
    dice = RefDice(smooth=1.0, class_weights=w)(logits, Y)
    dicece = RefDiceCE(smooth_dice=1.0, class_weights=w)(logits, Y)
    ce.backward()
    names, norms, heads = grad_summary(m)
    agg = RefMetrics(3)
    for i in range(4):
        agg.accumulate(logits[i].detach(), Y[i, 0])
    mdice, miou, macc = agg.compute_epoch_metrics()
    counts = np.stack([npy(agg.total_tp), npy(agg.total_fp), npy(agg.total_fn), npy(agg.total_tn)])
    bufs = {"buf." + n: npy(b) for n, b in m.named_buffers() if b.numel() <= 64 or "down1" in n}
    m.eval()
    with torch.no_grad():
        ev = m(X)
    save("unet_3_3_b4_128", logits=npy(logits), ce=ce.item(), wce=wce.item(), dice=dice.item(),
         dicece=dicece.item(), grad_names=names, grad_norms=norms, grad_heads=heads,
         argmax=npy(logits.argmax(1)).astype(np.uint8), counts=counts,
         metrics=np.array([mdice, miou, macc]), per_class_iou=npy(agg.get_last_per_class_iou()),
         eval_logits_sum=ev.double().sum().item(), eval_logits_sample=npy(ev[:, :, ::16, ::16]), **bufs)

    # a second, smaller full-model case whose dice+CE gradient is taken (config-5 style loss)
    m2 = RefUnet(3, 3); fill_module(m2, 1000); m2.train()
    X2 = fill((2, 3, 32, 48), 3, 0, 1); Y2 = labels((2, 1, 32, 48), 4, 3)
    lg = m2(X2)
    loss = RefDiceCE(smooth_dice=1.0, class_weights=w)(lg, Y2)
    loss.backward()
    n2, no2, h2 = grad_summary(m2)
    save("unet_3_3_b2_32x48_dicece", logits=npy(lg), loss=loss.item(), grad_names=n2, grad_norms=no2,
         grad_heads=h2)


if want("unet"): gen_unet()


# ---- C/D: CLIP decoder (SURVEY 8c answers C, D) -------------------------------------------------
def gen_clip():
    blk = RefBlock(16, 12, 8); fill_module(blk, 7000); blk.train()
    x = fill((1, 16, 3, 3), 31, -1, 1).requires_grad_(True)
    sk = fill((1, 12, 3, 3), 32, -1, 1).requires_grad_(True)
    y = blk(x, sk); gy = fill(tuple(y.shape), 5, -1, 1); (y * gy).sum().backward()
    save("decoderblock_16_12_8", y=npy(y), dx=npy(x.grad), dskip=npy(sk.grad), **full_grads(blk),
         **buffers(blk))

    blk = RefBlock(64, 96, 32); fill_module(blk, 7100); blk.train()
    x = fill((2, 64, 7, 7), 31, -1, 1).requires_grad_(True)
    sk = fill((2, 96, 7, 7), 32, -1, 1).requires_grad_(True)
    y = blk(x, sk); gy = fill(tuple(y.shape), 5, -1, 1); (y * gy).sum().backward()
    save("decoderblock_64_96_32", y=npy(y), dx=npy(x.grad), dskip=npy(sk.grad), **full_grads(blk),
         **buffers(blk))

    dec = RefDecoder(768, [1024, 512, 256, 128, 64]); head = torch.nn.Conv2d(64, 4, 1)
    both = torch.nn.ModuleDict({"decoder": dec, "output_layer": head})
    fill_module(both, 5000); both.train()
    x = fill((2, 768, 14, 14), 11, -1, 1)
    skips = [fill((2, 768, 14, 14), 20 + i, -1, 1) for i in range(4)]
    d = dec(x, skips); lg = head(d)
    Y = labels((2, 224, 224), 3, 4)
    ce = torch.nn.CrossEntropyLoss()(lg, Y); ce.backward()
    names, norms, heads = grad_summary(both)
    save("clip_decoder_b2", dec_sum=d.double().sum().item(), dec_max=d.max().item(),
         logits_sum=lg.double().sum().item(), logits_sample=npy(lg[:, :, ::8, ::8]),
         dec_sample=npy(d[:, :, ::16, ::16]), ce=ce.item(), grad_names=names, grad_norms=norms,
         grad_heads=heads, argmax_sample=npy(lg.argmax(1)[:, ::4, ::4]).astype(np.uint8))


if want("clip"): gen_clip()


# ---- losses on small logits with ignore_index, incl. gradient wrt logits ------------------------
def small_vit_config():
    from transformers import CLIPVisionConfig
    return CLIPVisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=4, num_attention_heads=2,
                            image_size=80, patch_size=16)


def gen_clip_vit():
    """The third-party encoder driven exactly as the reference's ClipViTEncoder.forward does (clipunet.py:41-63):
    CLIPVisionModel(pixel_values, output_hidden_states=True); last_hidden_state and hidden_states[i] without CLS,
    reshaped to [B,D,G,G].  Local random config (no hub access), portable-fill weights base 8000."""
    from transformers import CLIPVisionModel
    cfg = small_vit_config()
    m = CLIPVisionModel(cfg); fill_module(m, 8000); m.eval()
    x = fill((2, 3, 80, 80), 9, -1, 1)
    with torch.no_grad():
        out = m(pixel_values=x, output_hidden_states=True)
    g = cfg.image_size // cfg.patch_size

    def grid(hs):
        return npy(hs[:, 1:, :].reshape(2, g, g, cfg.hidden_size).permute(0, 3, 1, 2).contiguous())
    save("clip_vit_small", bottleneck=grid(out.last_hidden_state),
         **{f"skip{i}": grid(out.hidden_states[i]) for i in (1, 2, 3)},
         cls_last=npy(out.last_hidden_state[:, 0]), n_hidden=np.array(len(out.hidden_states)),
         transformers=np.array(__import__("transformers").__version__))


if want("clip_vit"): gen_clip_vit()


def gen_losses():
    out = {}
    lg0 = fill((2, 4, 12, 20), 41, -3, 3)
    Y = labels((2, 12, 20), 42, 4)
    w4 = torch.tensor([0.3, 1.1, 0.9, 1.7])
    cases = {
        "ce": lambda l: torch.nn.CrossEntropyLoss()(l, Y),
        "ce_w": lambda l: torch.nn.CrossEntropyLoss(weight=w4)(l, Y),
        "ce_w_ign3": lambda l: torch.nn.CrossEntropyLoss(weight=w4, ignore_index=3)(l, Y),
        "dice": lambda l: RefDice(smooth=1e-5)(l, Y.unsqueeze(1)),
        "dice_w_ign3": lambda l: RefDice(smooth=1.0, class_weights=w4, ignore_index=3)(l, Y.unsqueeze(1)),
        "dicece": lambda l: RefDiceCE()(l, Y),
        "dicece_w_ign3": lambda l: RefDiceCE(dice_weight=0.7, ce_weight=1.3, ignore_index=3,
                                             class_weights=w4, smooth_dice=1.0)(l, Y.unsqueeze(1)),
    }
    for k, fn in cases.items():
        l = lg0.clone().requires_grad_(True)
        v = fn(l); v.backward()
        out[k] = v.item(); out[k + "_grad"] = npy(l.grad)
    save("losses_small", **out)


if want("losses"): gen_losses()


CW4 = [0.2046795970925636, 1.0271954434416883, 1.2293222812780409, 0.5]


def gen_prompt():
    """The reference's own PromptModel.forward (prompt_based/prompt.py:33-56) and WeightedDiceNLLLoss, imported.  The
    constructor cannot run offline (ClipUNet() fetches from the hub), so the instance is assembled by hand around the
    reference's forward: `clip` = a frozen reference unet(3,4) standing in for the 4-class CLIP-UNet, `mask` = the
    reference unet(4,1) exactly as prompt.py:16 builds it."""
    from prompt_based.prompt import PromptModel as RefPrompt
    from utils.weighted_loss import WeightedDiceNLLLoss as RefDiceNLL, WeightedMemoryEfficientDiceLossPrompt as RefDiceP
    m = RefPrompt.__new__(RefPrompt)
    torch.nn.Module.__init__(m)
    m.clip = RefUnet(3, 4); m.mask = RefUnet(4, 1)
    m.softmax = torch.nn.Softmax(dim=1); m.sigmoid = torch.nn.Sigmoid()
    fill_module(m.clip, 9000); fill_module(m.mask, 9500)
    for p in m.clip.parameters():
        p.requires_grad = False
    m.train()
    x = fill((2, 3, 32, 48), 1, 0, 1); heat = fill((2, 1, 32, 48), 2, 0, 1)
    y = labels((2, 32, 48), 3, 4)
    cw = torch.tensor(CW4)
    stable_log = lambda t: torch.log(t + 1e-9)       # prompt.ipynb cell 0
    final = m(x, heat)
    loss_fn = RefDiceNLL(ignore_index=3, smooth_dice=1, class_weights=cw, apply_softmax=False, nll_nonlin=stable_log)
    loss = loss_fn(final, y)
    loss.backward()
    names, norms, heads = grad_summary(m.mask)
    out = {"final": npy(final), "loss": np.array(loss.item()), "gnames": names, "gnorms": norms, "gheads": heads}
    # the loss family on fixed probabilities / logits, values and input gradients
    pr = torch.softmax(fill((2, 4, 12, 20), 41, -3, 3), 1)
    Y = labels((2, 12, 20), 42, 4)
    variants = {
        "prob_log": dict(apply_softmax=False, nll_nonlin=stable_log, ignore_index=3, class_weights=cw, smooth_dice=1),
        "prob_log_plain": dict(apply_softmax=False, nll_nonlin=stable_log),
        "prob_identity": dict(apply_softmax=False, dice_weight=0.7, nll_weight=0.3),
        "default_softmax": dict(class_weights=cw),
        "softmax_log": dict(nll_nonlin=torch.log, ignore_index=0),
    }
    for tag, kw in variants.items():
        inp = pr.clone().requires_grad_(True)
        l = RefDiceNLL(**kw)(inp, Y if tag != "prob_log_plain" else Y.unsqueeze(1))
        l.backward()
        out[f"{tag}.loss"] = np.array(l.item()); out[f"{tag}.grad"] = npy(inp.grad)
    for tag, kw in {"dicep_prob": dict(apply_softmax=False, class_weights=cw, ignore_index=3, smooth=1),
                    "dicep_softmax": dict()}.items():
        inp = pr.clone().requires_grad_(True)
        l = RefDiceP(**kw)(inp, Y.unsqueeze(1))      # a [N,H,W] target fails the reference's own shape test (:213-217)
        l.backward()
        out[f"{tag}.loss"] = np.array(l.item()); out[f"{tag}.grad"] = npy(inp.grad)
    save("prompt_small", **out)


if want("prompt"): gen_prompt()


# ---- train_loop protocol (training.py:18-64 driven by hand: torchvision/tqdm.notebook absent) ---
def gen_trainloop():
    res = {}
    for acc in (1, 2):
        m = RefUnet(3, 3); fill_module(m, 1000)
        opt = torch.optim.AdamW(m.parameters(), weight_decay=0.01)
        data = [(fill((2, 3, 32, 32), 10 + i, 0, 1), labels((2, 1, 32, 32), 20 + i, 3)) for i in range(3)]
        loss_fn = torch.nn.CrossEntropyLoss()
        m.train(); total, nproc, per = 0.0, 0, []
        opt.zero_grad()
        for bi, (X, y) in enumerate(data):                       # training.py:38-60
            y = y.long()
            pred = m(X)
            loss = loss_fn(pred, y.squeeze(1))
            (loss / acc).backward()
            if (bi + 1) % acc == 0 or (bi + 1) == len(data):
                opt.step(); opt.zero_grad()
                total += loss.item(); nproc += 1; per.append(loss.item())
        res[f"acc{acc}_avg"] = total / nproc
        res[f"acc{acc}_losses"] = np.array(per)
        # conv biases ahead of BN have zero true gradient (fp noise only): checksum the rest
        res[f"acc{acc}_w0"] = npy(m.down1.doubleConvReLU[0].weight).copy()
        res[f"acc{acc}_out_w"] = npy(m.output.weight).copy()
        res[f"acc{acc}_rm"] = npy(m.down1.doubleConvReLU[1].running_mean).copy()
    save("trainloop_unet_32", **res)


if want("trainloop"): gen_trainloop()


# ---- autoencoder family: reference autoencoder/autoencoder.py ------------------------------------
def gen_autoencoder():
    import io, contextlib
    from autoencoder.autoencoder import SegmentationAutoencoder as RefSegAE, ReconstructionAutoencoder as RefRecAE
    B, H = 2, 32
    x = fill((B, 3, H, H), 1, 0, 1)
    y = labels((B, H, H), 2, 3)
    # segmentation autoencoder, trainable encoder, CE loss (training.py:47 style step)
    with contextlib.redirect_stdout(io.StringIO()):
        m = RefSegAE(3, base_channels=32, num_classes=3, freeze_encoder=False)
    fill_module(m, 3000); m.train()
    logits = m(x)
    loss = torch.nn.functional.cross_entropy(logits, y)
    loss.backward()
    names, norms, heads = grad_summary(m)
    save("autoencoder_seg_b2_32", logits=npy(logits), loss=np.array(loss.item()), gnames=names, gnorms=norms, gheads=heads,
         **{k: v for k, v in buffers(m).items() if "encoderPart1.bn1" in k or "decoderBlock3.convs.4" in k})
    # frozen encoder: only decoder / head parameters receive gradients
    with contextlib.redirect_stdout(io.StringIO()):
        mf = RefSegAE(3, base_channels=32, num_classes=3, freeze_encoder=True)
    fill_module(mf, 3000); mf.train()
    lf = torch.nn.functional.cross_entropy(mf(x), y)
    lf.backward()
    names, norms, heads = grad_summary(mf)
    save("autoencoder_seg_frozen_b2_32", loss=np.array(lf.item()), gnames=names, gnorms=norms, gheads=heads)
    # reconstruction autoencoder, MSE against the input
    r = RefRecAE(3, 3, base_channels=32); fill_module(r, 4000); r.train()
    rec = r(x)
    lr_ = torch.nn.functional.mse_loss(rec, x)
    lr_.backward()
    names, norms, heads = grad_summary(r)
    save("autoencoder_rec_b2_32", rec=npy(rec), loss=np.array(lr_.item()), gnames=names, gnorms=norms, gheads=heads)


if want("autoencoder"): gen_autoencoder()
print("done")
