"""GPU (-m gpu): the prompt model (reference prompt_based/prompt.py:6-56), its Dice + NLL losses on probabilities
(utils/weighted_loss.py:170-343) and the prompt loops (utils/training.py:153-199,242-296) on the HIP kernels, against
the fixture captured from the imported reference (tests/golden/prompt_small.npz) and the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle.fill import fill, labels, fill_module
from oracle import prompt_ref, unet_ref

pytestmark = pytest.mark.gpu
CW4 = [0.2046795970925636, 1.0271954434416883, 1.2293222812780409, 0.5]


@pytest.fixture(scope="module")
def seg():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import image_segmentation_amd as s
    return s


def cpu(t):
    return t.detach().float().cpu().numpy()


def stable_log(t):
    return torch.log(t + 1e-9)


def build_prompt(seg):
    clip = seg.unet(3, 4)                      # stand-in for the 4-class CLIP-UNet, as in tools/gen_golden.py:gen_prompt
    fill_module(clip, 9000)
    m = seg.PromptModel(clip=clip)
    fill_module(m.mask, 9500)
    return m.cuda().train()


def test_prompt_model_golden_fp32(seg, golden):
    g = golden("prompt_small")
    seg.set_compute_dtype(torch.float32)
    m = build_prompt(seg)
    assert not any(p.requires_grad for p in m.clip.parameters()) and all(p.requires_grad for p in m.mask.parameters())
    keys = list(m.state_dict().keys())
    assert "clip.down1.doubleConvReLU.0.weight" in keys and "mask.output.bias" in keys
    x = fill((2, 3, 32, 48), 1, 0, 1).cuda(); heat = fill((2, 1, 32, 48), 2, 0, 1).cuda()
    y = labels((2, 32, 48), 3, 4).cuda()
    final = m(x, heat)
    assert final.shape == (2, 4, 32, 48) and final.dtype == torch.float32
    assert np.abs(cpu(final) - g["final"]).max() < 1e-5
    assert np.abs(cpu(final.sum(1)) - 1.0).max() < 1e-5          # a probability vector per pixel
    loss_fn = seg.WeightedDiceNLLLoss(ignore_index=3, smooth_dice=1, class_weights=torch.tensor(CW4), apply_softmax=False,
                                      nll_nonlin=stable_log)
    loss = loss_fn(final, y)
    assert abs(loss.item() - float(g["loss"])) < 2e-5
    loss.backward()
    norms = {n: p.grad.double().norm().item() for n, p in m.mask.named_parameters()}
    for n, ref in zip(g["gnames"], g["gnorms"]):
        n = str(n)
        if n.endswith(("doubleConvReLU.0.bias", "doubleConvReLU.3.bias")):
            continue        # conv bias ahead of a batch-statistics BatchNorm: exactly zero here, fp32 noise in the reference
        assert abs(norms[n] - ref) <= 3e-3 * ref + 1e-7, (n, norms[n], ref)
    assert all(p.grad is None for p in m.clip.parameters())
    seg.set_compute_dtype(torch.bfloat16)


VARIANTS = {
    "prob_log": dict(apply_softmax=False, nll_nonlin=stable_log, ignore_index=3, class_weights=CW4, smooth_dice=1),
    "prob_log_plain": dict(apply_softmax=False, nll_nonlin=stable_log),
    "prob_identity": dict(apply_softmax=False, dice_weight=0.7, nll_weight=0.3),
    "default_softmax": dict(class_weights=CW4),
    "softmax_log": dict(nll_nonlin=torch.log, ignore_index=0),
}


@pytest.mark.parametrize("tag", list(VARIANTS))
def test_dice_nll_loss_golden(seg, golden, tag):
    g = golden("prompt_small")
    kw = dict(VARIANTS[tag])
    if "class_weights" in kw:
        kw["class_weights"] = torch.tensor(kw["class_weights"])
    pr = torch.softmax(fill((2, 4, 12, 20), 41, -3, 3), 1).cuda().requires_grad_(True)
    Y = labels((2, 12, 20), 42, 4).cuda()
    l = seg.WeightedDiceNLLLoss(**kw)(pr, Y.unsqueeze(1) if tag == "prob_log_plain" else Y)
    l.backward()
    assert abs(l.item() - float(g[tag + ".loss"])) < 2e-5, tag
    ref = g[tag + ".grad"]
    assert np.abs(cpu(pr.grad) - ref).max() < 1e-4 * max(1.0, np.abs(ref).max()), tag


def test_dice_prompt_loss_golden_and_errors(seg, golden):
    g = golden("prompt_small")
    Y = labels((2, 12, 20), 42, 4).cuda()
    for tag, kw in {"dicep_prob": dict(apply_softmax=False, class_weights=torch.tensor(CW4), ignore_index=3, smooth=1),
                    "dicep_softmax": dict()}.items():
        pr = torch.softmax(fill((2, 4, 12, 20), 41, -3, 3), 1).cuda().requires_grad_(True)
        l = seg.WeightedMemoryEfficientDiceLossPrompt(**kw)(pr, Y.unsqueeze(1))
        l.backward()
        assert abs(l.item() - float(g[tag + ".loss"])) < 2e-5
        ref = g[tag + ".grad"]
        assert np.abs(cpu(pr.grad) - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())
    pr = torch.zeros((2, 4, 12, 20), device="cuda")
    with pytest.raises(ValueError, match="Shape mismatch"):          # the reference rejects [N,H,W] here
        seg.WeightedMemoryEfficientDiceLossPrompt()(pr, Y)
    with pytest.raises(ValueError, match="multiple channels"):
        seg.WeightedDiceNLLLoss()(pr, torch.zeros((2, 4, 12, 20), device="cuda"))
    with pytest.raises(NotImplementedError):
        seg.WeightedDiceNLLLoss(nll_nonlin=torch.sqrt)
    with pytest.raises(RuntimeError, match="no CPU path"):
        seg.WeightedDiceNLLLoss(apply_softmax=False)(pr.cpu(), Y.cpu())


def test_prompt_loops_vs_oracle(seg):
    """train_loop_prompt / eval_loop_prompt (training.py:153-199,242-296) on the HIP model against the same protocol
    driven on the CPU oracle: per-epoch average loss, metrics, parameters after the steps."""
    from image_segmentation_amd import training
    seg.set_compute_dtype(torch.float32)
    training.VERBOSE = False
    m = build_prompt(seg)
    rc = unet_ref.unet(3, 4); rm = unet_ref.unet(4, 1)
    fill_module(rc, 9000); fill_module(rm, 9500)
    for p in rc.parameters():
        p.requires_grad = False
    rc.train(); rm.train()
    cw = torch.tensor(CW4)
    data = [(fill((2, 3, 32, 32), 10 + i, 0, 1), fill((2, 1, 32, 32), 20 + i, 0, 1), labels((2, 1, 32, 32), 30 + i, 4))
            for i in range(3)]
    opt = torch.optim.AdamW(m.mask.parameters(), weight_decay=0.01)
    ropt = torch.optim.AdamW(rm.parameters(), weight_decay=0.01)
    loss_fn = seg.WeightedDiceNLLLoss(ignore_index=3, smooth_dice=1, class_weights=cw, apply_softmax=False, nll_nonlin=stable_log)
    avg = training.train_loop_prompt(data, m, loss_fn, opt, 2, torch.device("cuda"))
    # the same protocol on the oracle (training.py:171-199)
    tot, nproc = 0.0, 0
    ropt.zero_grad()
    for i, (X, p, y) in enumerate(data):
        pred = prompt_ref.prompt_mix(rc(X), rm(torch.cat([X, p], 1)))
        loss = prompt_ref.dice_nll(pred, y.squeeze(1), ignore_index=3, class_weights=cw, smooth_dice=1, apply_softmax=False,
                                   nll_nonlin=stable_log)
        (loss / 2).backward()
        if (i + 1) % 2 == 0 or i + 1 == len(data):
            ropt.step(); ropt.zero_grad()
            tot += loss.item(); nproc += 1
    assert abs(avg - tot / nproc) < 5e-5
    for (n, a), (_, b) in zip(m.mask.named_parameters(), rm.named_parameters()):
        assert np.abs(cpu(a) - b.detach().numpy()).max() < 2e-3, n       # AdamW's first steps move every weight by ~lr
    # evaluation at the original (ragged) sizes
    agg = training.MetricsHistory(4, ignore_index=3)
    evald = [([fill((3, 24, 32), 40, 0, 1), fill((3, 32, 20), 41, 0, 1)], [fill((1, 24, 32), 42, 0, 1), fill((1, 32, 20), 43, 0, 1)],
              [labels((1, 24, 32), 44, 3), labels((1, 32, 20), 45, 3)])]
    val_fn = seg.WeightedDiceNLLLoss(ignore_index=3, class_weights=cw, apply_softmax=False, nll_nonlin=stable_log)
    vl, vd, vi = training.eval_loop_prompt(evald, m, val_fn, torch.device("cuda"), 32, agg)
    assert np.isfinite(vl) and 0.0 <= vi <= 1.0 and 0.0 <= vd <= 1.0
    training.VERBOSE = True
    seg.set_compute_dtype(torch.bfloat16)


def test_prompt_model_with_clipunet_bf16(seg):
    """The composition the reference ships: frozen ClipUNet (local random ViT-B/16 config) + unet(4,1), bf16 mode."""
    pytest.importorskip("transformers")
    seg.set_compute_dtype(torch.bfloat16)
    clip = seg.ClipUNet(num_classes=4, encoder=seg.ClipViTEncoder.from_config())
    m = seg.PromptModel(clip=clip).cuda().train()
    x = fill((2, 3, 224, 224), 1, 0, 1).cuda(); heat = fill((2, 1, 224, 224), 2, 0, 1).cuda()
    y = labels((2, 224, 224), 3, 4).cuda()
    final = m(x, heat)
    assert final.shape == (2, 4, 224, 224) and torch.isfinite(final).all()
    assert np.abs(cpu(final.sum(1)) - 1.0).max() < 1e-5
    loss = seg.WeightedDiceNLLLoss(ignore_index=3, smooth_dice=1, apply_softmax=False, nll_nonlin=stable_log)(final, y)
    loss.backward()
    assert torch.isfinite(loss) and all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.mask.parameters())
    assert all(p.grad is None for p in m.clip.parameters())
