"""GPU (-m gpu): parity at the image sizes the benchmark numbers are taken on (BASELINE configs 2 and 5): the U-Net at
1x3x256x256 and 1x3x512x512, forward + loss + backward, fp32 parity mode AND bf16 (the mode of the headline number),
against the CPU oracle (= the reference algorithm, reference unet/unet.py:93-105, utils/weighted_loss.py:140-166) on the
same seeded inputs and weights.  These shapes reach the 8x32 tile walk with tiles_x > 8, the register-stationary 256/512
levels and the multi-tile persistent loops that the small-image tests cannot.

Per-parameter checks instead of norm-only ones: fp32 mode relative L2 <= 2e-3, bf16 cosine >= 0.995 (a permuted,
sign-flipped or 10 %-off gradient fails both)."""
import numpy as np
import pytest
import torch

from oracle.fill import fill, labels, fill_module
from oracle import unet_ref, losses_ref

pytestmark = pytest.mark.gpu
CW3 = [0.2046795970925636, 1.0271954434416883, 1.2293222812780409]


@pytest.fixture(scope="module")
def seg():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import image_segmentation_amd as s
    return s


_ORACLE = {}


def oracle_run(S, loss_name):
    """CPU oracle forward + loss + backward at 1x3xSxS (cached: both compute modes compare against the same run)."""
    key = (S, loss_name)
    if key not in _ORACLE:
        torch.manual_seed(0)
        ref = unet_ref.unet(3, 3); fill_module(ref, 1000); ref.train()
        X = fill((1, 3, S, S), 1, 0, 1); Y = labels((1, S, S), 2, 3)
        lr = ref(X)
        if loss_name == "ce":
            loss = losses_ref.cross_entropy(lr, Y)
        else:
            loss = losses_ref.dice_ce(lr, Y, class_weights=torch.tensor(CW3), smooth_dice=1.0)
        loss.backward()
        _ORACLE[key] = (X, Y, lr.detach(), float(loss), {n: p.grad.detach().clone() for n, p in ref.named_parameters()})
    return _ORACLE[key]


def hip_run(seg, dtype, S, loss_name):
    X, Y, _, _, _ = oracle_run(S, loss_name)
    seg.set_compute_dtype(dtype)
    m = seg.unet(3, 3); fill_module(m, 1000); m.cuda().train()
    lg = m(X.cuda())
    if loss_name == "ce":
        loss = seg.CrossEntropyLoss()(lg, Y.cuda())
    else:
        loss = seg.WeightedDiceCELoss(smooth_dice=1.0, class_weights=torch.tensor(CW3))(lg, Y.cuda())
    loss.backward()
    torch.cuda.synchronize()
    seg.set_compute_dtype(torch.bfloat16)
    return lg.detach().float().cpu(), float(loss), {n: p.grad.detach().float().cpu() for n, p in m.named_parameters()}


def is_cancelled_bias(n):
    return n.endswith(".bias") and ("doubleConvReLU.0" in n or "doubleConvReLU.3" in n)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("S", [256, 512])
def test_unet_fp32_mode_full_size(seg, S):
    _, _, lr, loss_ref, gref = oracle_run(S, "ce")
    lg, loss, g = hip_run(seg, torch.float32, S, "ce")
    assert (lg - lr).abs().max().item() < 1e-3                        # north-star gate: logits within 1e-3
    assert torch.equal(lg.argmax(1), lr.argmax(1))                    # argmax masks bit-exact
    assert abs(loss - loss_ref) < 2e-5
    for n, r in gref.items():
        if is_cancelled_bias(n):
            assert g[n].abs().max().item() == 0.0, n                  # exact zeros where the reference holds fp32 noise
            continue
        rel = (g[n] - r).norm().item() / max(r.norm().item(), 1e-12)
        assert rel <= 2e-3, (n, rel)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("S,loss_name", [(256, "ce"), (512, "ce"), (512, "dicece")])
def test_unet_bf16_mode_full_size(seg, S, loss_name):
    """bf16 has no reference counterpart (the reference is fp32 only): gates are logits within bf16 noise of 23 conv+BN
    layers, >= 97.5 % argmax agreement, loss within 2e-2, per-parameter gradient cosine >= 0.995 and norm within 5 %
    (BatchNorm vectors / biases: cosine >= 0.98: few elements, heavy cancellation)."""
    _, _, lr, loss_ref, gref = oracle_run(S, loss_name)
    lg, loss, g = hip_run(seg, torch.bfloat16, S, loss_name)
    d = (lg - lr).abs()
    assert d.max().item() < 0.2 and d.mean().item() < 0.02, (d.max().item(), d.mean().item())
    assert (lg.argmax(1) == lr.argmax(1)).float().mean().item() > 0.975
    assert abs(loss - loss_ref) < 2e-2
    for n, r in gref.items():
        if is_cancelled_bias(n):
            assert g[n].abs().max().item() == 0.0, n
            continue
        a, b = g[n].double().flatten(), r.double().flatten()
        cos = float(a @ b / (a.norm() * b.norm() + 1e-30))
        ratio = float(a.norm() / (b.norm() + 1e-30))
        lo = 0.995 if r.dim() > 1 else 0.98
        assert cos >= lo, (n, cos)
        assert abs(ratio - 1.0) <= (0.05 if r.dim() > 1 else 0.10), (n, ratio)
