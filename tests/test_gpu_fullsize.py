"""GPU (-m gpu): parity at the image sizes the benchmark numbers are taken on (BASELINE configs 2 and 5): the U-Net at
1x3x256x256 and 1x3x512x512, forward + loss + backward, fp32 parity mode AND bf16 (the mode of the headline number),
against the CPU oracle (= the reference algorithm, reference unet/unet.py:93-105, utils/weighted_loss.py:140-166) on the
same seeded inputs and weights.  These shapes reach the 8x32 tile walk with tiles_x > 8, the register-stationary 256/512
levels and the multi-tile persistent loops that the small-image tests cannot.

Per-parameter checks instead of norm-only ones: fp32 mode relative L2 (<= 2e-2 at batch 1), bf16 cosine and norm ratio
per tensor plus the cosine of the whole gradient (a permuted, sign-flipped or 10 %-off gradient fails both)."""
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
    rels = {}
    for n, r in gref.items():
        if is_cancelled_bias(n):
            assert g[n].abs().max().item() == 0.0, n                  # exact zeros where the reference holds fp32 noise
            continue
        rels[n] = (g[n] - r).norm().item() / max(r.norm().item(), 1e-12)
    worst = sorted(rels.items(), key=lambda kv: -kv[1])[:5]
    print("fp32 mode, worst per-parameter relative L2:", worst)
    # With ONE image the batch statistics of the deep levels rest on 256..1024 values and every gradient is a sum of
    # terms that cancel almost completely behind the batch-statistics BatchNorm: fp32 summation order alone (oneDNN vs
    # the tiled kernels) moves the per-parameter gradients by up to 1e-2 relative (measured: 4e-3 at the first block,
    # 1e-2 at down5; 2e-3 is the bound at B = 4, 128 x 128 in test_gpu_modules.py).  2e-2 still fails a permuted,
    # sign-flipped or 10 %-off gradient by a wide margin.
    for n, rel in rels.items():
        assert rel <= 2e-2, (n, rel, worst)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("S,loss_name", [(256, "ce"), (512, "ce"), (512, "dicece")])
def test_unet_bf16_mode_full_size(seg, S, loss_name):
    """bf16 has no reference counterpart (the reference is fp32 only): gates are logits within bf16 noise of 23 conv+BN
    layers, >= 97.5 % argmax agreement, loss within 2e-2, and the gradient gates stated at the end."""
    _, _, lr, loss_ref, gref = oracle_run(S, loss_name)
    lg, loss, g = hip_run(seg, torch.bfloat16, S, loss_name)
    d = (lg - lr).abs()
    assert d.max().item() < 0.2 and d.mean().item() < 0.02, (d.max().item(), d.mean().item())
    assert (lg.argmax(1) == lr.argmax(1)).float().mean().item() > 0.975
    assert abs(loss - loss_ref) < 2e-2
    stats = {}
    for n, r in gref.items():
        if is_cancelled_bias(n):
            assert g[n].abs().max().item() == 0.0, n
            continue
        a, b = g[n].double().flatten(), r.double().flatten()
        stats[n] = (float(a @ b / (a.norm() * b.norm() + 1e-30)), float(a.norm() / (b.norm() + 1e-30)), r.dim())
    worst = sorted(stats.items(), key=lambda kv: kv[1][0])[:6]
    allg = torch.cat([g[n].double().flatten() for n in stats]); allr = torch.cat([gref[n].double().flatten() for n in stats])
    gcos = float(allg @ allr / (allg.norm() * allr.norm()))
    print(f"bf16 mode S={S} {loss_name}: whole-gradient cosine {gcos:.4f}; lowest per-parameter (cos, norm ratio, dim):", worst)
    # ONE image: the deep levels' batch statistics rest on 256..1024 values and every gradient is an almost completely
    # cancelling sum, so bf16 storage of z / dz (relative 2^-9 per element, not cancelling) shows up as direction noise
    # -- measured per-parameter cosines 0.83..0.99 with norm ratios within 2 % (weights) -- while the same tensors agree
    # to 1e-2 relative in fp32 mode (test above: the kernels' arithmetic is right).  Gates: the gradient as a whole
    # (cosine >= 0.95; measured 0.97..0.998), every weight tensor cosine >= 0.8 and norm within 6 %, every vector cosine >= 0.75 and norm
    # within 25 %: a sign flip (-1), a permutation (~0) or a 10 % scale error on a weight fails.
    assert gcos >= 0.95, gcos
    for n, (cos, ratio, dim) in stats.items():
        assert cos >= (0.8 if dim > 1 else 0.75), (n, cos, worst)
        assert abs(ratio - 1.0) <= (0.06 if dim > 1 else 0.25), (n, ratio, worst)
