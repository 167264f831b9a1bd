"""GPU (-m gpu): reference-format checkpoint files round trip (reference utils/training.py:502-544,
564-609).  A file written the way the reference writes it from a stock-PyTorch U-Net (the CPU oracle has the
reference's module tree, hence its state_dict) must load strict=True into the HIP model through start() and give
the same logits; the files start() writes must load strict=True back into the stock model."""
import os

import numpy as np
import pytest
import torch

from oracle.fill import fill, labels, fill_module
from oracle import unet_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def seg():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import image_segmentation_amd as s
    return s


def test_reference_checkpoint_resume_and_save(seg, tmp_path):
    from image_segmentation_amd import training
    training.VERBOSE = False
    seg.set_compute_dtype(torch.float32)
    dev = torch.device("cuda")
    name = "UNet.pytorch"

    # --- a checkpoint as the reference's start() writes it (training.py:564-609), from the stock model
    ref = unet_ref.unet(3, 3); fill_module(ref, 4242); ref.eval()
    ropt = torch.optim.AdamW(ref.parameters(), weight_decay=0.01)
    torch.save({"epoch": 7, "model_state_dict": ref.state_dict(), "optimizer_state_dict": ropt.state_dict(),
                "best_dev_dice": 0.5, "best_dev_miou": 2.0, "best_dev_loss": 0.25, "notes": "reference-format file"},
               tmp_path / name)

    # --- resume: start() loads it strict=True; epochs == saved epoch -> no training, the weights stay as loaded
    m = seg.unet(3, 3).to(dev)
    opt = torch.optim.AdamW(m.parameters(), weight_decay=0.01)
    best = training.start(str(tmp_path), name, m, opt, [], [], 1, dev, seg.CrossEntropyLoss(), seg.CrossEntropyLoss(),
                          target_size=32, num_classes=3, ignore_index=None, epochs=7)
    assert best == (2.0, 0.5, 0.25)                       # best metrics come from the file
    for (ka, va), (kb, vb) in zip(m.state_dict().items(), ref.state_dict().items()):
        assert ka == kb and torch.equal(va.cpu(), vb), ka
    x = fill((2, 3, 32, 32), 5, 0, 1)
    m.eval()
    with torch.no_grad():
        mine = m(x.to(dev)).float().cpu()
        want = ref(x)
    assert (mine - want).abs().max().item() < 1e-3
    assert torch.equal(mine.argmax(1), want.argmax(1))

    # --- train one more epoch (epochs = 8 > 7): a better mIoU than the stored 2.0 is impossible, so force a save by
    # starting from a file whose best mIoU is -inf; then the written files must carry the reference's keys and load
    # strict=True into the stock model
    name2 = "UNet2.pytorch"
    torch.save({"epoch": 0, "model_state_dict": ref.state_dict()}, tmp_path / name2)
    m2 = seg.unet(3, 3).to(dev)
    opt2 = torch.optim.AdamW(m2.parameters(), weight_decay=0.01)
    data = [(fill((2, 3, 32, 32), 10 + i, 0, 1), labels((2, 1, 32, 32), 20 + i, 3)) for i in range(2)]
    from image_segmentation_amd.metrics import MetricsHistory
    agg = MetricsHistory(3, None)
    training.start(str(tmp_path), name2, m2, opt2, data, data, 1, dev, seg.CrossEntropyLoss(), seg.CrossEntropyLoss(),
                   target_size=32, agg=agg, num_classes=3, ignore_index=None, epochs=1)
    ck = torch.load(tmp_path / name2, map_location="cpu", weights_only=True)
    assert {"epoch", "model_state_dict", "optimizer_state_dict", "best_dev_dice", "best_dev_miou", "best_dev_loss",
            "notes"} <= set(ck.keys()) and ck["epoch"] == 1
    mo = torch.load(tmp_path / f"MO_{name2}", map_location="cpu", weights_only=True)
    assert set(mo.keys()) == {"epoch", "model_state_dict"}
    assert os.path.isfile(tmp_path / "metrics" / name2)
    back = unet_ref.unet(3, 3)
    back.load_state_dict(mo["model_state_dict"], strict=True)
    back.eval(); m2.eval()
    with torch.no_grad():
        a = m2(x.to(dev)).float().cpu()
        b = back(x)
    assert (a - b).abs().max().item() < 1e-3 and torch.equal(a.argmax(1), b.argmax(1))
    # training moved the weights and the BatchNorm buffers went through the file
    assert not torch.equal(mo["model_state_dict"]["output.weight"], ref.state_dict()["output.weight"])
    assert int(mo["model_state_dict"]["down1.doubleConvReLU.1.num_batches_tracked"]) == 2
