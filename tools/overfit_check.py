"""Builder convenience: 40 AdamW steps on ONE fixed random batch through the HIP modules, bf16 and fp32 side by side --
the loss curves must fall together (a stale packed weight, a wrong gradient or a broken fused path shows here at once)."""
import torch, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_segmentation_amd as seg
torch.manual_seed(0)
for dt in (torch.bfloat16, torch.float32):
    seg.set_compute_dtype(dt)
    torch.manual_seed(1000); m = seg.unet(3, 3).cuda().train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, fused=True)
    g = torch.Generator().manual_seed(1); X = torch.rand((4, 3, 128, 128), generator=g).cuda(); Y = torch.randint(0, 3, (4, 128, 128), generator=g).cuda()
    lf = seg.CrossEntropyLoss()
    ls = []
    for i in range(40):
        opt.zero_grad(set_to_none=True)
        l = lf(m(X), Y); l.backward(); opt.step(); ls.append(l.item())
    print(dt, [round(v, 4) for v in ls[::5]], ls[-1])
