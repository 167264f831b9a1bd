import torch, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_segmentation_amd as seg
from oracle.fill import fill, labels, fill_module
torch.manual_seed(0)
for dt in (torch.bfloat16, torch.float32):
    seg.set_compute_dtype(dt)
    m = seg.unet(3, 3); fill_module(m, 1000); m = m.cuda().train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, fused=True)
    X = fill((4, 3, 128, 128), 1, 0, 1).cuda(); Y = labels((4, 128, 128), 2, 3).cuda()
    lf = seg.CrossEntropyLoss()
    ls = []
    for i in range(40):
        opt.zero_grad(set_to_none=True)
        l = lf(m(X), Y); l.backward(); opt.step(); ls.append(l.item())
    print(dt, [round(v, 4) for v in ls[::5]], ls[-1])
