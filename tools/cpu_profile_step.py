"""cProfile of the host side of the U-Net training step (where the 9 ms of enqueue time go)."""
import cProfile, pstats, sys, torch
sys.path.insert(0, '.')
import image_segmentation_amd as seg
import bench
seg.set_compute_dtype(torch.bfloat16)
m = seg.unet(3, 3).cuda().train()
opt = torch.optim.AdamW(m.parameters(), weight_decay=0.01, fused=True)
X = bench.fill((32, 3, 256, 256), 1, 0, 1).cuda(); Y = bench.labels((32, 256, 256), 2, 3).cuda()
lf = seg.CrossEntropyLoss()
def step():
    opt.zero_grad(set_to_none=True); l = lf(m(X), Y); l.backward(); opt.step()
for _ in range(10): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
