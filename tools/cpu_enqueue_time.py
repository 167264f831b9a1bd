import time, torch, sys
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
t0 = time.perf_counter()
for _ in range(30): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"30 steps: enqueue {1e3*(t1-t0)/30:.2f} ms/step, total {1e3*(t2-t0)/30:.2f} ms/step")
# a short burst from an idle queue: no back-pressure from the launch queue
best = 1e9
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): step()
    best = min(best, (time.perf_counter() - t0) / 3)
    torch.cuda.synchronize()
print(f"3-step bursts from an idle queue: enqueue {1e3*best:.2f} ms/step")
