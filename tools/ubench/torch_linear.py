import torch, time
def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
M=3152
for name,K,N in (("qkv",768,2304),("out",768,768),("fc1",768,3072),("fc2",3072,768)):
    x=torch.randn(M,K,device="cuda",dtype=torch.bfloat16); w=torch.randn(N,K,device="cuda",dtype=torch.bfloat16); b=torch.randn(N,device="cuda",dtype=torch.bfloat16)
    t=timeit(lambda: torch.nn.functional.linear(x,w,b))
    print(f"torch linear {name} {t:7.1f} us {2.0*M*K*N/t/1e6:7.1f} TF/s")
