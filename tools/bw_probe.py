import torch, time
x = torch.empty(1<<29, dtype=torch.float32, device="cuda")  # 2 GB
y = torch.empty_like(x)
for name, fn, nbytes in [("copy", lambda: y.copy_(x), 2*x.numel()*4), ("fill", lambda: y.fill_(1.0), x.numel()*4), ("sum", lambda: x.sum(), x.numel()*4)]:
    fn(); torch.cuda.synchronize()
    t=time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize()
    dt=(time.perf_counter()-t)/10
    print(name, "%.2f ms  %.2f TB/s" % (dt*1e3, nbytes/dt/1e12))
# strided row reads: 2048x2048 int32 matrices, read lower-left quadrant rows (128B pieces)
S = torch.zeros(275, 2048, 2048, dtype=torch.int32, device="cuda")
q = S[:, :1024, :1024]
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(5): z = q.sum()
torch.cuda.synchronize(); dt=(time.perf_counter()-t)/5
print("quadrant sum %.2f ms  %.2f TB/s" % (dt*1e3, 275*1024*1024*4/dt/1e12))
