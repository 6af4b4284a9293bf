import torch, time
x = torch.empty(4 << 30, dtype=torch.uint8, device='cuda')
y = torch.empty(4 << 30, dtype=torch.uint8, device='cuda')
for name, f, nbytes in (("memset (write only)", lambda: x.zero_(), 4 << 30), ("copy (read + write)", lambda: y.copy_(x), 8 << 30), ("sum (read only)", lambda: x.view(torch.int64).sum(), 4 << 30)):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"{name:24s} {nbytes / dt / 1e12:.2f} TB/s")
