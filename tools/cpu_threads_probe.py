import os, time, torch, torch.nn.functional as F
x = torch.randn(1, 256, 224, 224); w = torch.randn(256, 256, 3, 3)
for n in (8, 16, 32, 64, 128, 256):
    torch.set_num_threads(n)
    F.conv2d(x, w, padding=1)
    t0 = time.perf_counter(); F.conv2d(x, w, padding=1); F.conv2d(x, w, padding=1); dt = (time.perf_counter() - t0) / 2
    print(n, f"{2*224*224*256*256*9/dt/1e9:.0f} GFLOP/s", flush=True)
print(os.cpu_count(), len(os.sched_getaffinity(0)))
