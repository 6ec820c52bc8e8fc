import time, numpy as np, torch, os
from concurrent.futures import ThreadPoolExecutor
print("cpus", os.cpu_count(), len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads())
n = 1 << 20
t0 = time.perf_counter(); hp = torch.empty((4 * n, 39), dtype=torch.float32, pin_memory=True); t1 = time.perf_counter()
print(f"pinned alloc {hp.numel()*4/1e6:.0f} MB: {(t1-t0)*1e3:.1f} ms")
t0 = time.perf_counter(); hp2 = torch.empty((n, 27), dtype=torch.float64, pin_memory=True); t1 = time.perf_counter()
print(f"pinned alloc {hp2.numel()*8/1e6:.0f} MB: {(t1-t0)*1e3:.1f} ms")
src = np.random.rand(n, 27)
dst = hp2.numpy()
for th in (1, 2, 4, 8, 16):
    pool = ThreadPoolExecutor(th)
    def cp(k):
        lo, hi = k * n // th, (k + 1) * n // th
        np.copyto(dst[lo:hi], src[lo:hi])
    list(pool.map(cp, range(th)))
    t0 = time.perf_counter()
    for _ in range(5): list(pool.map(cp, range(th)))
    dt = (time.perf_counter() - t0) / 5
    print(f"memcpy 216 MB pageable->pinned, {th} threads: {dt*1e3:.1f} ms = {src.nbytes/dt/1e9:.1f} GB/s")
    pool.shutdown()
d = torch.empty((n, 27), dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
for name, f in (("H2D pinned", lambda: d.copy_(hp2, non_blocking=True)), ("D2H pinned", lambda: hp2.copy_(d, non_blocking=True)),
                ("H2D pageable", lambda: d.copy_(torch.from_numpy(src)))):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"{name} 216 MB: {dt*1e3:.1f} ms = {src.nbytes/dt/1e9:.1f} GB/s")
pg = np.empty((n, 27))
t0 = time.perf_counter(); torch.from_numpy(pg).copy_(d); torch.cuda.synchronize(); print("D2H pageable first", (time.perf_counter()-t0)*1e3)
t0 = time.perf_counter(); torch.from_numpy(pg).copy_(d); torch.cuda.synchronize(); print("D2H pageable", (time.perf_counter()-t0)*1e3, "ms")
