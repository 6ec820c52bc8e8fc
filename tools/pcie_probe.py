"""Host <-> device copy rates for the chunk sizes generate_lut moves (pageable straight copies against a pinned bounce buffer filled by
host threads): decides whether a bounce-buffer pipeline would pay.    python tools/pcie_probe.py"""
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

dev = torch.device("cuda:0")
rows, n = 1 << 18, 16
src = np.random.rand(rows * n, 27)                      # 905 MB pageable
dst = torch.empty((rows, 27), dtype=torch.float64, device=dev)
t = lambda f: (torch.cuda.synchronize(), time.perf_counter(), f(), torch.cuda.synchronize(), time.perf_counter())   # noqa: E731


def rate(name, f, nbytes):
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name:62s} {nbytes / dt / 1e9:6.1f} GB/s  ({dt * 1e3:.1f} ms)", flush=True)


nb = src.nbytes
rate("H2D pageable, chunk by chunk", lambda: [dst.copy_(torch.from_numpy(src[i * rows:(i + 1) * rows])) for i in range(n)], nb)
pin = [torch.empty((rows, 27), dtype=torch.float64).pin_memory() for _ in range(2)]
pin_np = [p.numpy() for p in pin]


def bounce(threads):
    pool = ThreadPoolExecutor(threads)
    s = torch.cuda.Stream()
    evs = [torch.cuda.Event(), torch.cuda.Event()]

    def fill(j, i):
        per = -(-rows // threads)
        list(pool.map(lambda k: np.copyto(pin_np[j][k * per:(k + 1) * per], src[i * rows + k * per:i * rows + min(rows, (k + 1) * per)]), range(threads)))

    def run():
        for i in range(n):
            j = i % 2
            if i >= 2:
                evs[j].synchronize()
            fill(j, i)
            with torch.cuda.stream(s):
                dst.copy_(pin[j], non_blocking=True)
                evs[j].record(s)
        s.synchronize()
    return run


for th in (1, 2, 4, 8):
    rate(f"H2D via 2 pinned bounce buffers, {th} host threads filling", bounce(th), nb)
rate("H2D from pinned only (no fill)", lambda: [dst.copy_(pin[i % 2], non_blocking=True) for i in range(n)], nb)
out = np.empty((rows * n, 13), dtype=np.float32)
dsrc = torch.rand((rows, 13), dtype=torch.float32, device=dev)
rate("D2H pageable, chunk by chunk", lambda: [torch.from_numpy(out[i * rows:(i + 1) * rows]).copy_(dsrc) for i in range(n)], out.nbytes)
pin2 = torch.empty((rows, 13), dtype=torch.float32).pin_memory()
rate("D2H into pinned only", lambda: [pin2.copy_(dsrc, non_blocking=True) for i in range(n)], out.nbytes)
a = np.random.rand(rows, 27)
b = np.empty_like(a)
t0 = time.perf_counter()
for _ in range(20):
    np.copyto(b, a)
print(f"host memcpy, one thread: {20 * a.nbytes / (time.perf_counter() - t0) / 1e9:.1f} GB/s")

# both directions at once: pageable copies issued from two host threads on two streams (what generate_lut does) ...
import threading
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def up():
    with torch.cuda.stream(s1):
        for i in range(n):
            dst.copy_(torch.from_numpy(src[i * rows:(i + 1) * rows]), non_blocking=True)
        s1.synchronize()


def down():
    with torch.cuda.stream(s2):
        for r in range(4):
            for i in range(n):
                torch.from_numpy(out[i * rows:(i + 1) * rows]).copy_(dsrc, non_blocking=True)
        s2.synchronize()


def both(fa, fb):
    def run():
        ta, tb = threading.Thread(target=fa), threading.Thread(target=fb)
        ta.start(); tb.start(); ta.join(); tb.join()
    return run


rate("H2D pageable alone (905 MB)", up, nb)
rate("D2H pageable alone (4 x 218 MB)", down, 4 * out.nbytes)
rate("both pageable, two threads / streams (sum of bytes)", both(up, down), nb + 4 * out.nbytes)
pin_out = [torch.empty((rows, 13), dtype=torch.float32).pin_memory() for _ in range(2)]


def up_pinned():
    bounce(4)()


def down_pinned():
    ev = [torch.cuda.Event(), torch.cuda.Event()]
    with torch.cuda.stream(s2):
        for r in range(4):
            for i in range(n):
                j = i % 2
                pin_out[j].copy_(dsrc, non_blocking=True)
                ev[j].record(s2)
                ev[j].synchronize()
                np.copyto(out[i * rows:(i + 1) * rows], pin_out[j].numpy())
        s2.synchronize()


rate("D2H via pinned + host memcpy alone", down_pinned, 4 * out.nbytes)
rate("both via pinned bounce buffers (sum of bytes)", both(up_pinned, down_pinned), nb + 4 * out.nbytes)
