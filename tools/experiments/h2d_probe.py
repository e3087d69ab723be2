#!/usr/bin/env python3
"""What bounds a scan of a file that is in the page cache: the copy out of the page cache into pinned memory (pread, by
a number of threads), or the host -> device copy out of that pinned memory?  Prints GB/s for both, and for the two
overlapped (a reader thread pool filling one pinned buffer while the other is on its way to the device)."""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import torch

size = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
path = "/tmp/qe_h2d_probe.bin"
with open(path, "wb") as f:
    chunk = os.urandom(1 << 24)
    for _ in range(size >> 24):
        f.write(chunk)
fd = os.open(path, os.O_RDONLY)
pinned = [torch.empty(size, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
views = [memoryview(p.numpy()) for p in pinned]
dev = torch.empty(size, dtype=torch.uint8, device="cuda")
SLICE = 4 << 20


def read_into(view, threads):
    def one(i):
        off = i * SLICE
        n = min(SLICE, size - off)
        got = os.preadv(fd, [view[off:off + n]], off)
        assert got == n
    with ThreadPoolExecutor(threads) as pool:
        list(pool.map(one, range((size + SLICE - 1) // SLICE)))


read_into(views[0], 16)  # warm: page cache and the pinned pages
for threads in (1, 2, 4, 8, 16, 32):
    t0 = time.perf_counter()
    for _ in range(3):
        read_into(views[0], threads)
    dt = (time.perf_counter() - t0) / 3
    print(f"pread page cache -> pinned, {threads:2d} threads: {size / dt / 1e9:6.1f} GB/s", flush=True)
torch.cuda.synchronize()
for _ in range(2):
    dev.copy_(pinned[0], non_blocking=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    dev.copy_(pinned[0], non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(f"host -> device out of pinned memory: {size / dt / 1e9:6.1f} GB/s", flush=True)
for threads in (8, 16, 32):
    t0 = time.perf_counter()
    reps = 6
    for i in range(reps):
        read_into(views[i & 1], threads)          # fills one buffer ...
        dev.copy_(pinned[i & 1], non_blocking=True)  # ... while the other one's copy is in flight
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"both overlapped (two pinned buffers), {threads:2d} reader threads: {size / dt / 1e9:6.1f} GB/s", flush=True)
os.close(fd)
os.unlink(path)
