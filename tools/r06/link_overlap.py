"""Does an upload overlap a download on this box's link?  Pinned host memory, two streams, 64 MiB pieces: each direction alone, then both at once.
python tools/r06/link_overlap.py"""
import time
import torch

dev = torch.device("cuda", 0)
n = 64 << 20
h_up, h_down = torch.empty(n, dtype=torch.uint8, pin_memory=True), torch.empty(n, dtype=torch.uint8, pin_memory=True)
d_up, d_down = torch.empty(n, dtype=torch.uint8, device=dev), torch.empty(n, dtype=torch.uint8, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
reps = 20


def run(up, down):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        if up:
            with torch.cuda.stream(s1):
                d_up.copy_(h_up, non_blocking=True)
        if down:
            with torch.cuda.stream(s2):
                h_down.copy_(d_down, non_blocking=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for _ in range(2):
    u, d, b = run(True, False), run(False, True), run(True, True)
    print("64 MiB up alone %.2f ms (%.1f GiB/s), down alone %.2f ms (%.1f GiB/s), both at once %.2f ms -- the sum would be %.2f, the longer one %.2f" % (
        u * 1e3, n / u / 2**30, d * 1e3, n / d / 2**30, b * 1e3, (u + d) * 1e3, max(u, d) * 1e3))
