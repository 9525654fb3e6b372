"""D2H / H2D rates of the box for page-locked memory (development aid): what the host front-end's transfers can reach."""
import time, torch
dev = torch.device("cuda", 0)
n = 1 << 30
d = torch.empty(n, dtype=torch.uint8, device=dev)
h = torch.empty(n, dtype=torch.uint8).pin_memory()
h2 = torch.empty(n, dtype=torch.uint8).pin_memory()
d2 = torch.empty(n, dtype=torch.uint8, device=dev)
for name, fn in (("D2H 1 stream", lambda: h.copy_(d, non_blocking=True)), ("H2D 1 stream", lambda: d.copy_(h, non_blocking=True))):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(4): fn()
    torch.cuda.synchronize()
    print("%s: %.1f GB/s" % (name, 4 * n / (time.perf_counter() - t) / 1e9))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def two():
    with torch.cuda.stream(s1): h.copy_(d, non_blocking=True)
    with torch.cuda.stream(s2): h2.copy_(d2, non_blocking=True)
two(); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(4): two()
torch.cuda.synchronize()
print("D2H 2 streams: %.1f GB/s" % (8 * n / (time.perf_counter() - t) / 1e9))
def bidir():
    with torch.cuda.stream(s1): h.copy_(d, non_blocking=True)
    with torch.cuda.stream(s2): d2.copy_(h2, non_blocking=True)
bidir(); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(4): bidir()
torch.cuda.synchronize()
print("D2H + H2D together: %.1f GB/s each way" % (4 * n / (time.perf_counter() - t) / 1e9))
