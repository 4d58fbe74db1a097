import torch, time, threading
n = 1600 << 20
h = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(3)]
d = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(3)]
s = [torch.cuda.Stream() for _ in range(3)]
def one(k, reps=4):
    with torch.cuda.stream(s[k]):
        for _ in range(reps):
            d[k].copy_(h[k], non_blocking=True)
for k in range(3): one(k, 1)
torch.cuda.synchronize()
for lanes in (1, 2, 3):
    t = time.time()
    for k in range(lanes): one(k)
    torch.cuda.synchronize()
    dt = time.time() - t
    print("H2D", lanes, "streams:", round(lanes * 4 * n / dt / 1e9, 1), "GB/s")
def back(k, reps=4):
    with torch.cuda.stream(s[k]):
        for _ in range(reps):
            h[k].copy_(d[k], non_blocking=True)
for lanes in (1, 2):
    t = time.time()
    for k in range(lanes): back(k)
    torch.cuda.synchronize()
    dt = time.time() - t
    print("D2H", lanes, "streams:", round(lanes * 4 * n / dt / 1e9, 1), "GB/s")
t = time.time(); one(0); back(1); torch.cuda.synchronize(); dt = time.time() - t
print("H2D + D2H together:", round(2 * 4 * n / dt / 1e9, 1), "GB/s total")
# 8 MB pieces
with torch.cuda.stream(s[0]):
    t = time.time()
    for r in range(2):
        for o in range(0, n, 8 << 20):
            d[0][o:o + (8 << 20)].copy_(h[0][o:o + (8 << 20)], non_blocking=True)
    torch.cuda.synchronize(); dt = time.time() - t
print("H2D in 8 MB pieces:", round(2 * n / dt / 1e9, 1), "GB/s")
