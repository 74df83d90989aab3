"""Which torch streams share a hardware queue with the main stream / with each other (a busy stream then holds up the other):
launches a ~15 ms sleep kernel on one stream and times a tiny op on another.  GPU box."""
import time, torch
dev = torch.device("cuda:0")
main = torch.cuda.current_stream(dev)
x = torch.zeros(1, device=dev); torch.cuda.synchronize()
cands = [torch.cuda.Stream(device=dev) for _ in range(10)]
def blocks(a, b):
    """does a busy stream a hold up a tiny op on stream b?"""
    torch.cuda.synchronize()
    with torch.cuda.stream(a): torch.cuda._sleep(30_000_000)
    t0 = time.perf_counter()
    with torch.cuda.stream(b): y = x + 1
    b.synchronize(); dt = time.perf_counter() - t0
    a.synchronize()
    return dt > 3e-3, dt
for i, st in enumerate(cands):
    print("cand", i, "blocks main:", blocks(st, main), " main blocks cand:", blocks(main, st)[0])
print("pairs that share:", [(i, j) for i in range(6) for j in range(i + 1, 6) if blocks(cands[i], cands[j])[0]])
