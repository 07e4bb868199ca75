"""dev tool (round 5): does a device -> host copy issued after a few ms of GPU idleness stall?  (bench: the first copy of the consensus call, 5 ms after the clustering call,
sometimes waits 15 - 25 ms on some boxes of the pool.)  8 MB device -> pinned host copies after sleeps of 0 .. 50 ms, with and without a busy kernel in between."""
import time, torch
dev = torch.device("cuda", 0)
src = torch.zeros(8 << 20, dtype=torch.uint8, device=dev); dst = torch.empty(8 << 20, dtype=torch.uint8).pin_memory()
big = torch.zeros(64 << 20, dtype=torch.float32, device=dev)
s = torch.cuda.Stream()
def copy_ms():
    t = time.perf_counter()
    with torch.cuda.stream(s):
        dst.copy_(src, non_blocking=True)
    s.synchronize()
    return (time.perf_counter() - t) * 1e3
for _ in range(5): copy_ms()
for rep in range(3):
    for idle in (0, 1, 2, 5, 10, 20, 50):
        r = []
        for _ in range(8):
            with torch.cuda.stream(s):
                for _ in range(20): big.mul_(1.0001)          # ~ a few ms of compute
            s.synchronize()
            time.sleep(idle / 1e3)
            r.append(copy_ms())
        print("idle %2d ms before the copy: copy+sync ms  min %.2f  median %.2f  max %.2f" % (idle, min(r), sorted(r)[len(r) // 2], max(r)), flush=True)
