import torch, time
d=torch.device("cuda",0)
a=torch.zeros(1<<20,dtype=torch.int64,device=d)      # 8 MB
h=torch.zeros(1<<20,dtype=torch.int64).pin_memory()
big=torch.zeros(1<<28,device=d)
s=torch.cuda.Stream(device=d)
def busy():
    for _ in range(20): big.mul_(1.0001)
    torch.cuda.synchronize()
for idle_ms in (0,1,3,7,15,40):
    res=[]
    for rep in range(5):
        busy()
        t=time.perf_counter()
        while (time.perf_counter()-t)*1e3 < idle_ms: pass
        t0=time.perf_counter()
        with torch.cuda.stream(s):
            h.copy_(a,non_blocking=True)
        s.synchronize()
        res.append((time.perf_counter()-t0)*1e3)
    print("idle %2d ms -> d2h 8MB: %s ms"%(idle_ms," ".join("%.2f"%x for x in res)))
