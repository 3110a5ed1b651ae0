import time, torch
dev=torch.device("cuda:0")
side=torch.cuda.Stream()
def t(n, rec, keep=0):
    xs=[]
    torch.cuda.synchronize()
    t0=time.perf_counter()
    for i in range(200):
        x=torch.empty(n, device=dev, dtype=torch.bfloat16)
        if rec: x.record_stream(side)
        xs.append(x)
        if len(xs)>keep: xs.pop(0)
    t1=time.perf_counter()
    return (t1-t0)/200*1e6
for n in (1<<16, 1<<22, 1<<26, 100_000_000):
    for rec in (0,1):
        for keep in (0,3):
            t(n,rec,keep)
            print("elems %10d record_stream %d keep %d: %.1f us per empty(+free)" % (n, rec, keep, t(n,rec,keep)))
