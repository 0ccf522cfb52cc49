"""Host-to-device bandwidth probe (pinned memory, one or two streams, chunk sizes) - bounds what the from-file path can reach."""
import time, torch
dev = torch.device("cuda:0")
n = 1536 << 20
src = torch.empty(n, dtype=torch.uint8).pin_memory()
src.fill_(7)
dst = torch.empty(n, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
def run(label, fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    print(f"{label}: {best * 1e3:.2f} ms  {n / best / 1e9:.1f} GB/s", flush=True)
run("one copy 1.5 GiB", lambda: dst.copy_(src, non_blocking=True))
for mb in (8, 32, 128):
    c = mb << 20
    def chunks():
        for o in range(0, n, c):
            dst[o:o + c].copy_(src[o:o + c], non_blocking=True)
    run(f"chunks of {mb} MiB, one stream", chunks)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def two():
    c = 32 << 20
    for k, o in enumerate(range(0, n, c)):
        with torch.cuda.stream(s1 if k % 2 == 0 else s2):
            dst[o:o + c].copy_(src[o:o + c], non_blocking=True)
run("chunks of 32 MiB, two streams", two)
# device-to-host
back = torch.empty(n, dtype=torch.uint8).pin_memory()
run("D2H one copy 1.5 GiB", lambda: back.copy_(dst, non_blocking=True))
def both():
    with torch.cuda.stream(s1):
        dst.copy_(src, non_blocking=True)
    with torch.cuda.stream(s2):
        back.copy_(dst, non_blocking=True)
run("H2D and D2H together (1.5 GiB each)", both)
# host memcpy rate into pinned memory with threads (numpy releases the GIL in copyto)
import numpy as np, threading, os
pg = np.ones(n, dtype=np.uint8)
pin = src.numpy()
for th in (1, 4, 8, 15, 30):
    part = n // th
    def work(k):
        np.copyto(pin[k * part:(k + 1) * part], pg[k * part:(k + 1) * part])
    best = 1e9
    for _ in range(3):
        ts = [threading.Thread(target=work, args=(k,)) for k in range(th)]
        t = time.perf_counter(); [x.start() for x in ts]; [x.join() for x in ts]; best = min(best, time.perf_counter() - t)
    print(f"host memcpy pageable -> pinned, {th} threads: {n / best / 1e9:.1f} GB/s", flush=True)
print("cpus", os.cpu_count())
