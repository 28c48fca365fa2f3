"""How many kernels of different HIP streams really run at once: S streams x one spin kernel (torch.cuda._sleep, one thread) each,
wall time of the batch against one kernel's.  usage: [GPU_MAX_HW_QUEUES=n] stream_concurrency_probe.py"""
import os, time, torch
cyc = 20_000_000
torch.cuda._sleep(1000); torch.cuda.synchronize()
t0 = time.perf_counter(); torch.cuda._sleep(cyc); torch.cuda.synchronize(); one = time.perf_counter() - t0
print("GPU_MAX_HW_QUEUES=%s  one spin kernel %.2f ms" % (os.environ.get("GPU_MAX_HW_QUEUES", "default"), one * 1e3))
for S in (2, 3, 4, 5, 6, 8, 12):
    streams = [torch.cuda.Stream() for _ in range(S)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for st in streams:
        with torch.cuda.stream(st):
            torch.cuda._sleep(cyc)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("  %2d streams: %.2f ms = %.2f x one kernel -> ~%d at a time" % (S, dt * 1e3, dt / one, round(S * one / dt)))
