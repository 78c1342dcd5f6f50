import os, time, subprocess, torch
print("affinity", sorted(os.sched_getaffinity(0)))
print(subprocess.run("lscpu | grep -i 'numa\|socket\|model name\|^CPU(s)'", shell=True, capture_output=True, text=True).stdout)
def h2d(tag):
    x = torch.empty(400 << 20, dtype=torch.uint8, pin_memory=True)
    x.fill_(1)
    d = torch.empty_like(x, device="cuda")
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); d.copy_(x, non_blocking=True); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    src = torch.empty(400 << 20, dtype=torch.uint8); src.fill_(2)
    t0 = time.perf_counter(); x.copy_(src); t1 = time.perf_counter() - t0
    print(tag, "cpu", os.sched_getcpu() if hasattr(os, "sched_getcpu") else "?", "H2D GB/s", 0.4194 / min(ts), "host copy GB/s (1 thread torch)", 0.4194 / t1)
h2d("fresh")
torch.set_num_threads(16)
a = torch.randn(4096, 4096)
for _ in range(20): b = a @ a
h2d("after 16-thread matmuls")
time.sleep(1.0)
h2d("after 1 s idle")
