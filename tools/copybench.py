import time, numpy as np, torch
dev = torch.device("cuda", 0)
n = 134217728
d = torch.empty(n, dtype=torch.uint8, device=dev)
h_page = np.empty(n, dtype=np.uint8); h_page[:] = 1
h_pin = torch.empty(n, dtype=torch.uint8).pin_memory()
hp = torch.from_numpy(h_page)
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
for name, f in [("D2H pageable", lambda: hp.copy_(d)), ("H2D pageable", lambda: d.copy_(hp)),
                ("D2H pinned", lambda: h_pin.copy_(d, non_blocking=True)), ("H2D pinned", lambda: d.copy_(h_pin, non_blocking=True)),
                ("host memcpy pinned->pageable", lambda: hp.copy_(h_pin)), ("host memcpy pageable->pinned", lambda: h_pin.copy_(hp))]:
    s = t(f)
    print(f"{name:32s} {s*1e3:8.2f} ms  {n/s/1e9:6.1f} GB/s")
import ctypes
t0 = time.perf_counter(); r = torch.cuda.cudart().cudaHostRegister(h_page.ctypes.data, n, 0); t1 = time.perf_counter()
print("hostRegister 128 MiB:", r, f"{(t1-t0)*1e3:.2f} ms")
s = t(lambda: hp.copy_(d)); print(f"D2H registered {s*1e3:.2f} ms {n/s/1e9:.1f} GB/s")
t0 = time.perf_counter(); torch.cuda.cudart().cudaHostUnregister(h_page.ctypes.data); print("unregister", f"{(time.perf_counter()-t0)*1e3:.2f} ms")
print("cpus", __import__("os").cpu_count())
