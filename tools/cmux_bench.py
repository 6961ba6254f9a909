"""Time the batched cbs_radix CMUX kernel alone (the `cmux` leg of bench.py): usage python tools/cmux_bench.py [B]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spf_amd

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
P = spf_amd.DEFAULT_128
dev = torch.device("cuda", 0)
eng = spf_amd.Engine(P, device=0)
g = torch.Generator(device=dev).manual_seed(5)
gg = torch.randn((B, P.cbs_ggsw_complex * 2), generator=g, device=dev, dtype=torch.float64) * (2.0 ** 60)
da = torch.randint(-(2 ** 63), 2 ** 63 - 1, (B, P.glwe_words), generator=g, device=dev, dtype=torch.int64)
db = torch.randint(-(2 ** 63), 2 ** 63 - 1, (B, P.glwe_words), generator=g, device=dev, dtype=torch.int64)
dc = torch.empty_like(da)
stream = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    eng.cmux_dev(stream, B, gg.data_ptr(), da.data_ptr(), db.data_ptr(), dc.data_ptr())
torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        eng.cmux_dev(stream, B, gg.data_ptr(), da.data_ptr(), db.data_ptr(), dc.data_ptr())
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 10)
by = B * (P.cbs_ggsw_complex * 16 + 3 * P.glwe_words * 8)
print(json.dumps({"B": B, "kernel_ms": round(best, 4), "cmux_per_s": round(B / best * 1e3, 1),
                  "GBs": round(by / best / 1e6, 1), "hbm_frac": round(by / best / 1e6 / 8000.0, 4),
                  "checksum": int(dc.sum().item())}))
