import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, spf_amd, numpy as np
B = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
P = spf_amd.DEFAULT_128
dev = torch.device("cuda", 0)
eng = spf_amd.Engine(P, device=0)
g = torch.Generator(device=dev).manual_seed(5)
gg = torch.randn((B, P.cbs_ggsw_complex * 2), generator=g, device=dev, dtype=torch.float64) * (2.0 ** 60)
da = torch.randint(-(2 ** 63), 2 ** 63 - 1, (B, P.glwe_words), generator=g, device=dev, dtype=torch.int64)
db = torch.randint(-(2 ** 63), 2 ** 63 - 1, (B, P.glwe_words), generator=g, device=dev, dtype=torch.int64)
st = torch.cuda.current_stream().cuda_stream
ref = None
os.environ["X"] = "1"
for r in range(reps):
    dc = torch.zeros_like(da)
    eng.cmux_dev(st, B, gg.data_ptr(), da.data_ptr(), db.data_ptr(), dc.data_ptr())
    torch.cuda.synchronize()
    if ref is None:
        ref = dc
        continue
    bad = torch.nonzero((dc != ref).any(dim=1)).flatten().tolist()
    for i in bad[:3]:
        w = torch.nonzero(dc[i] != ref[i]).flatten().tolist()
        j = w[0]
        delta = (int(dc[i, j]) - int(ref[i, j])) & ((1 << 64) - 1)
        cands = {}
        for name, t in (("d0", da), ("d1", db)):
            for back in (1024, 2048, -1024):
                k = i - back
                if 0 <= k < B:
                    cands[f"{name}[{k}]-{name}[{i}]"] = (int(t[k, j]) - int(t[i, j])) & ((1 << 64) - 1)
        cands["-d0"] = (-int(da[i, j])) & ((1 << 64) - 1)
        cands["out_ref[i-1024]-out_ref[i]"] = (int(ref[i - 1024, j]) - int(ref[i, j])) & ((1 << 64) - 1) if i >= 1024 else None
        hit = [n for n, v in cands.items() if v == delta]
        print(f"run {r}: unit {i} words {len(w)} first {j}: delta {delta:#x} matches {hit}")
