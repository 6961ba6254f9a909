"""Run the batched CMUX `reps` times on the same inputs and count the units whose output differs from the first run's."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, spf_amd
B = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
P = spf_amd.DEFAULT_128
dev = torch.device("cuda", 0)
eng = spf_amd.Engine(P, device=0)
g = torch.Generator(device=dev).manual_seed(5)
gg = torch.randn((B, P.cbs_ggsw_complex * 2), generator=g, device=dev, dtype=torch.float64) * (2.0 ** 60)
da = torch.randint(-(2 ** 63), 2 ** 63 - 1, (B, P.glwe_words), generator=g, device=dev, dtype=torch.int64)
db = torch.randint(-(2 ** 63), 2 ** 63 - 1, (B, P.glwe_words), generator=g, device=dev, dtype=torch.int64)
st = torch.cuda.current_stream().cuda_stream
outs = []
for r in range(reps):
    dc = torch.zeros_like(da)
    eng.cmux_dev(st, B, gg.data_ptr(), da.data_ptr(), db.data_ptr(), dc.data_ptr())
    torch.cuda.synchronize()
    outs.append(dc)
bad = [int((o != outs[0]).any(dim=1).sum().item()) for o in outs[1:]]
print("persist", os.environ.get("SPF_CMUX_PERSIST"), "B", B, "units differing from run 0, per run:", bad)
