import os, sys, time, json
if os.environ.get("SPF_PROBE_CPUS"):
    os.sched_setaffinity(0, set(range(int(os.environ["SPF_PROBE_CPUS"]))))   # before any thread exists: all of them inherit it
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import spf_amd
import bench
P = spf_amd.DEFAULT_128
dev = torch.device("cuda", 0)
eng = spf_amd.Engine(P, device=0)
from spf_amd.sharding import key_blob_tensors, replicate_keys
blobs = key_blob_tensors(eng, dev)
g = torch.Generator(device=dev); g.manual_seed(1)
blobs[0].copy_((torch.randn(P.bsk_complex * 2, generator=g, device=dev, dtype=torch.float64) * 2.0**67).view(torch.uint8))
blobs[1].copy_(torch.randint(-(2**63), 2**63-1, (P.ksk_words,), generator=g, device=dev, dtype=torch.int64).view(torch.uint8))
for w in (2, 3):
    n64 = blobs[w].numel() // 8
    blobs[w].copy_((torch.randn(n64, generator=g, device=dev, dtype=torch.float64) * 2.0**67).view(torch.uint8))
replicate_keys(eng, blobs, None, src=0)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
import resource
def cg():
    try:
        return dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat").read().strip().splitlines())
    except Exception as e:
        return {}
r0, c0, t0 = resource.getrusage(resource.RUSAGE_SELF), cg(), time.time()
print(json.dumps(bench._bench_evaluation_pool(eng, P, dev, torch, thread_counts=(T,), seconds=1.0)))
r1, c1, t1 = resource.getrusage(resource.RUSAGE_SELF), cg(), time.time()
print("wall", round(t1 - t0, 2), "user", round(r1.ru_utime - r0.ru_utime, 2), "sys", round(r1.ru_stime - r0.ru_stime, 2),
      "nvcsw", r1.ru_nvcsw - r0.ru_nvcsw, "nivcsw", r1.ru_nivcsw - r0.ru_nivcsw, "minflt", r1.ru_minflt - r0.ru_minflt, "majflt", r1.ru_majflt - r0.ru_majflt)
print({k: int(c1[k]) - int(c0[k]) for k in c1 if k in c0})
