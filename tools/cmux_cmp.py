import os, sys, subprocess, json
sys.path.insert(0, os.getcwd())
if len(sys.argv) > 2:
    import torch, spf_amd, numpy as np
    B = int(sys.argv[1])
    P = spf_amd.DEFAULT_128
    dev = torch.device("cuda", 0)
    eng = spf_amd.Engine(P, device=0)
    g = torch.Generator(device=dev).manual_seed(5)
    gg = torch.randn((B, P.cbs_ggsw_complex * 2), generator=g, device=dev, dtype=torch.float64) * (2.0 ** 60)
    da = torch.randint(-(2 ** 63), 2 ** 63 - 1, (B, P.glwe_words), generator=g, device=dev, dtype=torch.int64)
    db = torch.randint(-(2 ** 63), 2 ** 63 - 1, (B, P.glwe_words), generator=g, device=dev, dtype=torch.int64)
    dc = torch.zeros_like(da)
    eng.cmux_dev(torch.cuda.current_stream().cuda_stream, B, gg.data_ptr(), da.data_ptr(), db.data_ptr(), dc.data_ptr())
    torch.cuda.synchronize()
    np.save(sys.argv[2], dc.cpu().numpy())
    if len(sys.argv) > 3:   # oracle check of the listed units
        import oracle as O
        OP = O.DEFAULT_128
        for i in [int(x) for x in sys.argv[3].split(",")]:
            exp = O.cmux(da[i].cpu().numpy().view(np.uint64), db[i].cpu().numpy().view(np.uint64), gg[i].cpu().numpy().view(np.complex128), OP.N, OP.k, OP.cbs_radix_log, OP.cbs_count)
            got = dc[i].cpu().numpy().view(np.uint64)
            print("unit", i, "persist", os.environ.get("SPF_CMUX_PERSIST"), "equals oracle:", bool(np.array_equal(got, exp)), "words off:", int((got != exp).sum()))
else:
    import numpy as np
    B = sys.argv[1]
    for p in ("0", "1"):
        subprocess.run([sys.executable, __file__, B, f"/tmp/cm_{p}.npy"], env=dict(os.environ, SPF_CMUX_PERSIST=p), check=True)
    a, b = np.load("/tmp/cm_0.npy"), np.load("/tmp/cm_1.npy")
    bad = np.nonzero((a != b).any(axis=1))[0]
    if bad.size:
        units = ",".join(str(int(x)) for x in bad[:4])
        for p in ("0", "1"):
            subprocess.run([sys.executable, __file__, B, f"/tmp/cm_x.npy", units], env=dict(os.environ, SPF_CMUX_PERSIST=p), check=True)
    print("B", B, "bad units", bad.size, bad[:16], "words differing in first bad unit:", int((a[bad[0]] != b[bad[0]]).sum()) if bad.size else 0,
          "which words", np.nonzero(a[bad[0]] != b[bad[0]])[0][:12] if bad.size else "")
