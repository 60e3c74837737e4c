import importlib, json, sys, os
import torch
sys.path.insert(0, "/root/repo")
ofx = importlib.import_module("optical-flow-1_amd")
synth = importlib.import_module("optical-flow-1_amd.synth")
dev = torch.device("cuda:0")
ctx = ofx.Ofx(0, ofx.F64)
for nx, ny, Gs in ((1920, 1080, (1, 2, 4, 8, 16)), (3840, 2160, (1, 4, 16))):
    for G in Gs:
        I0, I1, out = [], [], []
        for k in range(G):
            a, b = synth.pair_device("P1", nx, ny, k, dev, torch.float64)
            I0.append(a); I1.append(b); out.append(torch.empty((ny, nx, 2), dtype=torch.float32, device=dev))
        ctx.set_option("concurrency", 1); ctx.set_option("profile", 1); ctx.set_option("fixed_work", 1)
        args = ([t.data_ptr() for t in I0], [t.data_ptr() for t in I1], [t.data_ptr() for t in out], nx, ny)
        ctx.tvl1_group_dev(*args)
        st = ctx.tvl1_group_dev(*args)
        ms, n = st[0].iter_ms[0], st[0].iter_launches[0]
        us_launch = ms * 1e3 / (n / 2)
        print(json.dumps({"size": "%dx%d" % (nx, ny), "G": G, "us_per_group_launch": round(us_launch, 2), "us_per_pair_launch": round(us_launch / G, 2),
                          "frac": round(120.0 * nx * ny * G / (us_launch * 1e-6) / 8e12, 4)}), flush=True)
        ctx.set_option("profile", 0); ctx.set_option("fixed_work", 0)
        del I0, I1, out
