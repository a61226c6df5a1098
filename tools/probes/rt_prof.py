import importlib, sys, time, json
sys.path.insert(0, '/root/repo')
import torch
pkg = importlib.import_module("signalsmith-stretch_amd")
sr, C, Q = 48000, 2, 128
for S in (1024, 4096):
    b = pkg.StretchBatch(S, C, preset="default", sample_rate=sr)
    t = torch.arange(Q*200, device="cuda", dtype=torch.float32)/sr
    x = (0.4*torch.sin(2*torch.pi*220.0*t)).expand(S, C, -1).contiguous()
    y = torch.empty((S, C, Q), dtype=torch.float32, device="cuda")
    for q in range(60):
        b.process(x[:, :, q*Q:(q+1)*Q], Q, out=y, ordered=False); b.synchronize()
    b.enableProfiling(1)
    rows=[]
    for q in range(60, 60+24):
        t0=time.perf_counter()
        b.process(x[:, :, q*Q:(q+1)*Q], Q, out=y, ordered=False); b.synchronize()
        dt=time.perf_counter()-t0
        ms, n = b.takeTimings()
        rows.append((round(dt*1e3,3), {k: round(v,3) for k,v in ms.items() if v>0.001}))
    b.enableProfiling(0)
    print(S, [r for r in rows if r[1].get('chain',0)>0][:2], "nohop:", rows[1])
