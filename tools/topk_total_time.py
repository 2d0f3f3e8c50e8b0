import sys, os, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from range_amd import _native
from tools import synth
from range_amd.bank import prepare_bank
bank = prepare_bank(*synth.make_bank(100000, 2024))
eng = _native.HipEngine("cuda:0"); eng.set_bank(bank.keys, bank.values, bank.xyz)
g = torch.Generator().manual_seed(0)
for B in (16, 32, 48, 64, 128, 256, 512, 1024, 2048, 4096, 10000):
    e32 = torch.nn.functional.normalize(torch.randn(B, 256, generator=g), dim=1).cuda()
    xq = torch.zeros(B, 4).cuda()
    res = {}
    for name, fn in (("scan+merge", lambda: eng.scan_stats(e32, xq, 12.0, 0.0, topk=16)),
                     ("stream+merge", lambda: eng.topk_stream(e32, 16))):
        if name == "stream+merge" and B > 10000: continue
        for _ in range(3): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): fn()
        b.record(); b.synchronize()
        res[name] = round(a.elapsed_time(b) / 20 * 1e3, 1)
    print(B, res)
