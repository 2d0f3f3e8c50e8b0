"""(with a -DRANGE_EXP_TS_STAMPS build loaded through RANGE_LIB_PATH) where a range_topk_stream
launch's time goes: per-wave stamps of the tile loop and per-workgroup stamps of the merge tail."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from range_amd import _native
from tools.scan_bench import make_keys
dev = torch.device("cuda:0")
for n in (100_000, 1_000_000):
    keys = make_keys(n, dev)
    eng = _native.HipEngine(dev); eng.set_keys(keys)
    g = torch.Generator().manual_seed(0)
    for B in (16, 32, 64):
        e32 = torch.nn.functional.normalize(torch.randn(B, 256, generator=g), dim=1).cuda()
        os.environ.pop("RANGE_TOPKS_STAMPS", None)
        for _ in range(5): eng.topk_stream(e32, 16)
        torch.cuda.synchronize()
        os.environ["RANGE_TOPKS_STAMPS"] = "1"
        print("N", n, "B", B, file=sys.stderr)
        for _ in range(3): eng.topk_stream(e32, 16)
        torch.cuda.synchronize()
    eng.close()
