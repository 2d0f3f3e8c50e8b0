"""(with a -DRANGE_EXP_TS_STAMPS build loaded through RANGE_LIB_PATH) the publish of the bf16 scan per
supergroup: the line "per wave over the tile loop" then reads waiting = wait at the publish's barrier,
issuing DMA = lane -> wave merges + LDS stores, arithmetic = workgroup merge + its stores + second barrier
(sums over the launch's supergroups; 64 queries = 1 supergroup, 256 = 4)."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from range_amd import _native
from tools.scan_bench import make_keys
dev = torch.device("cuda:0")
keys = make_keys(100_000, dev)
eng = _native.HipEngine(dev); eng.set_keys(keys)
g = torch.Generator().manual_seed(0)
for B in (64, 256):
    e32 = torch.nn.functional.normalize(torch.randn(B, 256, generator=g), dim=1).cuda()
    os.environ.pop("RANGE_TOPKS_STAMPS", None)
    for _ in range(3): eng.topk_stream(e32, 16)
    torch.cuda.synchronize()
    os.environ["RANGE_TOPKS_STAMPS"] = "1"
    print("N 100000 B", B, file=sys.stderr)
    eng.topk_stream(e32, 16)
    torch.cuda.synchronize()
