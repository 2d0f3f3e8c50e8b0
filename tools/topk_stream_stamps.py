import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from range_amd import _native, synth
from range_amd.bank import prepare_bank
bank = prepare_bank(*synth.make_bank(100000, 2024))
eng = _native.HipEngine("cuda:0"); eng.set_bank(bank.keys, bank.values, bank.xyz)
g = torch.Generator().manual_seed(0)
for B in (16, 64):
    e32 = torch.nn.functional.normalize(torch.randn(B, 256, generator=g), dim=1).cuda()
    os.environ.pop("RANGE_TOPKS_STAMPS", None)
    for _ in range(5): eng.topk_stream(e32, 16)
    torch.cuda.synchronize()
    os.environ["RANGE_TOPKS_STAMPS"] = "1"
    print("B", B, file=sys.stderr)
    for _ in range(3): eng.topk_stream(e32, 16)
    torch.cuda.synchronize()
