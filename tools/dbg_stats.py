import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import range_oracle as O
from range_amd import _native, synth
from range_amd.bank import prepare_bank
LOG2E=1.4426950408889634
for N,B in ((16,1),(32,1),(48,5),(64,5),(100,5),(500,33),(512,33),(1600,64),(20000,64)):
    locs, vals, keys = synth.make_bank(N, 77)
    bank = prepare_bank(locs, vals, keys); ob = O.prep_bank(locs, vals, keys)
    w = synth.make_encoder_weights(10,64,256,2,5)
    q = synth.make_queries(B, seed=5); e = O.encode(q, w, 10)
    eng = _native.HipEngine("cuda:0"); eng.set_bank(bank.keys, bank.values, bank.xyz)
    xq4 = np.zeros((B,4),np.float32); xq4[:,:3]=O.query_xyz(q)
    e32 = torch.from_numpy(e.astype(np.float32)).cuda(); xq=torch.from_numpy(xq4).cuda()
    st = eng.scan_stats(e32,xq,12.0,40.0).cpu().numpy().astype(np.float64)
    s,g = O.logits64(e,q,ob)
    m1,l1 = O.shard_stats64(s,12.0); m2,l2=O.shard_stats64(g,40.0)
    lse=(st[:,0]+np.log2(st[:,1]))/LOG2E; lse2=(st[:,2]+np.log2(st[:,3]))/LOG2E
    print(N,B,"pipe sem err", np.abs(lse-(m1+np.log(l1))).max(), "geo err", np.abs(lse2-(m2+np.log(l2))).max())
