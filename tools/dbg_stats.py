import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from oracle import range_oracle as O
from range_amd import _native, synth
from range_amd.bank import prepare_bank
N,B=3001,77
locs, vals, keys = synth.make_bank(N, 77)
bank = prepare_bank(locs, vals, keys)
w = synth.make_encoder_weights(10,64,256,2,5)
q = synth.make_queries(B, seed=5); e = O.encode(q, w, 10)
xq4 = np.zeros((B,4),np.float32); xq4[:,:3]=O.query_xyz(q)
e32 = torch.from_numpy(e.astype(np.float32)).cuda(); xq=torch.from_numpy(xq4).cuda()
def eng(b, off=0):
    g=_native.HipEngine("cuda:0"); g.set_bank(b.keys,b.values,b.xyz,off); return g
full=eng(bank); a=eng(bank.rows(0,1400)); b=eng(bank.rows(1400,N),1400)
for tau_geo in (40.0, 0.0):
    sf=full.scan_stats(e32,xq,12.0,tau_geo)
    sa=a.scan_stats(e32,xq,12.0,tau_geo); sb=b.scan_stats(e32,xq,12.0,tau_geo)
    st=full.merge_stats(torch.stack([sa,sb]))
    torch.cuda.synchronize()
    lse=lambda s:(s[:,0::2].double()+torch.log2(s[:,1::2].double()))
    d=(lse(st)-lse(sf)).abs()
    print(tau_geo, d.max().item(), d.argmax().item(), st[d.argmax()//2], sf[d.argmax()//2], sa[d.argmax()//2], sb[d.argmax()//2])
