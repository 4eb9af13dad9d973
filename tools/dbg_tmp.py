import numpy as np, torch, sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import oracle as O
from smelter_amd import ops
rng = np.random.default_rng(0)
n,hh,ww,cin,cout = 1,8,8,64,64
x = O.round_to_half(rng.standard_normal((n,cin,hh,ww)).astype(np.float32))
w = O.round_to_half((rng.standard_normal((cout,1,1,cin))*0.1).astype(np.float32))
b = np.arange(cout).astype(np.float32)*10
ref = O.conv2d(x, w, b)
xd = torch.from_numpy(np.ascontiguousarray(x.transpose(0,2,3,1))).cuda().half()
y = ops.conv2d(xd, torch.from_numpy(w).cuda().half(), torch.from_numpy(b).cuda(), cfg=9)
torch.cuda.synchronize()
got = y.float().cpu().numpy().transpose(0,3,1,2)
d = np.abs(got-ref)
print("max err", d.max())
bad = d > 0.1
print("bad per cout:", bad.sum(axis=(0,2,3)))
print("bad per pixel:", bad.sum(axis=(0,1)).ravel())
c = int(np.argmax(bad.sum(axis=(0,2,3))))
print("cout", c, "got", got[0,c].ravel()[:8], "ref", ref[0,c].ravel()[:8])
# which cout does the got value resemble?
p = 0
print("pixel0 got", np.round(got[0,:,0,0],1)); print("pixel0 ref", np.round(ref[0,:,0,0],1))
