import sys, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests'); sys.path.insert(0,'tests/golden')
import cases as C
from cases import O
from mnasnet_pytorch_amd import ConvBlock
import numpy as np
name='pw_16_48'
cin, cout, k, s, p, grp, N, H, W = C.PRIMITIVES[name]
m = ConvBlock(cin, cout, kernel_size=k, stride=s, padding=p, groups=grp)
sd = m.state_dict(); new={}
for kk,v in sd.items(): new[kk]=O.det_param(name+'.'+kk, tuple(v.shape), C.STATE_SEED).to(v.dtype)
m.load_state_dict(new); m=m.cuda().train()
x = C.det_input((N, cin, H, W)).cuda()
y = m(x)
torch.cuda.synchronize()
eng = m._engine()
prog = eng.programs[(N,H,W,True,False)][0]
bn = [t for t in prog.keep if t.dtype==torch.float32 and t.shape[0]==8][0]
yraw = prog.keep[1].float().cpu().permute(0,3,1,2)
import torch.nn.functional as F
ref_raw = F.conv2d(x.cpu(), new['conv.weight'], new['conv.bias'])
print('raw conv rel err', float((yraw-ref_raw).norm()/ref_raw.norm()))
print('mean hip', bn[5][:6].cpu().numpy()); print('mean ref', ref_raw.mean((0,2,3))[:6].numpy())
print('invstd hip', bn[6][:6].cpu().numpy()); print('invstd ref', (1/torch.sqrt(ref_raw.var((0,2,3),unbiased=False)+1e-5))[:6].numpy())
print('scratch', eng.scratch_stats[:8].cpu().numpy())
g=np.load('tests/golden/primitives.npz')
print('y err', float((y.detach().cpu()-torch.tensor(g[name+'/train/y'])).norm()/torch.tensor(g[name+'/train/y']).norm()))
