import sys, torch, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests'); sys.path.insert(0,'tests/golden')
import cases as C
from cases import O
from oracle import bf16_mirror as M
from test_oracle_golden import block_state, prim_state
from test_gpu_model import fill, rl2
from mnasnet_pytorch_amd import MBConv_block, ConvBlock
for name in sorted(C.PRIMITIVES):
    cin, cout, k, s, p, grp, N, H, W = C.PRIMITIVES[name]
    m = ConvBlock(cin, cout, kernel_size=k, stride=s, padding=p, groups=grp); fill(m,name); m=m.cuda().train()
    x0=C.det_input((N,cin,H,W)); x=x0.cuda().requires_grad_(cin!=3); y=m(x); cot=C.cotangent(tuple(y.shape)); (y*cot.cuda()).sum().backward()
    spec=O.ConvSpec("cb",cin,cout,k,s,p,grp); st=prim_state(name,spec); r=M.run([("conv",spec)],st,x0,True,cot,need_dx=True)
    print(name,'y %.5f'%rl2(y.detach().cpu(),r['y']), 'dx %.5f'%(rl2(x.grad.cpu(),r['dx']) if cin!=3 else 0), ' '.join('%s %.5f'%(kk.split('.')[0][0]+kk.split('.')[1][0],rl2(pp.grad.cpu(),r['grads']['cb.'+kk])) for kk,pp in m.named_parameters() if not kk.endswith('conv.bias')))
for name in sorted(C.BLOCKS):
    c,t,k,N,H,W=C.BLOCKS[name]
    m=MBConv_block(c,t,k); fill(m,name); m=m.cuda().train()
    x0=C.det_input((N,c,H,W)); x=x0.cuda().requires_grad_(True); y=m(x); cot=C.cotangent(tuple(y.shape)); (y*cot.cuda()).sum().backward()
    specs=O._block_specs("blk",c,t,k); st=block_state(name,specs); r=M.run([("block",specs)],st,x0,True,cot,need_dx=True)
    print(name,'y %.5f'%rl2(y.detach().cpu(),r['y']),'dx %.5f'%rl2(x.grad.cpu(),r['dx']))
    for kk,pp in m.named_parameters():
        if kk.endswith('conv.bias'): continue
        print('   ',kk,'%.5f'%rl2(pp.grad.cpu(),r['grads']['blk.'+kk]))
