import sys, time, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests'); sys.path.insert(0,'oracle')
from fluorosequencingimageanalysis_amd import _native as N
from _util import load_field, rois_of
g,img=load_field('f1_cfg2_512_500'); rois=rois_of(img,g['candidates'])
reps=int(sys.argv[1]) if len(sys.argv)>1 else 64
big=np.tile(rois,(reps,1)); n=len(big)
d=torch.from_numpy(np.ascontiguousarray(big.astype(np.uint16)).view(np.int16)).cuda()
rows=torch.zeros(n*128,dtype=torch.uint8,device='cuda')
ws=torch.zeros(N.lib().fsq_fit_workspace_bytes(n),dtype=torch.uint8,device='cuda')
mode=int(sys.argv[2]) if len(sys.argv)>2 else 0
L=N.lib(); s=torch.cuda.current_stream().cuda_stream
for it in range(3):
    torch.cuda.synchronize(); t=time.time()
    N.check(L.fsq_fit_rois(d.data_ptr(),n,mode,rows.data_ptr(),ws.data_ptr(),ws.numel(),s),'fit'); torch.cuda.synchronize(); dt=time.time()-t
    print('n=%d  %.3f s  %.3e fits/s'%(n,dt,n/dt))
r=rows.cpu().numpy().view(N.ROW_DTYPE)
print('mean nfev',r['nfev'].mean(),'niter',r['niter'].mean())

import ctypes
if hasattr(L,'fsq_debug_phase_cycles'):
    buf=(ctypes.c_ulonglong*16)(); L.fsq_debug_phase_cycles(buf,1); v=list(buf)
    names=['refill','J-evals','diff+peg','qrfac','pre-inner','lmpar','trial+logic','term']
    tot=max(sum(v[:8]),1)
    for nm,c in zip(names,v[:8]): print('%-12s %6.2f%%'%(nm,100*c/tot))
    print('qlm load %.2f%%  gnorm+diag %.2f%%'%(100*v[10]/(tot+v[10]+v[11]),100*v[11]/(tot+v[10]+v[11])))
    print('inner passes per trip',v[8]/max(v[9],1),'trips',v[9])

if hasattr(L,'fsq_debug_rphase'):
    buf=(ctypes.c_ulonglong*32)(); L.fsq_debug_rphase(buf,1); v=list(buf)
    kb=dict(zip(['loop/idle','load','lmpar run','step logic','trial eval','update logic','store','lmpar begin'],v[:8]))
    tot=sum(kb.values())
    for nm,c in kb.items(): print('kB %-12s %6.2f%%'%(nm,100*c/max(tot,1)))
    ka=dict(zip(['loop/idle','load+stage','J evals','diff+peg','qrfac: norms','gnorm etc','handover','qr: pivot','qr: owner','qr: updates','qr: downdate','qr: tail'],v[16:28]))
    tota=sum(ka.values())
    for nm,c in ka.items(): print('kA %-12s %6.2f%%'%(nm,100*c/max(tota,1)))
    print('kA total / kB total cycles: %.2f'%(tota/max(tot,1)))
    passes=float(r['niter'].sum())*3
    print('kA ticks per wave-pass (16 fits): %.0f   kB ticks per wave-pass (64 fits): %.0f'%(tota/(passes/16), tot/(passes/64)))
