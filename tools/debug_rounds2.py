
import sys, os, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests'); sys.path.insert(0,'oracle')
from fluorosequencingimageanalysis_amd import _native as N
from _util import load_field, rois_of
g,img=load_field('f5_small_96'); rois=rois_of(img,g['candidates'])[:1]
ST=np.dtype([('x',np.float64,7),('diag',np.float64,7),('sdiag',np.float64,7),('fvec',np.float64,25),('r',np.float64,28),('qtf',np.float64,7),
  ('llim1',np.float64),('fnorm',np.float64),('fnorm1',np.float64),('par',np.float64),('delta',np.float64),('xnorm',np.float64),('gnorm',np.float64),('vmax',np.float64),('vmean',np.float64),
  ('data',np.uint16,25),('pad0',np.uint16,3),('niter',np.int32),('nfev',np.int32),('status',np.int32),('ipvt',np.uint32),('fresh',np.int32),('h',np.int32),('w',np.int32),('field',np.int32)])
print('state size',ST.itemsize)
def run(mode,maxr=None):
    if maxr: os.environ['FSQ_DEBUG_MAX_ROUNDS']=str(maxr)
    else: os.environ.pop('FSQ_DEBUG_MAX_ROUNDS',None)
    d=torch.from_numpy(np.ascontiguousarray(rois.astype(np.uint16)).view(np.int16)).cuda()
    rows=torch.zeros(len(rois)*128,dtype=torch.uint8,device='cuda')
    ws=torch.zeros(N.lib().fsq_fit_workspace_bytes(len(rois)),dtype=torch.uint8,device='cuda')
    rc=N.lib().fsq_fit_rois(d.data_ptr(),len(rois),mode,rows.data_ptr(),ws.data_ptr(),ws.numel(),torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    st=ws.cpu().numpy()[4096:4096+ST.itemsize].view(ST)[0]
    return rows.cpu().numpy().view(N.ROW_DTYPE), st
np.set_printoptions(linewidth=220,precision=17)
for r in (1,2,3):
    a,st=run(0,r)
    print('after',r,'rounds: x',st['x'],'niter',st['niter'],'nfev',st['nfev'],'status',st['status'],'delta',st['delta'],'par',st['par'],'fnorm',st['fnorm'],'ipvt',hex(st['ipvt']))
    print('   diag',st['diag']); print('   qtf',st['qtf']); print('   r',st['r'][:8])
