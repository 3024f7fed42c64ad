import sys, time, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests'); sys.path.insert(0,'oracle')
from fluorosequencingimageanalysis_amd import _native as N
from _util import load_field, rois_of
g,img=load_field('f1_cfg2_512_500'); rois=rois_of(img,g['candidates'])
big=np.tile(rois,(256,1)); n=len(big)
d=torch.from_numpy(np.ascontiguousarray(big.astype(np.uint16)).view(np.int16)).cuda()
rows=torch.zeros(n*128,dtype=torch.uint8,device='cuda')
ws=torch.zeros(N.lib().fsq_fit_workspace_bytes(n),dtype=torch.uint8,device='cuda')
L=N.lib(); s=torch.cuda.current_stream().cuda_stream
N.check(L.fsq_fit_rois(d.data_ptr(),n,0,rows.data_ptr(),ws.data_ptr(),ws.numel(),s),'fit'); torch.cuda.synchronize()
