import sys, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests'); sys.path.insert(0,'oracle')
from fluorosequencingimageanalysis_amd import _native as N
import oracle as O
from _util import load_field, rois_of
g,img=load_field('f5_small_96'); rois=rois_of(img,g['candidates'])
def run(mode):
    d=torch.from_numpy(np.ascontiguousarray(rois.astype(np.uint16)).view(np.int16)).cuda()
    rows=torch.zeros(len(rois)*128,dtype=torch.uint8,device='cuda')
    ws=torch.zeros(N.lib().fsq_fit_workspace_bytes(len(rois)),dtype=torch.uint8,device='cuda')
    rc=N.lib().fsq_fit_rois(d.data_ptr(),len(rois),mode,rows.data_ptr(),ws.data_ptr(),ws.numel(),torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize(); print('rc',rc)
    return rows.cpu().numpy().view(N.ROW_DTYPE)
a=run(0); b=run(0x200)
for i in range(3):
    print('rounds',[a[k][i] for k in ('H','A','p2','p3','sigma_h','sigma_w','theta','status','niter','nfev')])
    print('quad  ',[b[k][i] for k in ('H','A','p2','p3','sigma_h','sigma_w','theta','status','niter','nfev')])
print('status eq',(a['status']==b['status']).mean(),'niter eq',(a['niter']==b['niter']).mean(), 'nfev eq', (a['nfev']==b['nfev']).mean())
