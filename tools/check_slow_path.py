import os, sys, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests'); sys.path.insert(0,'oracle')
from fluorosequencingimageanalysis_amd import _native as N
from _util import load_field, rois_of, bits_equal
g,img=load_field('f3_hard_256'); rois=rois_of(img,g['candidates'])
def fit():
    n=len(rois)
    d=torch.from_numpy(np.ascontiguousarray(rois.astype(np.uint16)).view(np.int16)).cuda()
    rows=torch.zeros(n*128,dtype=torch.uint8,device='cuda')
    ws=torch.zeros(N.lib().fsq_fit_workspace_bytes(n),dtype=torch.uint8,device='cuda')
    N.check(N.lib().fsq_fit_rois(d.data_ptr(),n,0,rows.data_ptr(),ws.data_ptr(),ws.numel(),torch.cuda.current_stream().cuda_stream),'fit')
    torch.cuda.synchronize()
    return rows.cpu().numpy().view(N.ROW_DTYPE)
os.environ['FSQ_DEBUG_FORCE_SLOW']=sys.argv[1] if len(sys.argv)>1 else '3'
got=fit()
print('slow count',N.lib().fsq_fit_last_slow_count())
p=np.stack([got[k] for k in ("H","A","p2","p3","sigma_h","sigma_w","theta")],axis=1)
bad=np.where(~bits_equal(p,g['params']).all(axis=1))[0]
print('n',len(rois),'bad',len(bad),bad[:20], 'bad%3',np.unique(bad%3))
for i in bad[:5]:
    print(i,'status',got['status'][i],g['status'][i],'niter',got['niter'][i],g['niter'][i],'nfev',got['nfev'][i], p[i], g['params'][i])
