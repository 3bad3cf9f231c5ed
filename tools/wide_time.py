import sys; sys.path.insert(0,'.')
import numpy as np, torch, opencv_amd as oa
from opencv_amd import synth
for (W,H,pairs,win,D) in ((1920,1080,1,51,64),(640,480,1,55,30),(1920,1080,4,51,64),(1920,1080,1,101,128),(3840,2160,1,63,256)):
    L,R=synth.rds(7,W,H,4,min(D-1,20),0)
    dL=torch.from_numpy(np.stack([L]*pairs)).cuda(); dR=torch.from_numpy(np.stack([R]*pairs)).cuda()
    dO=torch.empty((pairs,H,W),dtype=torch.uint8,device='cuda')
    ctx=oa.Context(0)
    for _ in range(2): oa.compute_batch_device(dL,dR,win,D,0,out=dO,ctx=ctx)
    torch.cuda.synchronize()
    ctx.profile_begin(5)
    for _ in range(5): oa.compute_batch_device(dL,dR,win,D,0,out=dO,ctx=ctx)
    torch.cuda.synchronize()
    n,m,_p=ctx.profile_end()
    print("%d,%d,%d,%d,%d %.4f ms %s  %.0f Gd/s"%(W,H,pairs,win,D,m/n,ctx.last_kernel(), W*H*D*pairs/(m/n)/1e6),flush=True)
