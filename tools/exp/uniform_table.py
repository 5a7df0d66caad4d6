import sys, os, json, time
sys.path.insert(0, '/root/repo' if os.path.isdir('/root/repo/tests') else os.environ.get('GRAFT_REPO_ROOT','.'))
import numpy as np, torch, _vitpkg
from bench import make_frames
V=_vitpkg.load_package(); V.initialize(); V.set_kernel(2)
dev=torch.device('cuda',0)
for fb,n in ((4608,81920),(4608,49152),(6912,36400)):
    sym=make_frames(n,fb,seed=fb,device=dev).reshape(-1)
    desc,sb,ob=V.make_descs([fb]*n)
    d_desc=torch.from_numpy(desc.view(np.uint8)).to(dev)
    out=torch.zeros(ob,dtype=torch.uint8,device=dev); out2=torch.zeros((n,fb//8),dtype=torch.uint8,device=dev)
    def t(fn):
        for _ in range(5): fn()
        torch.cuda.synchronize(); a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); a.record()
        for _ in range(10): fn()
        b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/10
    ms_d=t(lambda: V.decode_varlen_dev(sym,out,d_desc,n,fb)); ms_u=t(lambda: V.decode_batch_dev(sym.view(n,-1),out2,fb,n))
    print(json.dumps({"framebits":fb,"frames":n,"ms_desc_table":round(ms_d,4),"ms_batch":round(ms_u,4),"same":bool((out.view(n,-1)==out2).all())}),flush=True)
