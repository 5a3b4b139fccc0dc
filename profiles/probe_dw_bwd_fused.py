import sys; sys.path.insert(0, ".")
import os, torch, time
from textualdegremoval_amd import kernels as K
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(True); e=torch.cuda.Event(True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n*1e3
for name,(N,C,H,W) in {'restormer fus L1 gelu':(8,255,256,256),'restormer enc L1 gelu':(8,127,256,256),'L2 gelu':(8,510,128,128),'naf L0 mul':(4,64,512,512),'naf fus L0 mul':(4,128,512,512),'promptir 384':(8,127,384,384)}.items():
    t=torch.randn(N,2*C,H,W,device='cuda'); w=torch.randn(2*C,1,3,3,device='cuda'); b=torch.randn(2*C,device='cuda'); dg=torch.randn(N,C,H,W,device='cuda'); do=torch.randn(N,2*C,H,W,device='cuda')
    for kind in ('gelu','mul','none'):
        f={'gelu':lambda: K.dwgelu_bwd(dg,t,w,b),'mul':lambda: K.dwsg_bwd(dg,t,w,b),'none':lambda: K.dwconv_bwd(do,t,w)}[kind]
        os.environ.pop('TDR_DWSG_TWO_PASS',None); a=bench(f)
        os.environ['TDR_DWSG_TWO_PASS']='1'; b2=bench(f); os.environ.pop('TDR_DWSG_TWO_PASS')
        gb=(N*2*C*H*W*4*(2 if kind!='none' else 3)+ (N*C*H*W*4 if kind!='none' else 0))/1e9
        print(f'{name:24s} {kind:5s} fused {a:8.1f} us ({gb/a*1e3:5.2f} TB/s alg)  two-pass {b2:8.1f} us')
