import sys, os; sys.path.insert(0, '.')
import torch
from textualdegremoval_amd import kernels as K
def bench(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(True); e=torch.cuda.Event(True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n*1e3
for (B,C,heads,T) in [(16,768,12,1370),(4,768,12,1370),(4,1280,16,257),(4,1024,16,257)]:
    LD=((T+31)//32)*32
    qkv=torch.randn(B,3*C,LD//32,32,device='cuda')
    sc=(C//heads)**-0.5
    os.environ['TDR_ATTN_F32']='1'; a=bench(lambda: K.attention_fwd(qkv,heads,sc,T)); ref=K.attention_fwd(qkv,heads,sc,T)
    os.environ['TDR_ATTN_F32']='0'; b=bench(lambda: K.attention_fwd(qkv,heads,sc,T)); got=K.attention_fwd(qkv,heads,sc,T)
    fl=4*B*heads*T*T*(C//heads)/1e12
    print(f'B{B} C{C} h{heads} T{T}: f32 {a:8.1f} us ({fl/a*1e6:6.1f} TF)  hx2 {b:8.1f} us ({fl/b*1e6:6.1f} TF)  maxdiff {(ref-got).abs().max().item():.2e} (ref max {ref.abs().max().item():.2f})')
