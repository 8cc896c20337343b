import sys, torch
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd']
from FasterGSCudaBackend._backend import default_backend
be = default_backend(); dev='cuda'
img = torch.rand(3,1080,1920,device=dev); tgt = torch.rand(3,1080,1920,device=dev)
for _ in range(5): be.l1_dssim(img,tgt,0.8,0.2)
def t(fn, n=50):
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
fwd = t(lambda: be.l1_dssim_forward(img,tgt,0.8,0.2))
l,s2,scr = be.l1_dssim_forward(img,tgt,0.8,0.2)
bwd = t(lambda: be.l1_dssim_backward(img,tgt,scr))
both = t(lambda: be.l1_dssim(img,tgt,0.8,0.2))
print(f'forward+reduce {fwd:.1f} us  backward {bwd:.1f} us  both {both:.1f} us', flush=True)
