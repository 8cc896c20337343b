"""Loss kernels alone (1080p), for rocprofv3 --pmc passes: 20 forward + backward launches through the autograd node."""
import sys, torch
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd']
from harness.loss import l1_dssim_loss
x = torch.rand(3, 1080, 1920, device='cuda', requires_grad=True); y = torch.rand(3, 1080, 1920, device='cuda')
for _ in range(20):
    l1_dssim_loss(x, y).backward(); x.grad = None
torch.cuda.synchronize()
