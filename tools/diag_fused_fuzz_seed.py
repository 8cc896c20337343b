"""One fuzz configuration through the fused backward + Adam path, two steps, against oracle backward -> oracle Adam (tests/test_gpu_fused._run, spelled out):
the worst first-moment entry of the means after each step, whether the risk masks name its Gaussian, and at which band width they would.
usage: python tools/diag_fused_fuzz_seed.py SEED"""
import os, sys, numpy as np, torch
ROOT = os.getcwd()
sys.path[:0] = [ROOT + '/tests', ROOT, ROOT + '/faster-gaussian-splatting_amd']
import helpers
from oracle import oracle as O
from FasterGSCudaBackend._backend import default_backend
import test_gpu_fused as TF
O.build()
seed = int(sys.argv[1])
p, view, K, aa, label = helpers.fuzz_configuration(seed); print(label)
SIM = os.environ.get('FGS_DIAG_SIM') == '1'            # FGS_DIAG_SIM=1: the CPU simulation of the same sources (tests/sim) instead of the GPU
be = helpers.sim_backend() if SIM else default_backend(); DEV = 'cpu' if SIM else 'cuda:0'
ORDER, LRS, GRAD_OF = TF.ORDER, TF.LRS, TF.GRAD_OF
S, RS = helpers.settings_pair(view, K, aa, device=DEV)
n = p['means'].shape[0]
gen = torch.Generator().manual_seed(7)
P0 = {k: p[k].clone() for k in ORDER}
M0 = {k: torch.randn(p[k].shape, generator=gen) * 1e-3 for k in ORDER}
V0 = {k: torch.rand(p[k].shape, generator=gen) * 1e-6 for k in ORDER}
dP, dM, dV = ({k: d[k].to(DEV).contiguous().clone() for k in ORDER} for d in (P0, M0, V0))
oP, oM, oV = ({k: np.ascontiguousarray(d[k].numpy().copy()) for k in ORDER} for d in (P0, M0, V0))
gi = torch.randn(3, view.height, view.width, generator=gen) / (view.height * view.width)
gi_np, gi_dev = gi.numpy(), gi.to(DEV)
dens_dev, dens_o = torch.zeros(2, n, device=DEV), np.zeros((2, n), np.float32)
if os.environ.get('FGS_DIAG_LOAD'):                         # replay ONE step from dumped parameters (moments are irrelevant to the gradient)
    z = np.load(os.environ['FGS_DIAG_LOAD'])
    for k in ORDER:
        dP[k] = torch.from_numpy(z[k]).to(DEV).contiguous(); oP[k] = np.ascontiguousarray(z[k].copy())
for step in ((2,) if os.environ.get('FGS_DIAG_LOAD') else (1, 2)):
    res = be.forward(*[dP[k] for k in helpers.NAMES], RS)
    dec = helpers.decode_forward(be, res, n, view.width, view.height)
    # unfused gradient of the same state, for reference
    devP = {k: dP[k].cpu().numpy().copy() for k in ORDER}            # the oracle evaluated at the DEVICE's parameters of this step
    if os.environ.get('FGS_DIAG_DUMP'):                     # the device's parameters in front of this step, for a replay elsewhere (FGS_DIAG_LOAD)
        np.savez(os.environ['FGS_DIAG_DUMP'] + f'_step{step}.npz', **devP)
    f_at_dev = O.forward(*[devP[k] for k in helpers.NAMES], S, bucket_size=64)
    g_at_dev = O.backward(f_at_dev, S, gi_np, np.zeros((2, n), np.float32))
    g_dev = be.backward(torch.zeros(2, n, device=DEV), gi_dev, res.image, dP['means'], dP['scales'], dP['rotations'], dP['opacities'], dP['sh_coefficients_rest'], res.buffers, RS, res.state)
    lp = be.blob_layout(0, n, view.width, view.height, res.state[1], res.state[2])
    acc_dev = be.view(res.buffers[0].cpu(), lp, 'acc', torch.float32).reshape(n, 9).numpy().copy() if 'acc' in lp else None
    be.backward_adam_fused(dens_dev, gi_dev, res.image, [dP[k] for k in ORDER], [dM[k] for k in ORDER], [dV[k] for k in ORDER], res.buffers, RS, res.state, step, LRS)
    f = O.forward(*[oP[k] for k in helpers.NAMES], S, bucket_size=64)
    masks = helpers.flip_masks(O, f, S, dec)
    g = O.backward(f, S, gi_np, dens_o)
    gm_dev = dict(zip(helpers.GRAD_KEYS, g_dev))
    for k, lr in zip(ORDER, LRS):
        O.adam_step(np.ascontiguousarray(g[GRAD_OF[k]].reshape(oP[k].shape)), oP[k], oM[k], oV[k], step, lr)
    d = np.abs(dM['means'].cpu().numpy() - oM['means']).max(axis=1); i = int(d.argmax())
    gk = GRAD_OF['means']
    print(f'step {step}: worst exp_avg(means) Gaussian {i}: abs err {d[i]:.3e} of max {np.abs(oM["means"]).max():.3e}; prim-masked {bool(masks["prim"][i])} near {bool(masks["near"][i])} '
          f'n_touched dev {int(dec["n_touched"][i])} oracle {int(f["n_touched"][i])}; bounds dev {dec["screen_bounds"][i]} oracle {f["screen_bounds"][i]}')
    print('   unfused device grad', gm_dev[gk][i].cpu().numpy().ravel(), 'oracle grad', g[gk][i].ravel())
    print('   oracle grad AT THE DEVICE PARAMETERS', g_at_dev[gk][i].ravel(), '| colour (oracle state)', f['color'][i], '(device state)', f_at_dev['color'][i], 'device record', dec['color'][i])
    t64 = O.forward_backward_f64(f, S, gi_np)
    print('   fp64 value of the gradient', np.asarray(t64[gk]).reshape(g[gk].shape)[i].ravel(), '| oracle 2D sums: mean2d', g['_grad_mean2d'][i], 'conic', g['_grad_conic'][:, i])
    if acc_dev is not None:
        print('   device 2D sums (mean2d.xy, conic.abc, opacity, colour)', acc_dev[i], '| oracle opacity / sh0 grads', g['opacities'][i].ravel(), g['sh0'][i].ravel())
        bad = np.abs(acc_dev[:, :2] - g['_grad_mean2d']).max(axis=1); j = int(bad.argmax())
        print('   worst mean2d sum over all Gaussians: index', j, 'device', acc_dev[j, :2], 'oracle', g['_grad_mean2d'][j], 'n_touched', int(f['n_touched'][j]))
    sb = [int(v) for v in f['screen_bounds'][i]]
    npr = helpers.tiles_to_image(dec['n_processed_tiles'], view.width, view.height); fT = helpers.tiles_to_image(dec['final_T_tiles'], view.width, view.height, fill=1.0)
    o_np, o_T = f['n_processed'].reshape(npr.shape), f['final_T'].reshape(npr.shape)
    y0, y1, x0, x1 = sb[2], min(sb[3], view.height), sb[0], min(sb[1], view.width)
    dn = (npr[y0:y1, x0:x1] != o_np[y0:y1, x0:x1]); dT = np.abs(fT[y0:y1, x0:x1] - o_T[y0:y1, x0:x1])
    print(f'   forward inside its bounds {sb}: pixels whose last contributor differs {int(dn.sum())}, largest |final T difference| {float(dT.max()):.3e}, image difference {float(np.abs(res.image.cpu().numpy() - f["image"])[:, y0:y1, x0:x1].max()):.3e}')
    gw = (view.width + 15) // 16
    for ty in range(y0 // 12, (y1 + 11) // 12):
        for tx in range(x0 // 16, (x1 + 15) // 16):
            t = ty * gw + tx; r0, r1 = [int(v) for v in f['ranges'][t]]; lst = [int(v) for v in f['inst_prims'][r0:r1]]
            d0, d1 = [int(v) for v in dec['ranges'][t]]; dl = [int(v) for v in dec['inst_prims'][d0:d1]]
            print(f'   tile {t} ({tx},{ty}): list length oracle {r1 - r0} device {d1 - d0}, position of the Gaussian oracle {lst.index(i) if i in lst else None} device {dl.index(i) if i in dl else None}, lists equal {lst == dl}, max_n_processed device {int(dec["max_n_processed"][t])} oracle {int(o_np[ty*12:ty*12+12, tx*16:tx*16+16].max())}')
    print('   conic / opacity oracle state', f['conic_opacity'][i], 'device state', f_at_dev['conic_opacity'][i])
    print('   device exp_avg', dM['means'][i].cpu().numpy(), 'oracle', oM['means'][i], 'params dev', dP['means'][i].cpu().numpy(), 'oracle', oP['means'][i])
    for eps in (5e-6, 2e-5, 1e-4, 1e-3):
        mm = helpers.flip_masks(O, f, S, dec, eps=eps, eps_T=max(1e-5, eps))
        print(f'   band {eps:.0e}: Gaussian {i} prim-masked {bool(mm["prim"][i])} near {bool(mm["near"][i])}; masked prims {int(mm["prim"].sum())} near {int(mm["near"].sum())}')
    print('   V', dec['V'], f['V'], 'I', dec['I'], f['I'], 'masked prims', int(masks['prim'].sum()), 'near', int(masks['near'].sum()))
