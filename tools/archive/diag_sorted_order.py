import sys; sys.path[:0]=['/root/repo','/root/repo/faster-gaussian-splatting_amd','/root/repo/tests']
import numpy as np, torch, helpers
from oracle import oracle as O
from FasterGSCudaBackend._backend import default_backend
be = default_backend()
for name, (p, v) in (('wide', helpers.wide_image_scene()), ('big', helpers.many_big_footprints_scene())):
    S, RS = helpers.settings_pair(v, device='cuda')
    n = p['means'].shape[0]
    dp = {k: t.cuda().contiguous() for k, t in p.items()}
    res = be.forward(*[dp[k] for k in helpers.NAMES], RS); torch.cuda.synchronize()
    f = O.forward(*helpers.np_params(p), S, bucket_size=64)
    dec = helpers.decode_forward(be, res, n, v.width, v.height)
    print(name, 'V', dec['V'], f['V'], 'I', dec['I'], f['I'])
    for s in (0, 1):
        pk, kk = dec[f'prim_idx{s}'], dec[f'depth_keys{s}']
        eqp, eqk = np.array_equal(pk, f['prim_idx']), np.array_equal(kk, f['depth_keys'])
        print('  buf', s, 'prims equal', eqp, 'keys equal', eqk, 'keys sorted', bool(np.all(np.diff(kk.astype(np.int64)) >= 0)))
        if eqk and not eqp:
            bad = np.nonzero(pk != f['prim_idx'])[0]
            print('   mismatches', len(bad), 'first', bad[:6], 'keys there', kk[bad[:6]], 'ties?', [int((f['depth_keys'] == kk[b]).sum()) for b in bad[:6]])
    print('  inst equal', np.array_equal(dec['inst_keys'], f['inst_keys']), np.array_equal(dec['inst_prims'], f['inst_prims']))
