"""How much of the zero-gradient traffic is skippable at block granularity? (CPU, oracle forward.) At S2 one third of the Gaussians is
invisible per view: K12 writes 236 bytes of zeros for each and Adam reads them back. K12 already works on waves of 64 consecutive Gaussians:
the fraction of 64 / 256 / 1024-Gaussian blocks without a single visible Gaussian bounds what a block flag could save."""
import sys
sys.path[:0] = ['/root/repo', '/root/repo/faster-gaussian-splatting_amd', '/root/repo/tests']
import helpers
from oracle import oracle
from harness.scenes import make_garden_like, orbit_views
p = make_garden_like(3_000_000)
a = helpers.np_params(p)
for vi in (0, 2, 5):
    S, _ = helpers.settings_pair(orbit_views(8)[vi])
    vis = oracle.forward(*a, S, bucket_size=64)['n_touched'] > 0
    out = [f'view {vi}: visible {vis.mean():.3f}']
    for blk in (64, 256, 1024):
        m = vis.shape[0] // blk * blk
        any_vis = vis[:m].reshape(-1, blk).any(axis=1)
        out.append(f'blocks of {blk}: {1 - any_vis.mean():.3f} dead = {(~any_vis).sum() * blk / (~vis[:m]).sum():.3f} of the invisible Gaussians')
    print(' | '.join(out))
