"""helpers.check_blend_on_device_records (K10 / K11 against the oracle on the device's own records) at the sizes the suite does not run it at: S1, S3 and a 3840 x 2160 view."""
import os, sys
import numpy as np, torch
ROOT = os.environ.get('ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT + '/tests', ROOT, ROOT + '/faster-gaussian-splatting_amd']
import helpers
from oracle import oracle as O
from harness.scenes import make_garden_like, orbit_views
from FasterGSCudaBackend._backend import default_backend
O.build(); be = default_backend()
for name, n, view in (('S1 view 0', 1_000_000, orbit_views(8)[0]), ('S3 view 5', 6_000_000, orbit_views(8)[5]), ('2 M at 3840x2160', 2_000_000, orbit_views(8, width=3840, height=2160, focal=2840.0)[2])):
    r = helpers.check_blend_on_device_records(be, O, make_garden_like(n), view, device='cuda', label=name, max_masked=3e-3)
    sums = {k: v for k, v in r.items() if k in ('mean2d.x', 'mean2d.y', 'conic.a', 'conic.b', 'conic.c', 'opacity', 'colour.r', 'colour.g', 'colour.b')}
    print(f'{name}: image {r["image"]:.2e}, final T {r["final_T"]:.2e}, last contributor differs on {r["last_contributor_differs"]:.2e} of the pixels, masked pixels {r["masked_pixels"]:.2e}, '
          f'worst K11 sum {max(sums.values()):.2e} ({max(sums, key=sums.get)}), behind a borderline pair {max(v for k, v in r.items() if k.endswith("_near")):.2e}', flush=True)
