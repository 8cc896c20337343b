"""The binning stage's own radix sort (csrc/radix_sort.hip) through its debug hook: stable, bit-range limited, any length --
against numpy's stable argsort. CPU: the simulation build of the same source; GPU (-m gpu): the HIP library at full size."""
import ctypes as C

import numpy as np
import pytest
import torch

import helpers


def _sort(be, keys: np.ndarray, end_bit: int, dev: str):
    n = keys.shape[0]
    kt = torch.from_numpy(keys.view(np.int16 if keys.dtype == np.uint16 else np.int32).copy()).to(dev)
    k1 = torch.full_like(kt, -1)
    v0 = torch.arange(n, dtype=torch.int32, device=dev)
    v1 = torch.full_like(v0, -1)
    nbytes = int(be.lib.fgs_debug_radix_sort_temp_bytes(n, end_bit))
    temp = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream if dev != 'cpu' else 0
    ptr = lambda t: t.data_ptr() if t.numel() else None
    sel = be.lib.fgs_debug_radix_sort(ptr(kt), ptr(k1), ptr(v0), ptr(v1), n, keys.dtype.itemsize, end_bit, temp.data_ptr(), nbytes, stream)
    assert sel in (0, 1), be.lib.fgs_last_error()
    ks, vs = ((kt, v0), (k1, v1))[sel]
    return ks.cpu().numpy().view(keys.dtype), vs.cpu().numpy().astype(np.int64)


def _check(be, n, dtype, end_bit, dev, seed, n_distinct=None):
    rng = np.random.default_rng(seed)
    hi = 1 << (8 * np.dtype(dtype).itemsize)
    keys = rng.integers(0, hi, n, dtype=np.uint64).astype(dtype)
    if n_distinct:                                    # many duplicates: stability is visible
        keys = rng.choice(rng.integers(0, hi, n_distinct, dtype=np.uint64).astype(dtype), n)
    ks, vs = _sort(be, keys, end_bit, dev)
    masked = keys.astype(np.uint64) & np.uint64((1 << end_bit) - 1)
    order = np.argsort(masked, kind='stable')
    assert np.array_equal(vs, order), (n, dtype, end_bit)
    assert np.array_equal(ks, keys[order])


@pytest.mark.parametrize('n', [0, 1, 63, 64, 65, 1023, 4095, 4096, 4097, 12289, 40000])
def test_sort_lengths_sim(n):
    be = helpers.sim_backend()
    _check(be, n, np.uint32, 32, 'cpu', n)
    _check(be, n, np.uint16, 14, 'cpu', n + 1, n_distinct=300)


@pytest.mark.parametrize('dtype,end_bit', [(np.uint16, 7), (np.uint16, 16), (np.uint32, 17), (np.uint32, 32), (np.uint32, 21), (np.uint32, 3),
                                           (np.uint32, 1), (np.uint16, 9)])
def test_sort_bit_ranges_sim(dtype, end_bit):
    _check(helpers.sim_backend(), 9000, dtype, end_bit, 'cpu', end_bit, n_distinct=50 if end_bit > 6 else None)


def test_sort_many_workgroups_sim():
    """More workgroups than one chunk of the row scan (4096): 4 100 workgroups of 4 096 items would be 16.8 M items, too slow for
    the emulator -- the row scan is exercised with a small-item / many-row shape instead: see test_sort_on_device for full size."""
    _check(helpers.sim_backend(), 70_000, np.uint16, 14, 'cpu', 5, n_distinct=12_000)


@pytest.mark.gpu
@pytest.mark.parametrize('n', [1, 4097, 250_000, 2_049_194, 16_234_857, 17_000_000])
def test_sort_on_device(hip_backend, n):
    _check(hip_backend, n, np.uint32, 32, 'cuda', 1)                       # depth keys
    _check(hip_backend, n, np.uint16, 14, 'cuda', 2, n_distinct=12_240)   # tile keys at 1080p
    if n <= 2_049_194:
        _check(hip_backend, n, np.uint32, 17, 'cuda', 3, n_distinct=70_000)
