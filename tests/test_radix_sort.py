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


# ---- the depth sort as the forward pass runs it: key - bits(near), 9-bit digits (fgs_debug_depth_sort; option 9 selects the variants) ------
def _depth_sort(be, keys: np.ndarray, near: float, far: float, dev: str):
    n = keys.shape[0]
    kt = torch.from_numpy(keys.view(np.int32).copy()).to(dev)
    k1 = torch.full_like(kt, -1)
    v0 = torch.arange(n, dtype=torch.int32, device=dev)
    v1 = torch.full_like(v0, -1)
    nbytes = int(be.lib.fgs_debug_radix_sort_temp_bytes(n, 32))
    temp = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream if dev != 'cpu' else 0
    ptr = lambda t: t.data_ptr() if t.numel() else None
    sel = be.lib.fgs_debug_depth_sort(ptr(kt), ptr(k1), ptr(v0), ptr(v1), n, near, far, temp.data_ptr(), nbytes, stream)
    assert sel in (0, 1), be.lib.fgs_last_error()
    ks, vs = ((kt, v0), (k1, v1))[sel]
    return ks.cpu().numpy().view(np.uint32), vs.cpu().numpy().astype(np.int64)


def _depth_keys(n, near, far, seed, n_distinct=None):
    """float32 depths inside [near, far] incl. both ends (what survives the cull of kernels_forward.cuh:67), as their bit patterns"""
    rng = np.random.default_rng(seed)
    lo, hi = np.float32(near), np.float32(far)
    if lo > 0 and hi / lo > 4:
        d = np.exp(rng.uniform(np.log(float(lo)), np.log(float(hi)), n)).astype(np.float32)      # log-uniform: every exponent occurs
    else:
        d = rng.uniform(float(lo), float(hi), n).astype(np.float32)
    d = np.clip(d, lo, hi)
    if n >= 2:
        d[0], d[-1] = hi, lo
    if n_distinct:
        d = rng.choice(d[:n_distinct], n)
    return d.view(np.uint32)


def _check_depth(be, n, near, far, dev, seed, n_distinct=None):
    keys = _depth_keys(n, near, far, seed, n_distinct)
    ks, vs = _depth_sort(be, keys, near, far, dev)
    order = np.argsort(keys, kind='stable')                       # order by key - bits(near) == order by key
    assert np.array_equal(vs, order), (n, near, far)
    assert np.array_equal(ks, keys[order])


DEPTH_RANGES = [(0.2, 1e4), (0.01, 100.0), (1.0, 1.0000153), (0.0, 1e4), (3.0, 3.0), (0.5, 65504.0)]   # 27 / 27 / 7 / 31 / 1 / 28 bits


# the default mode over every range; the A/B modes on the two structurally different ones (27 bits above a base / 31 bits, base 0)
@pytest.mark.parametrize('near,far,mode', [(lo, hi, 1) for lo, hi in DEPTH_RANGES] + [(lo, hi, m) for m in (3, 2, 0) for lo, hi in DEPTH_RANGES[:4:3]])
def test_depth_sort_ranges_and_modes_sim(near, far, mode):
    be = helpers.sim_backend()
    assert be.lib.fgs_debug_set_option(9, mode) == 0
    try:
        for n, distinct in ((0, None), (1, None), (2047, None), (4100, 37)):
            _check_depth(be, n, near, far, 'cpu', 7 * n + mode, distinct)
    finally:
        be.lib.fgs_debug_set_option(9, 1)


def test_depth_sort_many_workgroups_sim():
    _check_depth(helpers.sim_backend(), 30_001, 0.2, 1e4, 'cpu', 5)                # 15 workgroups of 2048


@pytest.mark.gpu
@pytest.mark.parametrize('n', [4097, 2_049_194, 17_000_000])
def test_depth_sort_on_device_product(hip_backend, n):
    """The product library's depth sort (key - bits(near) in 9-bit passes, 4096-item workgroups)."""
    _check_depth(hip_backend, n, 0.2, 1e4, 'cuda', n + 1)
    if n == 2_049_194:
        _check_depth(hip_backend, n, 0.2, 1e4, 'cuda', 3, n_distinct=1000)   # heavy duplicates: stability
        _check_depth(hip_backend, n, 0.0, 1e4, 'cuda', 4)                    # base 0: 31 bits, four passes


@pytest.mark.gpu
@pytest.mark.parametrize('mode', [3, 2, 0])
@pytest.mark.parametrize('n', [4097, 2_049_194, 17_000_000])
def test_depth_sort_on_device(hip_dev_backend, n, mode):
    """The A/B modes of the dev library (fgs_debug_set_option(9, mode))."""
    hip_backend = hip_dev_backend
    assert hip_backend.lib.fgs_debug_set_option(9, mode) == 0
    try:
        _check_depth(hip_backend, n, 0.2, 1e4, 'cuda', n + mode)
        if n == 2_049_194:
            _check_depth(hip_backend, n, 0.2, 1e4, 'cuda', 3, n_distinct=1000)   # heavy duplicates: stability
            _check_depth(hip_backend, n, 0.0, 1e4, 'cuda', 4)                    # base 0: 31 bits, four passes
    finally:
        hip_backend.lib.fgs_debug_set_option(9, 1)

