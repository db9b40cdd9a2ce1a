"""oracle/dropout_rng.py (numpy) == the hash the HIP kernels use (host-evaluated through the C ABI)."""
import ctypes

import numpy as np
import pytest

from oracle import dropout_rng as R
from pixelrec_amd import lib


@pytest.mark.parametrize("seed,stream,first,p", [(2020, 0, 0, 0.1), (2020 * 1000003 + 17, 4, 123456, 0.1),
                                                   (0xFFFFFFFFFFFFFFFF, 6, 2 ** 33 + 5, 0.5), (7, 1, 0, 0.0)])
def test_mask_matches_library(seed, stream, first, p):
    L = lib.load()
    n = 20000
    buf = (ctypes.c_uint8 * n)()
    assert L.pxr_dropout_keep_host(seed, stream, first, n, p, ctypes.cast(buf, ctypes.c_void_p)) == 0
    got = np.frombuffer(buf, dtype=np.uint8).astype(bool)
    ref = R.hash32(seed, stream, np.arange(first, first + n, dtype=np.uint64)) >= np.uint32(R.drop_threshold(p))
    assert np.array_equal(got, ref)
    assert abs(got.mean() - (1 - p)) < 0.02          # Bernoulli(1-p) keep rate


def test_streams_and_seeds_decorrelate():
    a = R.keep_mask(1, 0, (100000,), 0.5)
    b = R.keep_mask(1, 1, (100000,), 0.5)
    c = R.keep_mask(2, 0, (100000,), 0.5)
    assert abs((a == b).mean() - 0.5) < 0.01 and abs((a == c).mean() - 0.5) < 0.01
