"""Wavefront FFT (csrc/wave_fft.h) against numpy, through the C-ABI diagnostic entry point."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import libllsm2_amd as llsm
    c = llsm.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("logn", [8, 9, 10, 11, 12])
@pytest.mark.parametrize("inverse", [False, True])
def test_wave_fft_matches_numpy(ctx, logn, inverse):
    rng = np.random.default_rng(100 + logn)
    n = 1 << logn
    z = (rng.standard_normal((37, n)) + 1j * rng.standard_normal((37, n))).astype(np.complex64)
    z[0] = 0; z[0, 1] = 1.0                      # a pure twiddle row
    z[1] = 1.0                                   # DC
    got = ctx.fft_selftest(z, inverse=inverse)
    ref = np.fft.ifft(z.astype(np.complex128), axis=1) * n if inverse else np.fft.fft(z.astype(np.complex128), axis=1)
    scale = np.sqrt(np.mean(np.abs(ref) ** 2, axis=1, keepdims=True))
    err = np.max(np.abs(got - ref) / scale)
    # float32 transform of 2^logn points: rounding grows ~ sqrt(log N); twiddles are exact to ~3e-7
    assert err < 4e-6, err
    rms = np.sqrt(np.mean(np.abs(got - ref) ** 2) / np.mean(np.abs(ref) ** 2))
    assert rms < 6e-7, rms
