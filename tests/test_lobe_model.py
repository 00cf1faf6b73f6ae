"""The closed-form Hann lobe of k_l1_env_wf (l1_kernels.hip hann_lobe_fast) restated in numpy float32 against the exact sum
of three Dirichlet kernels that llsm_harmonic_spectrum draws (dsputils.c:433-456; oracle/l1_oracle.c o_harmonic_spectrum):
the device code is checked through the layer-1 parity tests on the GPU, this pins the algebra and the dropped-term bound
(T >= 64) on the CPU."""
import numpy as np

F32 = np.float32
COEF = [1.4842879303107100e-04, -2.3460810354558236e-03, 2.6147847817654800e-02, -1.9075182412208421e-01,
        8.1174242528335364e-01, -1.6449340668482264e+00, 1.0]              # pi^(2n) / (2n + 1)!, alternating, as in the kernel


def lobe_fast(ud, T):
    kd = np.rint(ud)
    u, v, k = F32(ud), F32(ud - kd), kd.astype(int)
    w = v * v
    sc = np.full_like(w, F32(COEF[0]))
    for c in COEF[1:]:
        sc = (sc.astype(np.float64) * w + c).astype(F32)                     # fmaf
    snpi = np.where(k & 1, -v * sc, v * sc).astype(F32)
    om, op = F32(1) - u, F32(1) + u
    num, den = snpi.copy(), (u * om * op).astype(F32)
    for kk, n_, d_ in ((0, sc, om * op), (1, sc, u * op), (-1, -sc, u * om)):
        m = k == kk
        num[m] = n_[m]; den[m] = d_[m]
    c3 = F32(7.0 / 120.0 * np.pi ** 4 / float(T) ** 3)
    return F32(0.5) * (F32(T) * num / den - c3 * u * snpi)


def lobe_exact(ud, T):
    dt, sn = ud / T, np.sin(np.pi * ud)

    def kern(x, sign):
        s = np.sin(np.pi * x)
        with np.errstate(all="ignore"):
            r = sign * sn / s
        return np.where(np.abs(s) < 1e-300, float(T), r)
    return 0.5 * kern(dt, 1) + 0.25 * kern(dt - 1.0 / T, -1) + 0.25 * kern(dt + 1.0 / T, -1)


def test_closed_form_lobe_matches_the_dirichlet_sum():
    rng = np.random.default_rng(0)
    for T in (64, 100, 367, 1102, 9600):                                   # 64 = L1_LOBE_FAST_MIN_T
        ud = np.concatenate([rng.uniform(-4.5, 4.5, 100000), [0, 1, -1, 1e-9, 1 - 1e-9, -1 + 1e-9, 0.5, -0.5, 1.5, 2, 3, 4],
                             rng.normal(0, 1e-5, 500), 1 + rng.normal(0, 1e-5, 500), -1 + rng.normal(0, 1e-5, 500)])
        a, b = lobe_fast(ud, T).astype(np.float64), lobe_exact(ud, T)
        assert np.max(np.abs(a - b)) / (T / 2) < 4e-7, T                    # of the lobe's peak T / 2
        main = np.abs(ud) <= 1.5                                           # where a lobe can be the maximum at its bin
        assert np.max(np.abs(a - b)[main] / np.abs(b[main])) < 1e-6, T
        big = np.abs(b) > 1e-3 * T / 2
        assert np.max(np.abs(a - b)[big] / np.abs(b[big])) < (4e-6 if T == 64 else 1e-6), T


def test_peaks_of_the_three_kernels():
    for T in (64, 1102):
        assert lobe_fast(np.array([0.0]), T)[0] == F32(T) / 2              # D(0) / 2
        assert abs(lobe_fast(np.array([1.0]), T)[0] - T / 4) < 1e-4 * T     # D(0) / 4 of the shifted kernels
        assert abs(lobe_fast(np.array([-1.0]), T)[0] - T / 4) < 1e-4 * T
