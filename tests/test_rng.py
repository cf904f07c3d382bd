"""The in-library noise generator (k_sample.hip: Philox4x32-10 + Box-Muller, keyed by (seed, sample id, step, component)) --
what ddmi_sample / ddmi_perturb draw when the caller supplies no noise (the reference draws torch.normal at
utils/sampling.py:140-154; its stream cannot be reproduced, so the generator is pinned as a generator):

* Random123's known-answer vectors for philox4x32-10 through the DEVICE code (ddmi_debug_philox), next to an independent
  python restatement of the round function;
* the key -> draw mapping of normal_draw restated in numpy (counter = (sample lo, sample hi, step, component), key = seed);
* moments, lag correlations across components / samples / steps and a Kolmogorov-Smirnov distance over 1e6 draws (GPU)."""
import ctypes as C
import os

import numpy as np
import pytest

from diffdock_amd import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "hipemu", "libddmi_emu.so")

# Random123 kat_vectors: philox4x32 10  counter[4] key[2]  ->  expected[4]
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def philox4x32_10(ctr, key):
    """Plain-python Philox4x32-10 (Salmon et al., SC'11): ten rounds of two 32x32 -> 64 multiplies, key bumped by the Weyl constants."""
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c0, 0xCD9E8D57 * c2
        c0, c1, c2, c3 = (p1 >> 32) ^ c1 ^ k0, p1 & 0xffffffff, (p0 >> 32) ^ c3 ^ k1, p0 & 0xffffffff
        k0, k1 = (k0 + 0x9E3779B9) & 0xffffffff, (k1 + 0xBB67AE85) & 0xffffffff
    return c0, c1, c2, c3


def device_philox(lib, ctrs, keys):
    c = np.asarray(ctrs, dtype=np.uint32).reshape(-1, 4)
    k = np.asarray(keys, dtype=np.uint32).reshape(-1, 2)
    out = np.zeros_like(c)
    L.check(lib, lib.ddmi_debug_philox(c.ctypes.data, k.ctypes.data, c.shape[0], out.ctypes.data))
    return out


def test_python_restatement_reproduces_the_known_answers():
    for ctr, key, want in KAT:
        assert philox4x32_10(ctr, key) == want


@pytest.fixture(scope="module")
def emu():
    import subprocess
    r = subprocess.run(["make", "-j8", "-C", os.path.join(ROOT, "diffdock_amd", "csrc"), "emu"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return L.load(EMU)


def check_kat_and_random_blocks(lib):
    got = device_philox(lib, [k[0] for k in KAT], [k[1] for k in KAT])
    for row, (_, _, want) in zip(got, KAT):
        assert tuple(int(x) for x in row) == want
    rng = np.random.default_rng(0)
    ctrs = rng.integers(0, 2 ** 32, size=(64, 4), dtype=np.uint64).astype(np.uint32)
    keys = rng.integers(0, 2 ** 32, size=(64, 2), dtype=np.uint64).astype(np.uint32)
    got = device_philox(lib, ctrs, keys)
    for c, k, g in zip(ctrs, keys, got):
        assert tuple(int(x) for x in g) == philox4x32_10([int(x) for x in c], [int(x) for x in k])


def test_kernel_sources_reproduce_the_known_answers_under_the_emulator(emu):
    check_kat_and_random_blocks(emu)


def normal_from_block(o0, o1):
    u1 = ((o0 >> 8).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
    u2 = ((o1 >> 8).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
    return np.sqrt(-2.0 * np.log(u1.astype(np.float64))) * np.cos(6.283185307179586 * u2.astype(np.float64))


@pytest.mark.gpu
def test_device_generator_known_answers_keying_and_moments():
    import torch
    lib = L.load()
    check_kat_and_random_blocks(lib)
    seed, step, n_s, n_c = 0x1234567890abcdef, 7, 125000, 8          # 1e6 draws: sample ids [2^33, 2^33 + n_s) x 8 components
    s0 = 2 ** 33
    out = torch.empty(n_s, n_c, device="cuda:0")
    L.check(lib, lib.ddmi_debug_normal(seed, s0, n_s, step, n_c, out.data_ptr(), torch.cuda.current_stream().cuda_stream))
    z = out.cpu().numpy().astype(np.float64)
    # keying: counter = (sample lo, sample hi, step, component), key = (seed lo, seed hi); first two words -> Box-Muller
    for (i, c) in ((0, 0), (17, 3), (n_s - 1, n_c - 1)):
        sid = s0 + i
        o = philox4x32_10((sid & 0xffffffff, sid >> 32, step, c), (seed & 0xffffffff, seed >> 32))
        want = normal_from_block(np.array([o[0]], dtype=np.uint32), np.array([o[1]], dtype=np.uint32))[0]
        assert abs(z[i, c] - want) < 2e-5 * max(1.0, abs(want)), (i, c, z[i, c], want)
    n = z.size
    assert np.isfinite(z).all()
    assert abs(z.mean()) < 4.0 / np.sqrt(n)                            # 4 sigma of the sample mean
    assert abs(z.var() - 1.0) < 4.0 * np.sqrt(2.0 / n)
    assert abs((z ** 3).mean()) < 4.0 * np.sqrt(15.0 / n)              # skewness 0
    assert abs((z ** 4).mean() - 3.0) < 4.0 * np.sqrt(96.0 / n)        # kurtosis 3
    bound = 4.0 / np.sqrt(n)
    assert abs((z[:, :-1] * z[:, 1:]).mean()) < bound * 1.1            # neighbouring components of one sample
    assert abs((z[:-1] * z[1:]).mean()) < bound * 1.1                  # neighbouring samples
    out2 = torch.empty(n_s, n_c, device="cuda:0")
    L.check(lib, lib.ddmi_debug_normal(seed, s0, n_s, step + 1, n_c, out2.data_ptr(), torch.cuda.current_stream().cuda_stream))
    assert abs((z * out2.cpu().numpy()).mean()) < bound * 1.1          # consecutive steps
    # Kolmogorov-Smirnov distance to the standard normal
    from scipy.special import ndtr
    zs = np.sort(z.reshape(-1))
    cdf = ndtr(zs)
    d = max(np.abs(cdf - np.arange(1, n + 1) / n).max(), np.abs(cdf - np.arange(0, n) / n).max())
    assert d < 1.95 / np.sqrt(n)                                       # 0.1 % critical value
    # float32 Box-Muller: the largest magnitude a 24-bit uniform can produce is sqrt(-2 ln(2^-25)) = 5.89
    assert np.abs(z).max() < 5.9
