"""CPU tests that pin the oracle itself: analytic known-answer cases for the Keras
semantics that are easy to get wrong (SURVEY.md 7 'hard parts'), an independent
torch implementation, the plain-C restatement, and the committed golden vectors."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT


def test_same_padding_rule(oracle):
    # TF SAME: stride 2, 10 -> 5 pads 0 before / 1 after; stride 1 pads 1/1
    assert oracle.same_padding(10, 3, 2) == (5, 0, 1)
    assert oracle.same_padding(5, 3, 1) == (5, 1, 1)
    assert oracle.same_padding(400, 3, 1) == (400, 1, 1)
    assert oracle.same_padding(7, 3, 2) == (4, 1, 1)
    assert oracle.same_padding(9, 3, 2) == (5, 1, 1)


def test_conv_same_stride2_is_bottom_right_padded(oracle):
    """one-hot taps: tap (0,0) must read x[2oy,2ox]; tap (2,2) reads x[2oy+2,2ox+2]
    which is zero padding for the last row/col (symmetric padding gets this wrong)."""
    x = np.arange(100, dtype=np.float64).reshape(1, 10, 10, 1) + 1
    for (ky, kx) in [(0, 0), (1, 1), (2, 2), (0, 2)]:
        w = np.zeros((3, 3, 1, 1))
        w[ky, kx, 0, 0] = 1
        y = oracle.conv2d(x, w, np.zeros(1), 2, "same", "linear")[0, :, :, 0]
        for oy in range(5):
            for ox in range(5):
                iy, ix = 2 * oy + ky, 2 * ox + kx
                exp = x[0, iy, ix, 0] if iy < 10 and ix < 10 else 0.0
                assert y[oy, ox] == exp


def test_conv_transpose_scatter_no_flip(oracle):
    """single input pixel -> the kernel itself lands at out[2i+a, 2j+b, co] (no flip);
    two neighbours overlap in exactly one column for k=3, s=2."""
    rng = np.random.default_rng(0)
    w = rng.standard_normal((3, 3, 2, 3))  # (kh,kw,Cout,Cin)
    x = np.zeros((1, 2, 2, 3))
    x[0, 0, 0, 1] = 1.0
    y = oracle.conv2d_transpose(x, w, np.zeros(2), 2, "valid", "linear")
    assert y.shape == (1, 5, 5, 2)
    np.testing.assert_allclose(y[0, :3, :3, :], w[:, :, :, 1], rtol=0, atol=0)
    assert np.all(y[0, 3:, :, :] == 0) and np.all(y[0, :, 3:, :] == 0)
    x[0, 0, 1, 2] = 2.0
    y2 = oracle.conv2d_transpose(x, w, np.zeros(2), 2, "valid", "linear")
    exp = np.zeros((5, 5, 2))
    exp[:3, :3] += w[:, :, :, 1]
    exp[:3, 2:5] += 2.0 * w[:, :, :, 2]
    np.testing.assert_allclose(y2[0], exp, atol=1e-15)


def test_conv_transpose_2x2_is_pixel_shuffle_of_per_pixel_gemm(oracle):
    rng = np.random.default_rng(1)
    w = rng.standard_normal((2, 2, 4, 6))
    b = rng.standard_normal(4)
    x = rng.standard_normal((2, 3, 5, 6))
    y = oracle.conv2d_transpose(x, w, b, 2, "valid", "linear")
    assert y.shape == (2, 6, 10, 4)
    for a in range(2):
        for bb in range(2):
            np.testing.assert_allclose(y[:, a::2, bb::2, :], x @ w[a, bb].T + b, atol=1e-13)


def test_flatten_and_reshape_are_nhwc_views(oracle, enc_weights):
    rng = np.random.default_rng(2)
    x = rng.standard_normal((1, 10, 10, 1))
    z, acts = oracle.encoder_forward(x, enc_weights, np.float64, True)
    a2 = acts[1]
    f = a2.reshape(1, -1)
    assert f[0, (3 * 5 + 2) * 128 + 7] == a2[0, 3, 2, 7]


def test_numpy_oracle_matches_torch_float64(oracle, enc_weights, dec_weights):
    from oracle.sr_oracle_torch import TorchSR
    rng = np.random.default_rng(3)
    x = rng.standard_normal((3, 10, 10, 1)).astype(np.float32)
    y = oracle.superres_forward(x, enc_weights, dec_weights, np.float64)
    yt = TorchSR(enc_weights, dec_weights, torch.float64).forward(x)
    assert oracle.rel_l2(yt, y) < 1e-13
    # glorot-initialised decoder too (Keras' default initialiser)
    dg = oracle.synthetic_decoder(7, init="glorot")
    assert oracle.rel_l2(TorchSR(enc_weights, dg, torch.float64).forward(x), oracle.superres_forward(x, enc_weights, dg, np.float64)) < 1e-13


def test_float32_restatements_agree_with_float64(oracle, enc_weights, dec_weights):
    from oracle.sr_oracle_torch import TorchSR
    rng = np.random.default_rng(4)
    x = rng.standard_normal((2, 10, 10, 1)).astype(np.float32)
    ref = oracle.superres_forward(x, enc_weights, dec_weights, np.float64)
    assert oracle.rel_l2(oracle.superres_forward(x, enc_weights, dec_weights, np.float32), ref) < 5e-6
    assert oracle.rel_l2(TorchSR(enc_weights, dec_weights, torch.float32).forward(x), ref) < 5e-6


@pytest.fixture(scope="module")
def c_oracle():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libsr_oracle.so"))
    lib.sr_oracle_forward_f32.restype = ctypes.c_int
    return lib


def _ptrs(ws, names):
    arrs = []
    for n in names:
        arrs += [np.ascontiguousarray(ws[f"{n}/kernel"], np.float32), np.ascontiguousarray(ws[f"{n}/bias"], np.float32)]
    P = ctypes.POINTER(ctypes.c_float)
    return arrs, (P * len(arrs))(*[a.ctypes.data_as(P) for a in arrs])


def test_c_oracle_matches_numpy(c_oracle, oracle, enc_weights, dec_weights):
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 10, 10, 1)).astype(np.float32)
    ka, pe = _ptrs(enc_weights, oracle.ENCODER_LAYERS)
    kb, pd = _ptrs(dec_weights, oracle.DECODER_LAYERS)
    y = np.empty((2, 400, 400, 1), np.float32)
    z = np.empty((2, 50), np.float32)
    P = ctypes.POINTER(ctypes.c_float)
    rc = c_oracle.sr_oracle_forward_f32(x.ctypes.data_as(P), 2, pe, pd, y.ctypes.data_as(P), z.ctypes.data_as(P))
    assert rc == 0
    assert oracle.rel_l2(z, oracle.encoder_forward(x, enc_weights, np.float64)) < 5e-6
    assert oracle.rel_l2(y, oracle.superres_forward(x, enc_weights, dec_weights, np.float64)) < 5e-6


def test_pre_post_helpers_match_reference_semantics(oracle):
    x = np.array([[1.0, 2.0], [3.0, 5.0]], np.float32)
    assert oracle.standardize_with_stats(x, 1.0, 0).max() == np.float32(4.0) / np.float32(1e-8)
    out = oracle.standardize_with_stats(x, 0.5, 2.0)
    assert out.dtype == np.float32  # python-float stats keep the float32 array dtype (NEP 50)
    m, s = oracle.adaptive_blend(x, 0.2, 0.3, 0.3)
    assert m == pytest.approx(0.7 * 0.2 + 0.3 * float(np.mean(x)))
    assert s == pytest.approx(0.7 * 0.3 + 0.3 * float(np.std(x)))
    _, s0 = oracle.adaptive_blend(np.zeros((2, 2), np.float32), 0.0, 1.0, 1.0)
    assert s0 == 1e-8
    arr = np.array([1.0, np.nan, np.inf, -np.inf], np.float32)
    g, nn, ni = oracle.nan_guard(arr)
    assert (nn, ni) == (1, 2) and g.tolist() == [1.0, 0.0, 0.0, 0.0]


def test_golden_vectors(oracle, enc_weights, dec_weights):
    g = np.load(os.path.join(GOLDEN, "golden_vectors.npz"))
    z = oracle.encoder_forward(g["x_std"], enc_weights, np.float64)
    np.testing.assert_allclose(z, g["latent_std"], rtol=1e-12, atol=1e-12)
    zb = oracle.encoder_forward(g["x_blend"], enc_weights, np.float64)
    np.testing.assert_allclose(zb, g["latent_blend"], rtol=1e-12, atol=1e-12)
    y = oracle.superres_forward(g["x_std"][:3], enc_weights, dec_weights, np.float64)[..., 0]
    idx = g["probe_idx"]
    np.testing.assert_allclose(y[:, idx[:, 0], idx[:, 1]], g["probe_val"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(np.sqrt((y ** 2).sum(axis=(1, 2))), g["probe_l2"], rtol=1e-12)
    for name, cs in zip(g["enc_names"], g["enc_checksum"]):
        assert enc_weights[str(name)].astype(np.float64).sum() == cs


def test_all_three_trained_encoders_load_and_match_their_golden_latents(srcfd, oracle):
    """Every trained weight set the reference checkout holds (its decoder files are absent): the legacy-H5 reader on all three
    files -- one of the names carries parentheses, a '+' and a blank -- the stats file its suffix selects, and the float64 oracle's
    latents on the 15 real coarse fields against the committed vectors (the three sets are different trainings: their latents differ)."""
    import torch
    from conftest import ENCODER_SETS
    from oracle.sr_oracle_torch import TorchSR  # noqa: F401  (second implementation, used below through encoder_forward_torch)
    g = np.load(os.path.join(GOLDEN, "golden_vectors.npz"))
    seen = []
    for key, (h5, txt) in ENCODER_SETS.items():
        m = srcfd.SRModel.load_h5(os.path.join(GOLDEN, h5), None, device=-1)
        w = m.weights()
        assert m.input_shape == (10, 10, 1) and m.output_shape == (1, 1, 50) and len(w) == 8
        for name, cs in zip(sorted(w), g[f"enc3_checksum_{key}"]):
            assert w[name].astype(np.float64).sum() == cs
        lr, hr = srcfd.load_stats(os.path.join(GOLDEN, txt), 10, 400)
        o_lr, _ = oracle.component_stats(oracle.parse_stats(os.path.join(GOLDEN, txt)), 10, 400)
        assert lr == o_lr and all(s > 0 for _, s in lr.values())
        z = oracle.encoder_forward(g[f"x3_{key}"], w, np.float64)
        np.testing.assert_allclose(z, g[f"latent3_{key}"], rtol=1e-12, atol=1e-12)
        seen.append(g[f"latent3_{key}"])
    assert not np.allclose(seen[0], seen[1], atol=1e-3) and not np.allclose(seen[0], seen[2], atol=1e-3)
    np.testing.assert_array_equal(g["latent3_multiBC"], g["latent_std"])      # the multiBC set is the one the other fixtures use


def test_work_per_sample_matches_survey(srcfd, oracle, enc_weights, dec_weights):
    m = srcfd.SRModel.from_weights(enc_weights, dec_weights, device=-1)
    assert m.macs_per_sample == oracle.MACS_PER_SAMPLE == 140_024_128
    assert m.has_fused_path
    assert m.input_shape == (10, 10, 1) and m.output_shape == (400, 400, 1)
