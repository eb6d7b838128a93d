"""The drop-in surface: Keras-compatible `load_model` / `Model.predict`, the stand-in
`tensorflow` package, and the `ml_super_resolution` harness."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ENCODER_H5, ROOT, STATS_TXT, require_gpu

COMPAT = os.path.join(ROOT, "sr-for-cfd_amd", "compat")


@pytest.fixture(scope="module")
def decoder_h5(srcfd, dec_weights, tmp_path_factory):
    """Synthetic decoder written in the legacy Keras-H5 layout by libsrcfd's own writer."""
    p = tmp_path_factory.mktemp("models") / "vanilla_decoder400_from_10_synthetic.h5"
    srcfd.SRModel.from_weights(None, dec_weights, device=-1).save_h5(None, str(p))
    return str(p)


def _reference_script_class():
    """The user-side class exactly as the solver scripts define it (PyCFD...:676-689),
    importing the names the scripts import."""
    code = (
        "import tensorflow as tf\n"
        "from tensorflow.keras import Model\n"
        "class SuperResolutionAE(Model):\n"
        "    def __init__(self, encoder_lr, decoder_hr, **kwargs):\n"
        "        super().__init__(**kwargs)\n"
        "        self.encoder_lr = encoder_lr\n"
        "        self.decoder_hr = decoder_hr\n"
        "    def call(self, inputs, training=False):\n"
        "        z = self.encoder_lr(inputs, training=training)\n"
        "        recon_hr = self.decoder_hr(z, training=training)\n"
        "        return recon_hr\n")
    return code


def test_stand_in_tensorflow_imports_and_loads(decoder_h5):
    """In a fresh interpreter with compat/ on the path the script-side imports resolve, sub-models
    load with the right shapes, and errors follow the reference's conventions."""
    code = _reference_script_class() + (
        f"enc = tf.keras.models.load_model({ENCODER_H5!r}, compile=False)\n"
        f"dec = tf.keras.models.load_model({decoder_h5!r}, compile=False)\n"
        "assert enc.input_shape == (None, 10, 10, 1) and enc.output_shape == (None, 50), (enc.input_shape, enc.output_shape)\n"
        "assert dec.input_shape == (None, 50) and dec.output_shape == (None, 400, 400, 1)\n"
        "assert enc.count_params() == 490674 and dec.count_params() == 2218817\n"
        "m = SuperResolutionAE(enc, dec)\n"
        "assert [c.path for c in m._chain(None)] == [enc.path, dec.path]\n"
        "try:\n    tf.keras.models.load_model('/no/such/file.h5', compile=False)\n"
        "except (IOError, OSError) as e:\n    print('missing ->', type(e).__name__)\n"
        "print('ok')\n")
    env = dict(os.environ, PYTHONPATH=COMPAT)
    out = subprocess.check_output([sys.executable, "-c", code], env=env, text=True)
    assert "missing -> FileNotFoundError" in out and out.strip().endswith("ok")


def test_call_with_tensor_arithmetic_is_rejected(srcfd):
    kc = importlib.import_module("sr-for-cfd_amd.keras_compat")
    enc = kc.load_model(ENCODER_H5, compile=False)

    class Bad(kc.Model):
        def call(self, inputs, training=False):
            return enc(inputs) * 2.0

    with pytest.raises(NotImplementedError):
        Bad().predict(np.zeros((1, 10, 10, 1), np.float32))


def test_harness_errors_without_touching_the_gpu(srcfd, coarse_cases, decoder_h5, tmp_path):
    pl = importlib.import_module("sr-for-cfd_amd.pipeline")
    case = coarse_cases["ldc_Re800_double"]
    with pytest.raises(FileNotFoundError):
        pl.ml_super_resolution(case, 10, 400, str(tmp_path / "nope.txt"), ENCODER_H5, decoder_h5)
    with pytest.raises(KeyError):
        pl.ml_super_resolution(case, 10, 100, STATS_TXT, ENCODER_H5, decoder_h5)  # no mean100_* keys
    with pytest.raises(FileNotFoundError):
        pl.ml_super_resolution(case, 10, 400, STATS_TXT, ENCODER_H5, str(tmp_path / "dec.h5"))


def test_tiled_layout_round_trip(srcfd):
    """config-5 tiling: tiles are cut and stitched consistently (identity 'model')."""
    pl = importlib.import_module("sr-for-cfd_amd.pipeline")

    class Up:  # nearest-neighbour x2 stand-in with the SRModel.predict signature
        def predict(self, x, in_affine=None, out_affine=None):
            return np.repeat(np.repeat(x, 2, axis=1), 2, axis=2)

    f = np.arange(40 * 40 * 3, dtype=np.float32).reshape(40, 40, 3)
    y = pl.tiled_super_resolution(f, Up(), lr_dim=10)
    np.testing.assert_array_equal(y, np.repeat(np.repeat(f, 2, axis=0), 2, axis=1))


def test_handle_cache_follows_the_file_and_the_device(srcfd, dec_weights, tmp_path, monkeypatch):
    """The Keras-style surface keeps ONE device handle per (files, precision, device): a re-saved weight file (new mtime)
    closes and replaces the stale handle rather than leaking it, and the device comes from set_default_device /
    LOCAL_RANK rather than always being GPU 0.  Host-only handles here (no GPU in this container)."""
    kc = importlib.import_module("sr-for-cfd_amd.keras_compat")
    kc.clear_handle_cache()
    p = str(tmp_path / "vanilla_decoder400_from_10_cache.h5")
    srcfd.SRModel.from_weights(None, dec_weights, device=-1).save_h5(None, p)
    h1 = kc._device_handle((p,), "fp32")
    assert kc._device_handle((p,), "fp32") is h1 and len(kc._HANDLE_CACHE) == 1
    w2 = {k: v * 2 for k, v in dec_weights.items()}
    srcfd.SRModel.from_weights(None, w2, device=-1).save_h5(None, p)
    os.utime(p, (os.path.getatime(p), os.path.getmtime(p) + 5))
    h2 = kc._device_handle((p,), "fp32")
    assert h2 is not h1 and h1._h is None and len(kc._HANDLE_CACHE) == 1          # stale handle closed, not kept
    np.testing.assert_array_equal(h2.weights()["dense_1/bias"], w2["dense_1/bias"])
    # device choice: explicit > LOCAL_RANK > 0, and always -1 on a box without GPUs
    monkeypatch.setattr(kc, "device_count", lambda: 8)
    monkeypatch.setenv("LOCAL_RANK", "5")
    assert kc._pick_device() in (5, )
    kc.set_default_device(3)
    assert kc._pick_device() == 3
    kc.set_default_device(None)
    monkeypatch.setattr(kc, "device_count", lambda: 0)
    assert kc._pick_device() == -1
    kc.clear_handle_cache()
    assert h2._h is None


# ---------------------------------------------------------------- GPU ------------
@pytest.mark.gpu
def test_predict_through_user_subclass_matches_oracle(srcfd, oracle, enc_weights, dec_weights, decoder_h5, coarse_cases):
    require_gpu(srcfd)
    kc = importlib.import_module("sr-for-cfd_amd.keras_compat")

    class SuperResolutionAE(kc.Model):
        def __init__(self, encoder_lr, decoder_hr, **kwargs):
            super().__init__(**kwargs)
            self.encoder_lr, self.decoder_hr = encoder_lr, decoder_hr

        def call(self, inputs, training=False):
            return self.decoder_hr(self.encoder_lr(inputs, training=training), training=training)

    enc = kc.load_model(ENCODER_H5, compile=False)
    dec = kc.load_model(decoder_h5, compile=False)
    model = SuperResolutionAE(enc, dec)
    lr, _ = srcfd.load_stats(STATS_TXT, 10, 400)
    x = ((coarse_cases["ldc_Re1000_double"]["u"].astype(np.float32) - lr["u"][0]) / lr["u"][1]).astype(np.float32)
    xb = np.expand_dims(x, axis=(0, -1))
    y = model.predict(xb, verbose=0)[0, ..., 0]  # the reference's exact call shape, PyCFD...:855-858
    ref = oracle.superres_forward(xb, enc_weights, dec_weights, np.float64)[0, ..., 0]
    assert y.shape == (400, 400) and y.dtype == np.float32
    assert oracle.rel_l2(y[None], ref[None]) <= 1e-5
    z = enc.predict(xb)  # a loaded sub-model predicts on its own too
    assert z.shape == (1, 50)
    assert oracle.rel_l2(z, oracle.encoder_forward(xb, enc_weights, np.float64)) <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["ldc_Re800_double", "ldc_Re1000_single"])
def test_harness_ldc_matches_oracle(srcfd, oracle, enc_weights, dec_weights, decoder_h5, coarse_cases, case):
    require_gpu(srcfd)
    pl = importlib.import_module("sr-for-cfd_amd.pipeline")
    out = pl.ml_super_resolution(coarse_cases[case], 10, 400, STATS_TXT, ENCODER_H5, decoder_h5)
    ref = oracle.ml_super_resolution(coarse_cases[case], 10, 400, oracle.parse_stats(STATS_TXT), enc_weights, dec_weights,
                                     dtype=np.float32, net_dtype=np.float64)
    _, hr = srcfd.load_stats(STATS_TXT, 10, 400)
    for c in ("u", "v", "p"):
        assert out[c].shape == (400, 400) and out[c].dtype == np.float32
        a = (out[c] - hr[c][0]) / hr[c][1]
        b = (ref[c] - hr[c][0]) / hr[c][1]
        assert oracle.rel_l2(a[None], b[None]) <= 1e-5
    # the consumer side: transposed injection into the solver's float64 state (PyCFD...:936-938)
    Var = np.zeros((3, 402, 402))
    pl.inject_into_solver_state(out, Var)
    assert Var[0, 1 + 7, 1 + 3] == out["u"][3, 7] and Var[2, 400, 1] == out["p"][0, 399]


@pytest.mark.gpu
def test_config1_ldc_re400_from_our_coarse_solver(srcfd, oracle, enc_weights, dec_weights, decoder_h5):
    """BASELINE config 1 end to end: the coarse 10x10 double-lid field at Re = 400 from this repo's coarse solver (the
    reference checkout has none), through `ml_super_resolution` exactly as PyCFD_ML_accelerated.py:1425-1445 calls it, f32."""
    require_gpu(srcfd)
    from conftest import COARSE_RE400, GOLDEN
    pl = importlib.import_module("sr-for-cfd_amd.pipeline")
    co = importlib.import_module("sr-for-cfd_amd.coarse")
    fresh = co.run_coarse_simulation(400.0, 10, bc=co.LDC_DOUBLE_LID)
    stored = srcfd.read_coarse_fields(os.path.join(GOLDEN, COARSE_RE400))
    for c in "uvp":
        np.testing.assert_array_equal(fresh[c], stored[c])
    out = pl.ml_super_resolution(fresh, 10, 400, STATS_TXT, ENCODER_H5, decoder_h5)
    ref = oracle.ml_super_resolution(fresh, 10, 400, oracle.parse_stats(STATS_TXT), enc_weights, dec_weights,
                                     dtype=np.float32, net_dtype=np.float64)
    _, hr = srcfd.load_stats(STATS_TXT, 10, 400)
    for c in "uvp":
        assert out[c].shape == (400, 400) and np.isfinite(out[c]).all()
        a, b = (out[c] - hr[c][0]) / hr[c][1], (ref[c] - hr[c][0]) / hr[c][1]
        assert oracle.rel_l2(a[None], b[None]) <= 1e-5


@pytest.mark.gpu
def test_harness_bfs_with_resampling_and_blend_matches_reference_recipe(srcfd, oracle, enc_weights, dec_weights, decoder_h5, coarse_cases):
    """BASELINE config 3: BFS Re400 coarse field, aspect-ratio correction (lx=10, ly=3) and
    adaptive normalisation (blend 0.3), bfs_ml_accelerated.py:1473 call."""
    require_gpu(srcfd)
    pl = importlib.import_module("sr-for-cfd_amd.pipeline")
    case = coarse_cases["bfs_Re400"]
    out = pl.ml_super_resolution_bfs(case, 10, 400, STATS_TXT, ENCODER_H5, decoder_h5,
                                     use_aspect_ratio_correction=True, lx=10.0, ly=3.0, blend_factor=0.3)
    # reference recipe re-stated with the oracle + scipy
    sq = pl.reshape_rectangular_to_square(case, 10, 10, 10.0, 3.0)
    ref_sq = oracle.ml_super_resolution(sq, 10, 400, oracle.parse_stats(STATS_TXT), enc_weights, dec_weights,
                                        use_adaptive_normalization=True, blend_factor=0.3, dtype=np.float32, net_dtype=np.float64)
    ref = pl.reshape_square_to_rectangular(ref_sq, 400, 400, 10.0, 3.0)
    for c in ("u", "v", "p"):
        assert out[c].shape == (400, 400)
        scale = np.linalg.norm(ref[c])
        assert np.linalg.norm(out[c] - ref[c]) / scale <= 2e-5


@pytest.mark.gpu
def test_reserve_is_the_set_up_a_first_call_would_do(srcfd, enc_weights, dec_weights):
    """srcfd_model_reserve(n): the lazy set-up of a forward (16-bit operand packs, activation workspaces) done on request;
    results are those of an unreserved handle, a smaller or repeated reserve is a no-op, a larger one re-allocates."""
    require_gpu(srcfd)
    rng = np.random.default_rng(31)
    x = rng.standard_normal((7, 10, 10, 1)).astype(np.float32)
    for prec in ("bf16", "fp32"):
        a = srcfd.SRModel.from_weights(enc_weights, dec_weights, device=0)
        b = srcfd.SRModel.from_weights(enc_weights, dec_weights, device=0)
        a.precision = b.precision = prec
        b.reserve(7)
        b.reserve(3)
        b.reserve(0)
        ya = a.predict(x)
        np.testing.assert_array_equal(b.predict(x), ya)
        b.reserve(40)
        np.testing.assert_array_equal(b.predict(x), ya)
        np.testing.assert_array_equal(b.predict(np.concatenate([x] * 5))[7:14], ya)


@pytest.mark.gpu
def test_footprint_is_what_reserve_allocates(srcfd, enc_weights, dec_weights):
    """srcfd_model_footprint computes sizes without allocating (bench.py's capacity estimate for N ranks on one host rests on it):
    the device memory that reserve(n) actually takes is its activation workspace, up to the allocator's granularity, and the
    host-only handle gives the same numbers as the device handle."""
    require_gpu(srcfd)
    import torch
    host_only = srcfd.SRModel.from_weights(enc_weights, dec_weights, device=-1)
    for prec, n in (("bf16", 768), ("fp32", 768), ("f16", 48)):
        m = srcfd.SRModel.from_weights(enc_weights, dec_weights, device=0)
        m.precision = prec
        fp = m.footprint(n)
        assert fp == host_only.footprint(n, prec)
        m.reserve(1)                      # operand packs, streams, events: everything that does not scale with n
        torch.cuda.synchronize()
        free0, _ = torch.cuda.mem_get_info()
        m.reserve(n)
        torch.cuda.synchronize()
        free1, _ = torch.cuda.mem_get_info()
        took = free0 - free1
        one = m.footprint(1)["device_workspace"]
        want = fp["device_workspace"] - one
        print(f"{prec} n={n}: reserve took {took / 1e6:.1f} MB, footprint says {want / 1e6:.1f} MB (+ {one / 1e6:.2f} for n = 1)")
        assert abs(took - want) <= 0.02 * want + (64 << 20), (prec, took, want)    # hipMalloc rounds to its pool granularity
        assert fp["host_result"] == n * 160000 * 4
        del m


@pytest.mark.gpu
def test_config3_end_to_end_from_a_coarse_bfs_solve_made_here(srcfd, decoder_h5, coarse_cases):
    """BASELINE config 3 without any stored input: coarse backward-facing-step solve (csrc/coarse_solver.cpp, the reference's
    __main__ settings) -> ml_super_resolution with aspect-ratio correction and blend 0.3.  The coarse field is within the
    reference's own run-to-run spread of its stored one (tests/test_coarse_solver.py), so the two SR results agree to that."""
    require_gpu(srcfd)
    pl = importlib.import_module("sr-for-cfd_amd.pipeline")
    coarse = importlib.import_module("sr-for-cfd_amd.coarse")
    solved = coarse.run_bfs_coarse_simulation(400.0, 10, bc=coarse.BFS_DEFAULT)
    kw = dict(use_aspect_ratio_correction=True, lx=10.0, ly=3.0, blend_factor=0.3)
    a = pl.ml_super_resolution_bfs(solved, 10, 400, STATS_TXT, ENCODER_H5, decoder_h5, **kw)
    b = pl.ml_super_resolution_bfs(coarse_cases["bfs_Re400"], 10, 400, STATS_TXT, ENCODER_H5, decoder_h5, **kw)
    for c in "uvp":
        assert a[c].shape == (400, 400) and np.isfinite(a[c]).all()
        assert np.linalg.norm(a[c] - b[c]) / np.linalg.norm(b[c]) <= 1e-5


@pytest.mark.gpu
def test_batched_call_prepares_its_inputs_on_the_device(srcfd, decoder_h5, coarse_cases):
    """SURVEY 8f-2 / 8a row a3 on the device: the 10x10 aspect-ratio resampling, the float32 cast and the adaptive blend
    (np.mean / np.std of the float32 field, NumPy scalar promotion included) for a batch of coarse fields equal the host
    recipe's numbers -- the (mean, std) pairs bit for bit, the resampled inputs to one float32 ulp -- and the batched call
    returns what the per-field calls return."""
    import ctypes as C
    import torch
    require_gpu(srcfd)
    pl = importlib.import_module("sr-for-cfd_amd.pipeline")
    rs = importlib.import_module("sr-for-cfd_amd.resample")
    L = importlib.import_module("sr-for-cfd_amd._lib")
    rng = np.random.default_rng(77)
    base = coarse_cases["bfs_Re400"]
    batch = [base] + [{c: base[c] * (1 + 0.05 * rng.standard_normal()) + 0.01 * rng.standard_normal((10, 10)) for c in "uvp"} for _ in range(6)]
    batch.append({c: np.full((10, 10), 0.25) for c in "uvp"})          # zero variance: the max(input_std, 1e-8) branch
    lr_stats, _ = srcfd.load_stats(STATS_TXT, 10, 400)
    dev = torch.device("cuda", 0)
    fields = np.stack([np.stack([b[c] for c in "uvp"]) for b in batch]).reshape(-1, 10, 10)
    n = fields.shape[0]
    ry, rx = rs.rect_to_square_matrices(10, 10, 10.0, 3.0)
    for resample in (True, False):
        for adaptive in (1, 0):
            x = torch.empty((n, 10, 10), dtype=torch.float32, device=dev)
            ain = torch.empty((n, 2), dtype=torch.float32, device=dev)
            f_dev = torch.from_numpy(fields).to(dev)
            tr = torch.from_numpy(np.tile(np.array([lr_stats[c] for c in "uvp"], np.float64), (len(batch), 1))).to(dev)
            Ry = torch.from_numpy(np.array(ry)).to(dev) if resample else None
            Rx = torch.from_numpy(np.array(rx)).to(dev) if resample else None
            p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
            L.check(L.lib.srcfd_prepare_inputs_device(p(f_dev), n, 10, 10, p(Ry), p(Rx), 10, p(tr), adaptive, 0.3, p(x), p(ain), None))
            torch.cuda.synchronize()
            xg, ag = x.cpu().numpy(), ain.cpu().numpy()
            for i, b in enumerate(batch):      # the host recipe: pipeline._prepare's expressions (= bfs_ml_accelerated.py:1086-1097)
                sq = {c: (ry @ b[c] @ rx.T if resample else b[c]) for c in "uvp"}
                for k, c in enumerate("uvp"):
                    x32 = sq[c].astype(np.float32)
                    np.testing.assert_array_max_ulp(xg[3 * i + k], x32, maxulp=1)
                    mean_lr, std_lr = lr_stats[c]
                    if adaptive:
                        g32 = xg[3 * i + k]    # statistics of the device's own float32 field (it may differ from numpy's by 1 ulp)
                        input_mean, input_std = g32.reshape(1, -1).mean(axis=1)[0], g32.reshape(1, -1).std(axis=1)[0]
                        mean_lr = (1 - 0.3) * mean_lr + 0.3 * input_mean
                        std_lr = (1 - 0.3) * std_lr + 0.3 * max(input_std, 1e-8)
                    want = np.array([mean_lr, std_lr], np.float32)
                    np.testing.assert_array_equal(ag[3 * i + k].view(np.uint32), want.view(np.uint32))
    # the batched call == the per-field calls (BFS defaults: adaptive blend, aspect-ratio correction both ways)
    kw = dict(use_aspect_ratio_correction=True, lx=10.0, ly=3.0, blend_factor=0.3)
    got = pl.ml_super_resolution_batch(batch[:4], 10, 400, STATS_TXT, ENCODER_H5, decoder_h5, use_adaptive_normalization=True, **kw)
    for b, g in zip(batch[:4], got):
        one = pl.ml_super_resolution_bfs(b, 10, 400, STATS_TXT, ENCODER_H5, decoder_h5, **kw)
        for c in "uvp":
            assert g[c].shape == (400, 400) and g[c].dtype == np.float64
            assert np.linalg.norm(g[c] - one[c]) / np.linalg.norm(one[c]) <= 1e-5
    ldc = pl.ml_super_resolution_batch([coarse_cases["ldc_Re800_double"], coarse_cases["ldc_Re1000_single"]], 10, 400, STATS_TXT, ENCODER_H5, decoder_h5)
    one = pl.ml_super_resolution(coarse_cases["ldc_Re1000_single"], 10, 400, STATS_TXT, ENCODER_H5, decoder_h5)
    for c in "uvp":
        np.testing.assert_array_equal(ldc[1][c], one[c])


@pytest.mark.gpu
@pytest.mark.parametrize("side", [11, 12, 16, 23, 32])
def test_device_statistics_follow_numpy_beyond_128_elements(srcfd, side):
    """include/srcfd.h promises numpy's float32 mean / std bit for bit for sides up to 32: above 128 elements numpy's
    pairwise sum recurses (n/2 rounded down to a multiple of 8); 11x11 = 121 is the last one-block size, 12x12 the first split."""
    import ctypes as C
    import torch
    require_gpu(srcfd)
    L = importlib.import_module("sr-for-cfd_amd._lib")
    rng = np.random.default_rng(side)
    n = 6
    fields = (rng.standard_normal((n, side, side)) * rng.uniform(0.1, 30.0, (n, 1, 1)) + rng.standard_normal((n, 1, 1))).astype(np.float64)
    dev = torch.device("cuda", 0)
    f_dev = torch.from_numpy(fields).to(dev)
    tr_np = np.stack([rng.standard_normal(n) * 0.2, rng.uniform(0.1, 2.0, n)], 1)
    tr = torch.from_numpy(tr_np).to(dev)
    x = torch.empty((n, side, side), dtype=torch.float32, device=dev)
    ain = torch.empty((n, 2), dtype=torch.float32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    L.check(L.lib.srcfd_prepare_inputs_device(p(f_dev), n, side, side, None, None, side, p(tr), 1, 0.3, p(x), p(ain), None))
    torch.cuda.synchronize()
    xg, ag = x.cpu().numpy(), ain.cpu().numpy()
    for i in range(n):
        x32 = fields[i].astype(np.float32)
        np.testing.assert_array_equal(xg[i], x32)
        input_mean, input_std = np.mean(x32), np.std(x32)      # the reference's expressions, bfs_ml_accelerated.py:1091-1097
        # the training statistics are Python floats in the reference (parsed from the stats file): weak scalars next to np.float32
        mean_lr = (1 - 0.3) * float(tr_np[i, 0]) + 0.3 * input_mean
        std_lr = (1 - 0.3) * float(tr_np[i, 1]) + 0.3 * max(input_std, 1e-8)
        want = np.array([mean_lr, std_lr], np.float32)
        np.testing.assert_array_equal(ag[i].view(np.uint32), want.view(np.uint32), err_msg=f"side {side} sample {i}")


def test_result_pool_policy_size_classes_and_lru_eviction(monkeypatch):
    """The recycling pool's policy on a stand-in allocator (libc malloc; the real one is srcfd_host_alloc): sizes share a few
    classes, the cache is bounded, and a released buffer that does not fit evicts the least recently used ones instead of being
    freed itself (a caller with varying batch sizes kept 4 GiB pinned and lost the pool)."""
    import ctypes as C
    import gc
    import importlib
    eng = importlib.import_module("sr-for-cfd_amd.engine")
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.malloc.argtypes = [C.c_size_t]
    libc.free.argtypes = [C.c_void_p]
    live = {}

    class FakeLib:
        @staticmethod
        def srcfd_host_alloc(nbytes, out):
            p = libc.malloc(nbytes)
            out._obj.value = p
            live[p] = nbytes
            return 0

        @staticmethod
        def srcfd_host_free(p):
            live.pop(p.value)
            libc.free(p)

    class FakeL:
        lib = FakeLib

    monkeypatch.setattr(eng, "L", FakeL)
    monkeypatch.setenv("SRCFD_RESULT_POOL", "1")
    P = eng._PinnedPool
    assert P.size_class(8 << 20) == 8 << 20 and P.size_class((8 << 20) + 1) == 10 << 20 and P.size_class(5000) == 8192
    for nb in (4 << 20, 491_520_000, 123_456_789):
        assert nb <= P.size_class(nb) <= nb * 1.25 + 4096
    pool = P(min_bytes=1 << 20, cap_bytes=24 << 20)
    a = pool.empty((2 << 20,))              # 8 MiB class
    b = pool.empty(((2 << 20) - 100,))      # same class: 8 MiB - 400 B rounds up to 8 MiB
    pa, pb = a.ctypes.data, b.ctypes.data
    assert pool.stats["allocated"] == 2 and sorted(live.values()) == [8 << 20] * 2
    del a
    gc.collect()
    assert pool.cached == 8 << 20
    c = pool.empty((2 << 20,))              # reuse of a's buffer
    assert c.ctypes.data == pa and pool.stats["reused"] == 1 and pool.cached == 0
    d = pool.empty((4 << 20,))              # 16 MiB class
    del b, c
    gc.collect()
    assert pool.cached == 16 << 20 and len(pool.free) == 2
    del d
    gc.collect()                            # 16 + 16 > 24: the least recently released 8 MiB buffer (b's) goes, d's stays cached
    assert pool.cached == 24 << 20 and pool.stats["evicted"] == 1 and pb not in live and pa in live
    e = pool.empty((4 << 20,))
    assert pool.stats["reused"] == 2        # d's buffer came back although the cache was full when it was released
    big = pool.empty((8 << 20,))            # 32 MiB > cap: never cached
    del big, e
    gc.collect()
    assert pool.cached <= 24 << 20
    small = pool.empty((1000,))             # below min_bytes: plain numpy
    assert small.ctypes.data not in live
    pool.trim()
    assert pool.cached == 0 and not live


@pytest.mark.gpu
def test_large_results_come_from_the_page_locked_pool(srcfd, enc_weights, dec_weights):
    """`predict` returns a NEW array per call (the reference's contract, PyCFD_ML_accelerated.py:858); large ones live in recycled
    page-locked buffers: the values equal the `out=` path's, a buffer returns to the pool only when the array AND its views are gone
    (a field sliced out of a result keeps it alive), the next call of the same size reuses it, and small results are plain numpy."""
    import gc
    import importlib
    require_gpu(srcfd)
    eng = importlib.import_module("sr-for-cfd_amd.engine")
    pool = eng._result_pool
    rng = np.random.default_rng(4)
    n = 40                                                  # 40 x 640 kB = 25.6 MB: above the pool's 4 MB threshold, two chunks? no: one
    x = rng.standard_normal((n, 10, 10, 1)).astype(np.float32)
    m = srcfd.SRModel.from_weights(enc_weights, dec_weights, device=0)
    m.precision = "bf16"
    ref = np.empty((n, 400, 400, 1), np.float32)
    m.predict(x, out=ref)
    before = dict(pool.stats)
    y = m.predict(x)
    assert pool.stats["allocated"] + pool.stats["reused"] == before["allocated"] + before["reused"] + 1
    np.testing.assert_array_equal(y, ref)
    addr = y.ctypes.data
    field = y[3, ..., 0]                                    # what the solver harness keeps
    del y
    gc.collect()
    y2 = m.predict(x)                                       # the first buffer is still referenced by `field`: a different one
    assert y2.ctypes.data != addr
    np.testing.assert_array_equal(field, ref[3, ..., 0])
    del field, y2
    gc.collect()
    reused = pool.stats["reused"]
    y3 = m.predict(x)
    assert pool.stats["reused"] == reused + 1               # recycled, no new page-locked allocation
    np.testing.assert_array_equal(y3, ref)
    small = m.predict(x[:2])                                # 1.3 MB: ordinary numpy memory
    np.testing.assert_array_equal(small, ref[:2])
    # a batch of several chunks through the overlapped copy path (> 128 samples), against the single-buffer path into pageable memory
    xb = rng.standard_normal((300, 10, 10, 1)).astype(np.float32)
    aout = np.stack([rng.standard_normal(300) * 0.1, rng.uniform(0.05, 0.3, 300)], 1).astype(np.float32)
    refb = np.empty((300, 400, 400, 1), np.float32)
    m.predict(xb, out_affine=aout, nan_guard=True, out=refb)
    yb, bad = m.predict(xb, out_affine=aout, nan_guard=True, return_nonfinite=True)
    np.testing.assert_array_equal(yb, refb)
    assert bad == 0
    del y3, yb
    gc.collect()
    pool.trim()


@pytest.mark.gpu
def test_verbose_call_prints_the_reference_style_report_and_same_fields(srcfd, decoder_h5, coarse_cases, capsys):
    """verbose=True reports the blended statistics and the range of every component (the reference prints them on every
    call, bfs_ml_accelerated.py:1096-1145); the quiet default skips those range scans but returns the same arrays."""
    require_gpu(srcfd)
    pl = importlib.import_module("sr-for-cfd_amd.pipeline")
    case = coarse_cases["bfs_Re400"]
    kw = dict(use_aspect_ratio_correction=True, lx=10.0, ly=3.0, blend_factor=0.3)
    quiet = pl.ml_super_resolution_bfs(case, 10, 400, STATS_TXT, ENCODER_H5, decoder_h5, **kw)
    assert capsys.readouterr().out == ""
    loud = pl.ml_super_resolution_bfs(case, 10, 400, STATS_TXT, ENCODER_H5, decoder_h5, verbose=True, **kw)
    text = capsys.readouterr().out
    for c in ("u", "v", "p"):
        np.testing.assert_array_equal(loud[c], quiet[c])
        assert f"{c.upper()}: adaptive norm (blend=0.30)" in text
        assert f"range [{loud[c].min():.6f}, {loud[c].max():.6f}]" in text


@pytest.mark.gpu
def test_tiled_sr_config5(srcfd, oracle, enc_weights, dec_weights):
    """40x40x3 -> 1600x1600x3 through 4x4 tiles (BASELINE config 5), f16 operands: every one of the 48 tile samples of the
    stitched field equals the per-tile predict bit for bit (tiling = slicing + independent samples, nothing else), and
    tiles in three different corners / components match the float64 oracle."""
    require_gpu(srcfd)
    pl = importlib.import_module("sr-for-cfd_amd.pipeline")
    rng = np.random.default_rng(9)
    field = rng.standard_normal((40, 40, 3)).astype(np.float32)
    lr, hr = srcfd.load_stats(STATS_TXT, 10, 400)
    ain = np.array([lr[c] for c in "uvp"], np.float32)
    aout = np.array([hr[c] for c in "uvp"], np.float32)
    field = field * ain[:, 1] + ain[:, 0]
    m = srcfd.SRModel.from_weights(enc_weights, dec_weights, device=0)
    m.precision = "f16"
    y = pl.tiled_super_resolution(field, m, lr_dim=10, in_affine=ain, out_affine=aout)
    assert y.shape == (1600, 1600, 3) and np.isfinite(y).all()
    for ty in range(4):
        for tx in range(4):
            for c in range(3):
                t = np.ascontiguousarray(field[10 * ty:10 * ty + 10, 10 * tx:10 * tx + 10, c])[None, ..., None]
                one = m.predict(t, in_affine=ain[c:c + 1], out_affine=aout[c:c + 1])[0, ..., 0]
                np.testing.assert_array_equal(y[400 * ty:400 * ty + 400, 400 * tx:400 * tx + 400, c], one)
    for ty, tx, c in ((1, 2, 1), (0, 0, 0), (3, 3, 2), (2, 0, 1)):
        t = field[10 * ty:10 * ty + 10, 10 * tx:10 * tx + 10, c][None, ..., None]
        ts = ((t - ain[c, 0]) / ain[c, 1]).astype(np.float32)
        ref = oracle.superres_forward(ts, enc_weights, dec_weights, np.float64)[0, ..., 0]
        got = (y[400 * ty:400 * ty + 400, 400 * tx:400 * tx + 400, c] - aout[c, 0]) / aout[c, 1]
        assert oracle.rel_l2(got[None], ref[None]) <= 3e-3, (ty, tx, c)


@pytest.mark.gpu
def test_two_handles_on_two_threads(srcfd, enc_weights, dec_weights):
    """The ABI's threading contract (include/srcfd.h; SURVEY.md 8b "threading"): distinct handles may be driven from distinct
    threads at the same time (ctypes drops the GIL around every call), and srcfd_last_error() is per thread -- each thread
    reads the message of ITS failed call, whatever the other thread did in between."""
    import threading
    require_gpu(srcfd)
    L = importlib.import_module("sr-for-cfd_amd._lib")
    rng = np.random.default_rng(21)
    xs = [rng.standard_normal((5 + 3 * i, 10, 10, 1)).astype(np.float32) for i in range(2)]
    models = [srcfd.SRModel.from_weights(enc_weights, dec_weights, device=0) for _ in range(2)]
    models[0].precision, models[1].precision = "bf16", "fp32"
    want = [m.predict(x) for m, x in zip(models, xs)]              # single-threaded results
    got, errs, msgs = [None, None], [None, None], [None, None]
    gate = threading.Barrier(2)

    def work(i):
        try:
            gate.wait()
            for _ in range(12):                                     # overlapping forward passes on the two handles
                y = models[i].predict(xs[i])
            got[i] = y
            gate.wait()
            # a failing call per thread, interleaved: 0 fails, 1 fails, then each reads its own message
            if i == 0:
                rc = L.lib.srcfd_model_set_precision(models[0]._h, 99)
                gate.wait(); gate.wait()
            else:
                gate.wait()
                rc = L.lib.srcfd_predict(models[1]._h, None, -3, None, None, None, 0, None)
                gate.wait()
            assert rc != 0
            msgs[i] = L.last_error()
        except Exception as e:   # noqa: BLE001 - reported below
            errs[i] = e
            gate.abort()

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert errs == [None, None], errs
    np.testing.assert_array_equal(got[0], want[0])
    np.testing.assert_array_equal(got[1], want[1])
    assert "precision" in msgs[0] and "precision" not in msgs[1] and msgs[1], msgs


LDC_BC = {"u": {"left": ("dirichlet", 0.0), "right": ("dirichlet", 0.0), "top": ("dirichlet", 1.0), "bottom": ("dirichlet", 0.0)},
          "v": {s: ("dirichlet", 0.0) for s in ("left", "right", "top", "bottom")},
          "p": {s: ("neumann", 0.0) for s in ("left", "right", "top", "bottom")}}  # BoundaryConditions defaults, PyCFD...:47-67


def test_bc_arrays_and_inlet_profile_match_the_oracle(srcfd, oracle):
    pl = importlib.import_module("sr-for-cfd_amd.pipeline")
    t, v = pl.bc_arrays(LDC_BC)
    assert t.tolist() == [[0, 0, 0, 0], [0, 0, 0, 0], [1, 1, 1, 1]] and v[0].tolist() == [0.0, 0.0, 1.0, 0.0]

    class E:  # the solvers' BoundaryCondition objects
        def __init__(self, t, v):
            self.type, self.value = t, v

    class BC:
        u_boundaries = {k: E(*LDC_BC["u"][k]) for k in LDC_BC["u"]}
        v_boundaries = {k: E(*LDC_BC["v"][k]) for k in LDC_BC["v"]}
        p_boundaries = {k: E(*LDC_BC["p"][k]) for k in LDC_BC["p"]}
    t2, v2 = pl.bc_arrays(BC())
    assert (t2 == t).all() and (v2 == v).all()
    a, b = pl.bfs_inlet_profiles(400, 3.0 / 400, 1.0, 2.0, 1.0), oracle.bfs_inlet_profiles(400, 3.0 / 400, 1.0, 2.0, 1.0)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])


@pytest.mark.gpu
def test_solver_state_handoff_is_bit_identical_to_the_reference_recipe(srcfd, oracle, decoder_h5, coarse_cases):
    """SR call + transposed float64 injection + ghost cells in one device pass == ml_super_resolution
    followed by the reference's host steps (PyCFD...:936-943), bit for bit."""
    require_gpu(srcfd)
    pl = importlib.import_module("sr-for-cfd_amd.pipeline")
    case = coarse_cases["ldc_Re1000_double"]
    hr = pl.ml_super_resolution(case, 10, 400, STATS_TXT, ENCODER_H5, decoder_h5)
    t, v = pl.bc_arrays(LDC_BC)
    want = oracle.inject_and_apply_bc(hr, t, v)
    Var = np.full((3, 402, 402), 7.0)  # stale contents must be overwritten everywhere, corners included
    got = pl.ml_super_resolution_into_solver(case, 10, 400, STATS_TXT, ENCODER_H5, decoder_h5, LDC_BC, Var=Var)
    assert got is Var
    np.testing.assert_array_equal(got, want)
    assert got[0, 5, 401] == 2.0 - got[0, 5, 400] and got[2, 0, 9] == got[2, 1, 9] and got[1, 0, 0] == 0.0


@pytest.mark.gpu
def test_solver_state_handoff_bfs_with_resampling_and_inlet(srcfd, oracle, decoder_h5, coarse_cases):
    require_gpu(srcfd)
    pl = importlib.import_module("sr-for-cfd_amd.pipeline")
    case = coarse_cases["bfs_Re400"]
    kw = dict(use_aspect_ratio_correction=True, lx=10.0, ly=3.0, use_adaptive_normalization=True, blend_factor=0.3)
    hr = pl.ml_super_resolution(case, 10, 400, STATS_TXT, ENCODER_H5, decoder_h5, **kw)
    bc = {"u": {"left": ("dirichlet", 0.0), "right": ("neumann", 0.0), "top": ("dirichlet", 0.0), "bottom": ("dirichlet", 0.0)},
          "v": {"left": ("dirichlet", 0.0), "right": ("neumann", 0.0), "top": ("dirichlet", 0.0), "bottom": ("dirichlet", 0.0)},
          "p": {"left": ("neumann", 0.0), "right": ("dirichlet", 0.0), "top": ("neumann", 0.0), "bottom": ("neumann", 0.0)}}
    prof = pl.bfs_inlet_profiles(400, 3.0 / 400, 1.0, 2.0, 1.0)
    t, v = pl.bc_arrays(bc)
    want = oracle.inject_and_apply_bc(hr, t, v, prof)
    got = pl.ml_super_resolution_into_solver(case, 10, 400, STATS_TXT, ENCODER_H5, decoder_h5, bc, left_profiles=prof, **kw)
    np.testing.assert_array_equal(got, want)


@pytest.mark.gpu
def test_solver_state_handoff_random_boundary_conditions(srcfd, oracle, decoder_h5, coarse_cases):
    """Every mix of Dirichlet / Neumann sides, arbitrary values, inlet profiles on any subset of the components, with and
    without the resampler: the device pass equals the host recipe applied to the same super-resolved fields."""
    require_gpu(srcfd)
    pl = importlib.import_module("sr-for-cfd_amd.pipeline")
    rng = np.random.default_rng(17)
    sides = ("left", "right", "top", "bottom")
    for it in range(12):
        resample = it % 2 == 1
        case = coarse_cases["bfs_Re400" if resample else "ldc_Re800_single"]
        kw = dict(use_aspect_ratio_correction=True, lx=10.0, ly=3.0, use_adaptive_normalization=True, blend_factor=0.3) if resample else {}
        bc = {c: {s: (("dirichlet", "neumann")[int(rng.integers(0, 2))], float(rng.standard_normal())) for s in sides} for c in "uvp"}
        prof = {k: rng.standard_normal(400) for k in range(3) if rng.integers(0, 2)} or None
        hr = pl.ml_super_resolution(case, 10, 400, STATS_TXT, ENCODER_H5, decoder_h5, **kw)
        t, v = pl.bc_arrays(bc)
        want = oracle.inject_and_apply_bc(hr, t, v, prof)
        got = pl.ml_super_resolution_into_solver(case, 10, 400, STATS_TXT, ENCODER_H5, decoder_h5, bc, left_profiles=prof, **kw)
        np.testing.assert_array_equal(got, want, err_msg=f"case {it}: {bc} profiles {None if prof is None else sorted(prof)}")


def test_stacked_adaptive_statistics_equal_the_per_component_calls():
    """`_prepare` takes np.mean / np.std of the three float32 components as two row reductions over the stacked batch;
    the reference calls them per component (bfs_ml_accelerated.py:1091-1092).  Same values, bit for bit."""
    rng = np.random.default_rng(11)
    for shape in ((10, 10), (20, 20), (7, 13)):
        for scale in (1.0, 1e-3, 250.0):
            x = (rng.standard_normal((3,) + shape) * scale + rng.standard_normal((3, 1, 1))).astype(np.float32)
            rows = np.ascontiguousarray(x[..., None]).reshape(3, -1)
            m, s = rows.mean(axis=1), rows.std(axis=1)
            for i in range(3):
                assert m[i] == np.mean(x[i]) and m.dtype == np.float32
                assert s[i] == np.std(x[i]) and s.dtype == np.float32
