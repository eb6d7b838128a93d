"""Training path: gradients vs a float64 autograd oracle, Adam vs the
Keras formula, the epoch batching rule, and the data-parallel collective on 2 gloo ranks."""
import importlib
import os
import socket

import numpy as np
import pytest

from conftest import ROOT, require_gpu


def test_epoch_batches_partition_like_shuffle_batch():
    tr = importlib.import_module("sr-for-cfd_amd.train")
    n = 87  # the reference's training-set size (sr-ae-conv.ipynb:r70)
    b = list(tr.epoch_batches(n, 8, epoch=3, seed=0))
    assert [len(x) for x in b] == [8] * 10 + [7]
    assert sorted(np.concatenate(b).tolist()) == list(range(n))
    assert not np.array_equal(np.concatenate(b), np.concatenate(list(tr.epoch_batches(n, 8, epoch=4, seed=0))))
    # two ranks: each global batch of 16 is split in contiguous halves; union = same permutation
    r0 = list(tr.epoch_batches(n, 8, 3, 0, 0, 2))
    r1 = list(tr.epoch_batches(n, 8, 3, 0, 1, 2))
    assert len(r0) == len(r1) == 6
    merged = np.concatenate([np.concatenate([a, c]) for a, c in zip(r0, r1)])
    np.testing.assert_array_equal(merged, np.concatenate(b))


def test_created_pair_saves_as_encoder_and_decoder_files(srcfd, tmp_path):
    """A model built from layer specs (what training exports) writes the vanilla_encoder / vanilla_decoder pair."""
    synth = importlib.import_module("sr-for-cfd_amd.synth")
    enc, dec = synth.keras_default_init(3)
    m = srcfd.SRModel.from_weights(enc, dec, device=-1)
    e, d = str(tmp_path / "vanilla_encoder10_to_400_x.h5"), str(tmp_path / "vanilla_decoder400_from_10_x.h5")
    m.save_h5(e, d)
    me, md = srcfd.SRModel.load_h5(e, None, device=-1), srcfd.SRModel.load_h5(None, d, device=-1)
    assert me.input_shape == (10, 10, 1) and me.output_shape == (1, 1, 50) and md.output_shape == (400, 400, 1)
    both = srcfd.SRModel.load_h5(e, d, device=-1).weights()
    for k, v in {**enc, **dec}.items():
        np.testing.assert_array_equal(both[k], v)
    # glorot limits: |w| <= sqrt(6 / (fan_in + fan_out)), zero biases
    assert np.abs(enc["dense/kernel"]).max() <= np.sqrt(6.0 / (3200 + 128)) and not dec["dense_1/bias"].any()
    assert np.abs(dec["conv2d_transpose/kernel"]).max() <= np.sqrt(6.0 / (9 * 128 + 9 * 256))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _dp_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr = importlib.import_module("sr-for-cfd_amd.train")
    g = torch.arange(10, dtype=torch.float32) * (rank + 1)  # stands for this rank's flat gradient
    tr.allreduce_sum_(g)
    if rank == 0:
        q.put((g.tolist(), tr.world_size(), tr.rank()))
    dist.destroy_process_group()


def test_gradient_allreduce_two_ranks_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    g, w, r = q.get(timeout=120)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert g == [3.0 * i for i in range(10)] and w == 2 and r == 0


@pytest.mark.gpu
def test_gradients_match_autograd_oracle(srcfd, oracle, enc_weights, dec_weights):
    require_gpu(srcfd)
    import torch
    from oracle import sr_oracle_autograd as ag
    tr = importlib.import_module("sr-for-cfd_amd.train")
    rng = np.random.default_rng(31)
    n = 3
    x = rng.standard_normal((n, 10, 10, 1)).astype(np.float32)
    y = rng.standard_normal((n, 400, 400, 1)).astype(np.float32)
    model = srcfd.SRModel.from_weights(enc_weights, dec_weights, device=0)
    t = tr.Trainer(model, max_batch=4)
    assert t.n_params == 2_709_491
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    t.grads.zero_()
    t.sse.zero_()
    t.forward_backward(xd, yd)
    torch.cuda.synchronize()
    loss_ref, g_ref = ag.loss_and_grads(x, y, enc_weights, dec_weights)
    loss = float(t.sse.item()) / (n * 160000)
    assert abs(loss - loss_ref) <= 1e-5 * abs(loss_ref)
    g = t.grads.cpu().numpy().astype(np.float64)
    # per-tensor relative L2 (f32 arithmetic through 11 layers against float64 autograd)
    off = 0
    for name in ag.flat_order(enc_weights, dec_weights):
        size = {**enc_weights, **dec_weights}[name].size
        a, b = g[off:off + size], g_ref[off:off + size]
        rel = np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
        assert rel <= 2e-4, (name, rel)
        off += size
    assert off == t.n_params
    # determinism: a second pass gives bit-identical gradients (no float atomics)
    g1 = t.grads.clone()
    t.grads.zero_()
    t.forward_backward(xd, yd)
    torch.cuda.synchronize()
    assert torch.equal(g1, t.grads)
    # data parallelism by construction: per-rank gradients of the global mean add up to the full-batch gradient
    t.grads.zero_()
    t.forward_backward(xd[:2].contiguous(), yd[:2].contiguous(), global_batch=n)   # "rank 0"
    ga = t.grads.clone()
    t.grads.zero_()
    t.forward_backward(xd[2:].contiguous(), yd[2:].contiguous(), global_batch=n)   # "rank 1"
    torch.cuda.synchronize()
    summed = (ga + t.grads).cpu().numpy().astype(np.float64)
    assert np.linalg.norm(summed - g) <= 1e-5 * np.linalg.norm(g)


@pytest.mark.gpu
@pytest.mark.parametrize("switch,n", [("SRCFD_TRAIN_TAIL", 1), ("SRCFD_TRAIN_TAIL", 3), ("SRCFD_TRAIN_TAIL", 8),
                                      ("SRCFD_TRAIN_ENC", 1), ("SRCFD_TRAIN_ENC", 7), ("SRCFD_TRAIN_ENC", 19)])
def test_fused_stages_match_layer_by_layer_step(srcfd, enc_weights, dec_weights, monkeypatch, switch, n):
    """SRCFD_TRAIN_TAIL: the last four layers as two launches (tail32<TRAIN> forward, tail_bwd32 backward with recomputation;
    csrc/train_tail.hip) against the layer-by-layer step (= 0): same loss, same gradient per tensor up to f32 summation
    order.  n = 1 and 3 leave the last 16-pixel tile partial (2500 n is not a multiple of 16); targets with structure near
    the image border exercise the output conv's SAME padding in the gradient.
    SRCFD_TRAIN_ENC: the encoder's four layers forward as two launches (csrc/train_enc.hip: the dense layer's K dimension cut
    over 50 workgroups, every layer's pre-activation and activation stored) against the six generic launches; n = 1, 7 leave
    the 16-sample group partial, n = 19 spans two groups."""
    require_gpu(srcfd)
    import torch
    from oracle import sr_oracle_autograd as ag
    tr = importlib.import_module("sr-for-cfd_amd.train")
    rng = np.random.default_rng(40 + n)
    x = rng.standard_normal((n, 10, 10, 1)).astype(np.float32)
    y = rng.standard_normal((n, 400, 400, 1)).astype(np.float32)
    y[:, :2] += 3.0
    y[:, :, -2:] -= 3.0
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()

    def run(env):
        monkeypatch.setenv(switch, env)
        t = tr.Trainer(srcfd.SRModel.from_weights(enc_weights, dec_weights, device=0), max_batch=max(8, n))
        out = []
        for _ in range(3):   # the third call replays the captured graph
            t.grads.zero_()
            t.sse.zero_()
            t.forward_backward(xd, yd)
            torch.cuda.synchronize()
            out.append((float(t.sse.item()), t.grads.cpu().numpy().astype(np.float64)))
        for sse, g in out[1:]:
            assert sse == out[0][0] and np.array_equal(g, out[0][1])   # run to run identical, plain launches and graph replay
        return out[0]

    sse0, g0 = run("0")
    sse1, g1 = run("1")
    assert abs(sse1 - sse0) <= 2e-6 * abs(sse0)
    assert not np.array_equal(g0, g1)      # the switch did select other kernels
    off = 0
    both = {**enc_weights, **dec_weights}
    for name in ag.flat_order(enc_weights, dec_weights):
        size = both[name].size
        a, b = g1[off:off + size], g0[off:off + size]
        rel = np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
        assert rel <= 2e-5, (name, rel)
        off += size


@pytest.mark.gpu
def test_overwrite_mode_and_stream_split_do_not_change_a_bit(srcfd, enc_weights, dec_weights, monkeypatch):
    """SRCFD_TRAIN_OVERWRITE (srcfd_trainer_forward_backward_ex): gradients and the squared-error sum are stored, not added --
    over buffers full of garbage they must equal the accumulating call over zeroed buffers bit for bit (every parameter is
    written by exactly one thread of the step's one slab-sum launch, shared ConvT biases included); calling the accumulating
    form twice doubles.  Where the weight gradients are split between the two streams (SRCFD_TRAIN_AUX_FROM) and whether a
    second stream is used at all changes nothing either."""
    require_gpu(srcfd)
    import torch
    tr = importlib.import_module("sr-for-cfd_amd.train")
    rng = np.random.default_rng(77)
    n = 5
    xd = torch.from_numpy(rng.standard_normal((n, 10, 10, 1)).astype(np.float32)).cuda()
    yd = torch.from_numpy(rng.standard_normal((n, 400, 400, 1)).astype(np.float32)).cuda()

    def run(aux_from, tail="1"):
        if aux_from is None:
            monkeypatch.delenv("SRCFD_TRAIN_AUX_FROM", raising=False)
        else:
            monkeypatch.setenv("SRCFD_TRAIN_AUX_FROM", aux_from)
        monkeypatch.setenv("SRCFD_TRAIN_TAIL", tail)
        t = tr.Trainer(srcfd.SRModel.from_weights(enc_weights, dec_weights, device=0), max_batch=8)
        res = []
        for rep in range(3):   # plain launches, capture, replay
            t.grads.zero_(); t.sse.zero_()
            t.forward_backward(xd, yd)
            torch.cuda.synchronize()
            acc = (float(t.sse.item()), t.grads.cpu().numpy().copy())
            t.grads.fill_(float("nan")); t.sse.fill_(1e300)
            t.forward_backward(xd, yd, overwrite=True)
            torch.cuda.synchronize()
            ow = (float(t.sse.item()), t.grads.cpu().numpy().copy())
            assert ow[0] == acc[0] and np.array_equal(ow[1], acc[1]), rep
            res.append(acc)
        assert all(r[0] == res[0][0] and np.array_equal(r[1], res[0][1]) for r in res)
        t.forward_backward(xd, yd)          # accumulates on top of the stored values
        torch.cuda.synchronize()
        assert float(t.sse.item()) == 2 * res[0][0] and np.array_equal(t.grads.cpu().numpy(), 2 * res[0][1])
        for rep in range(3):                # SRCFD_TRAIN_SAME_PARAMS: the re-packing is skipped, nothing else changes (plain, capture, replay)
            t.forward_backward(xd, yd, overwrite=True, same_params=True)
            torch.cuda.synchronize()
            assert float(t.sse.item()) == res[0][0] and np.array_equal(t.grads.cpu().numpy(), res[0][1]), rep
        t.params.mul_(1.5)                  # ... and a caller that breaks the contract gets the OLD parameters' gradients
        t.forward_backward(xd, yd, overwrite=True, same_params=True)
        torch.cuda.synchronize()
        assert np.array_equal(t.grads.cpu().numpy(), res[0][1])
        t.forward_backward(xd, yd, overwrite=True)
        torch.cuda.synchronize()
        assert not np.array_equal(t.grads.cpu().numpy(), res[0][1])
        return res[0]

    base = run(None)
    assert np.isfinite(base[1]).all() and np.abs(base[1]).max() > 0
    for aux_from in ("0", "3", "6", "99"):
        r = run(aux_from)
        assert r[0] == base[0] and np.array_equal(r[1], base[1]), aux_from
    lbl = run(None, tail="0")               # layer by layer: the four ConvT#0 phases AND the tail's layers go through the table
    lbl2 = run("8", tail="0")
    assert lbl2[0] == lbl[0] and np.array_equal(lbl2[1], lbl[1])


@pytest.mark.gpu
@pytest.mark.parametrize("hw,n", [((6, 7), 3), ((3, 18), 2), ((9, 4), 5)])
def test_fused_tail_on_other_image_sizes(srcfd, monkeypatch, hw, n):
    """tail32<TRAIN> / tail_bwd32 take the spatial size of the tail's input as a parameter: a small graph with the same last four
    layers (Conv 3x3 3->64 in front; ConvT 64->32->16->8, Conv 8->1) on rows of 7, 18 and 4 pixels -- 16-pixel tiles that span
    several rows and samples, a last tile that is partial, strips shorter than a wave -- against the layer-by-layer step."""
    require_gpu(srcfd)
    import torch
    tr = importlib.import_module("sr-for-cfd_amd.train")
    h, w = hw
    rng = np.random.default_rng(100 + h * w)

    def glorot(shape, fan_in, fan_out):
        lim = np.sqrt(6.0 / (fan_in + fan_out))
        return rng.uniform(-lim, lim, shape).astype(np.float32)

    specs = [dict(kind="conv2d", name="front", k=3, stride=1, same=True, act="swish", w=glorot((3, 3, 3, 64), 27, 576), b=(0.1 * rng.standard_normal(64)).astype(np.float32))]
    cin = 64
    for i, cout in enumerate((32, 16, 8)):
        specs.append(dict(kind="conv2d_transpose", name=f"up{i}", k=2, stride=2, same=False, act="swish", w=glorot((2, 2, cout, cin), 4 * cin, 4 * cout),
                          b=(0.1 * rng.standard_normal(cout)).astype(np.float32)))
        cin = cout
    specs.append(dict(kind="conv2d", name="out", k=3, stride=1, same=True, act="linear", w=glorot((3, 3, 8, 1), 72, 9), b=np.array([0.05], np.float32)))
    x = torch.from_numpy(rng.standard_normal((n, h, w, 3)).astype(np.float32)).cuda()
    y = torch.from_numpy(rng.standard_normal((n, 8 * h, 8 * w, 1)).astype(np.float32)).cuda()

    def run(env):
        monkeypatch.setenv("SRCFD_TRAIN_TAIL", env)
        m = srcfd.SRModel.from_layers(specs, (h, w, 3), device=0)
        assert m.output_shape == (8 * h, 8 * w, 1)
        t = tr.Trainer(m, max_batch=8)
        t.grads.zero_()
        t.sse.zero_()
        t.forward_backward(x, y)
        torch.cuda.synchronize()
        return float(t.sse.item()), t.grads.cpu().numpy().astype(np.float64), t

    sse0, g0, _ = run("0")
    sse1, g1, t = run("1")
    assert abs(sse1 - sse0) <= 2e-6 * abs(sse0)
    sizes = [int(np.prod(s["w"].shape)) for s in specs for _ in (0,)]
    off = 0
    for s in specs:
        for size in (s["w"].size, s["b"].size):
            a, b = g1[off:off + size], g0[off:off + size]
            assert np.linalg.norm(a - b) <= 2e-5 * max(np.linalg.norm(b), 1e-30), (s["name"], size)
            off += size
    assert off == t.n_params and sizes


@pytest.mark.gpu
def test_adam_and_ragged_batch_and_loss_decreases(srcfd, oracle, enc_weights, dec_weights):
    require_gpu(srcfd)
    import torch
    from oracle import sr_oracle_autograd as ag
    tr = importlib.import_module("sr-for-cfd_amd.train")
    rng = np.random.default_rng(32)
    model = srcfd.SRModel.from_weights(enc_weights, dec_weights, device=0)
    t = tr.Trainer(model, max_batch=8)
    p0 = t.params.cpu().numpy().astype(np.float64)
    x = rng.standard_normal((8, 10, 10, 1)).astype(np.float32)
    y = (0.3 * rng.standard_normal((8, 400, 400, 1))).astype(np.float32)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    l0 = t.step(xd, yd)
    g = t.grads.cpu().numpy().astype(np.float64)
    p_ref, _, _ = ag.adam_reference(p0, g, np.zeros_like(p0), np.zeros_like(p0), 1)
    np.testing.assert_allclose(t.params.cpu().numpy(), p_ref, rtol=2e-6, atol=2e-9)  # first step moves every weight by ~lr
    losses = [l0] + [t.step(xd, yd) for _ in range(5)]
    assert losses[-1] < losses[0]
    l7 = t.step(xd[:7].contiguous(), yd[:7].contiguous())  # Keras' ragged last batch (87 = 10*8 + 7)
    assert np.isfinite(l7)
    with pytest.raises(ValueError):
        t.step(torch.zeros((9, 10, 10, 1), device="cuda"), torch.zeros((9, 400, 400, 1), device="cuda"))
    # trained weights flow back into an inference handle
    m2 = t.export_model()
    yp = m2.predict(x[:2])
    ref = oracle.superres_forward(x[:2], {k: v for k, v in t.weights().items() if k.split("/")[0] in oracle.ENCODER_LAYERS},
                                  {k: v for k, v in t.weights().items() if k.split("/")[0] in oracle.DECODER_LAYERS}, np.float64)
    assert oracle.rel_l2(yp, ref) <= 1e-5


@pytest.mark.gpu
def test_fit_from_keras_default_init_on_the_dummy_recipe(srcfd):
    """sr-ae-conv.ipynb end to end at small scale: dummy data recipe (c72-91), component statistics,
    glorot init, shuffle(len).batch(8) epochs; the reconstruction loss must fall."""
    require_gpu(srcfd)
    tr = importlib.import_module("sr-for-cfd_amd.train")
    ds = importlib.import_module("sr-for-cfd_amd.datasets")
    synth = importlib.import_module("sr-for-cfd_amd.synth")
    x_lr, x_hr, res, comps, bcs = ds.dummy_pairs(10, 400, n_per_component=6, seed=0)
    xl, xh, stats_lr, stats_hr = ds.component_standardize(x_lr, x_hr, comps)
    enc, dec = synth.keras_default_init(0)
    t = tr.Trainer(srcfd.SRModel.from_weights(enc, dec, device=0), max_batch=8)
    hist = tr.fit(t, xl, xh, epochs=4, batch_size=8, seed=0)
    assert len(hist) == 4 and all(np.isfinite(hist)) and hist[-1] < hist[0]
    assert 0.5 < hist[0] < 1.5  # unit-variance targets, near-zero initial prediction


@pytest.mark.gpu
def test_step_variants_give_identical_parameters(srcfd, monkeypatch):
    """The step's structure (weight gradients on a second stream, swish / swish' folded into GEMM epilogues, replay as a
    hipGraph from the second identical call on) must not change a bit of the result: five Adam steps on moving x / y
    tensors with a ragged step in the middle, every switch off vs the default."""
    require_gpu(srcfd)
    import torch
    tr = importlib.import_module("sr-for-cfd_amd.train")
    synth = importlib.import_module("sr-for-cfd_amd.synth")
    enc, dec = synth.keras_default_init(3)
    rng = np.random.default_rng(8)
    batches = []
    for n in (4, 4, 3, 4, 4, 3):  # the 3-sample key comes back: it is captured on its second sighting too
        batches.append((rng.standard_normal((n, 10, 10, 1)).astype(np.float32), rng.standard_normal((n, 400, 400, 1)).astype(np.float32)))

    def run(env):
        for k in ("SRCFD_TRAIN_OVERLAP", "SRCFD_TRAIN_FUSE", "SRCFD_TRAIN_GRAPH"):
            monkeypatch.setenv(k, env)
        t = tr.Trainer(srcfd.SRModel.from_weights(enc, dec, device=0), max_batch=4)
        losses = []
        keep = []  # hold earlier batches so that later ones land at new addresses
        for x, y in batches:
            xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
            keep.append((xd, yd))
            losses.append(float(t.step(xd, yd)))
        return losses, {k: v.copy() for k, v in t.weights().items()}

    base_l, base_w = run("0")
    l, w = run("1")
    assert l == base_l
    for k in base_w:
        assert np.array_equal(w[k], base_w[k]), k


@pytest.mark.gpu
def test_training_driver_end_to_end(srcfd, tmp_path):
    """The notebook's __main__ (c374-604) in miniature: files -> split -> fit -> evaluate -> the three artefacts,
    which then load through the same path the solvers use."""
    require_gpu(srcfd)
    import json
    import subprocess
    import sys
    ds = importlib.import_module("sr-for-cfd_amd.datasets")
    h5 = importlib.import_module("sr-for-cfd_amd.h5")
    rng = np.random.default_rng(5)
    w = h5.H5Writer()
    for Re in (100, 200, 800):
        base = {c: rng.standard_normal((400, 400)) for c in "uvp"}
        ds.append_solution(w, Re, 400, base, "single_lid(u_top=1)")
        ds.append_solution(w, Re, 10, {c: ds.avg_pool(base[c][None, ..., None].astype(np.float32), 40)[0, ..., 0] for c in "uvp"}, "single_lid(u_top=1)")
    data = str(tmp_path / "simulation_result.h5")
    w.save(data)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "sr-for-cfd_amd", "train_main.py"), "--data", data, "--epochs", "3", "--suffix", "t",
                          "--out-dir", str(tmp_path), "--log-every", "0"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rep = json.loads(out.stdout.strip().splitlines()[-1])
    assert rep["train_samples"] == 6 and len(rep["Re800"]["mae"]) == 3 and np.isfinite(rep["average_nmae_percent"])
    enc, dec = str(tmp_path / "vanilla_encoder10_to_400_t.h5"), str(tmp_path / "vanilla_decoder400_from_10_t.h5")
    lr, hr = srcfd.load_stats(str(tmp_path / "standardization_stats_10to400_t.txt"), 10, 400)
    m = srcfd.SRModel.load_h5(enc, dec, device=0)
    assert m.output_shape == (400, 400, 1) and all(np.isfinite(v) for c in "uvp" for v in lr[c] + hr[c])


def _dp_gpu_worker(rank, world, port, q, ROOT_):
    import sys
    sys.path.insert(0, ROOT_)
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)   # both ranks share cuda:0 here; RCCL needs one GPU per rank
    srcfd = importlib.import_module("sr-for-cfd_amd")
    tr = importlib.import_module("sr-for-cfd_amd.train")
    synth = importlib.import_module("sr-for-cfd_amd.synth")
    enc, dec = synth.keras_default_init(0)
    t = tr.Trainer(srcfd.SRModel.from_weights(enc, dec, device=0), max_batch=4)
    rng = np.random.default_rng(77)
    x = rng.standard_normal((6, 10, 10, 1)).astype(np.float32)
    y = rng.standard_normal((6, 400, 400, 1)).astype(np.float32)
    lo, hi = (0, 3) if rank == 0 else (3, 6)
    losses = [t.step(torch.from_numpy(x[lo:hi]).cuda(), torch.from_numpy(y[lo:hi]).cuda()) for _ in range(2)]
    q.put((rank, losses, t.params.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_data_parallel_step_equals_single_process_step(srcfd):
    """Two ranks x micro-batch 3 (gradient all-reduce) == one process x batch 6: same losses, same weights after Adam."""
    require_gpu(srcfd)
    import torch
    import torch.multiprocessing as mp
    tr = importlib.import_module("sr-for-cfd_amd.train")
    synth = importlib.import_module("sr-for-cfd_amd.synth")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_dp_gpu_worker, args=(r, 2, port, q, ROOT)) for r in range(2)]
    for p in ps:
        p.start()
    got = dict()
    for _ in range(2):
        r, losses, params = q.get(timeout=300)
        got[r] = (losses, params)
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    np.testing.assert_array_equal(got[0][1], got[1][1])           # replicas stay identical
    assert got[0][0] == got[1][0]
    enc, dec = synth.keras_default_init(0)
    t = tr.Trainer(srcfd.SRModel.from_weights(enc, dec, device=0), max_batch=8)
    rng = np.random.default_rng(77)
    x = torch.from_numpy(rng.standard_normal((6, 10, 10, 1)).astype(np.float32)).cuda()
    y = torch.from_numpy(rng.standard_normal((6, 400, 400, 1)).astype(np.float32)).cuda()
    ref_losses = [t.step(x, y) for _ in range(2)]
    np.testing.assert_allclose(got[0][0], ref_losses, rtol=1e-6)
    ref = t.params.cpu().numpy()
    # summation order differs (two partial gradients added vs one pass): agreement to f32 rounding of the Adam update
    assert np.linalg.norm(got[0][1] - ref) <= 1e-5 * np.linalg.norm(ref)
