"""CPU oracle for the 10x10 -> 400x400 super-resolution hot path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the product path
(``sr-for-cfd_amd``) never does and fails loudly without its HIP library.

PARITY UNPINNED: the arithmetic of the reference lives in Keras 3.8.0 /
TensorFlow (``.h5`` root attr ``keras_version``; ``requirements.txt:6-7``
unpinned), neither of which exists in this image, and the reference holds no
test, golden vector or fixture for this path.  This
file is a restatement of the published Keras layer semantics, cross-checked
against an independent implementation (``torch.nn.functional`` on CPU, see
``sr_oracle_torch.py``) and analytic known-answer cases in ``tests/``.

What it restates (paths relative to the reference checkout):
  * encoder_10      sr-ae-conv.ipynb:c162-169 (+ ``model_config`` in the .h5)
  * decoder_400     sr-ae-conv.ipynb:c277-287
  * composition     sr-ae-conv.ipynb:c289-304, PyCFD_ML_accelerated.py:676-689
  * standardise     PyCFD_ML_accelerated.py:665-673, sr-ae-conv.ipynb:c111-113
  * stats parsing   PyCFD_ML_accelerated.py:787-809
  * adaptive blend  bfs_ml_accelerated.py:1091-1100
  * NaN/Inf guard   PyCFD_ML_accelerated.py:869-876
  * ml_super_resolution glue  PyCFD_ML_accelerated.py:764-879,
                              bfs_ml_accelerated.py:979-1137

Everything is plain numpy; ``dtype`` selects float64 (the reference answer the
tolerances are stated against) or float32 (same-precision restatement).
"""
from __future__ import annotations

import math
from typing import Dict, Sequence, Tuple

import numpy as np

COMPONENTS = ("u", "v", "p")

# Layer list of the two sub-models, in Keras creation order.
ENCODER_LAYERS = ("conv2d", "conv2d_1", "dense", "latent_vector")
DECODER_LAYERS = (
    "dense_1",
    "conv2d_transpose",
    "conv2d_transpose_1",
    "conv2d_transpose_2",
    "conv2d_transpose_3",
    "conv2d_transpose_4",
    "output_image_400",
)

# kernel shapes, Keras layouts: Conv2D (kh,kw,Cin,Cout); Conv2DTranspose
# (kh,kw,Cout,Cin); Dense (in,out).
ENCODER_SHAPES = {
    "conv2d": (3, 3, 1, 64),
    "conv2d_1": (3, 3, 64, 128),
    "dense": (3200, 128),
    "latent_vector": (128, 50),
}
DECODER_SHAPES = {
    "dense_1": (50, 36864),
    "conv2d_transpose": (3, 3, 128, 256),
    "conv2d_transpose_1": (2, 2, 64, 128),
    "conv2d_transpose_2": (2, 2, 32, 64),
    "conv2d_transpose_3": (2, 2, 16, 32),
    "conv2d_transpose_4": (2, 2, 8, 16),
    "output_image_400": (3, 3, 8, 1),
}


# ----------------------------------------------------------------------------
# element-wise
# ----------------------------------------------------------------------------
def silu(x: np.ndarray) -> np.ndarray:
    """swish / silu: x * sigmoid(x) (activation "swish" c164, serialised "silu")."""
    # stable in both tails: for x<0 use exp(x)/(1+exp(x))
    out = np.empty_like(x)
    pos = x >= 0
    ex = np.exp(-x[pos])
    out[pos] = x[pos] / (1.0 + ex)
    en = np.exp(x[~pos])
    out[~pos] = x[~pos] * en / (1.0 + en)
    return out


def _act(x: np.ndarray, activation: str) -> np.ndarray:
    if activation in ("swish", "silu"):
        return silu(x)
    if activation in ("linear", None):
        return x
    if activation == "relu":
        return np.maximum(x, 0)
    if activation == "sigmoid":
        return 1.0 / (1.0 + np.exp(-x))
    if activation == "tanh":
        return np.tanh(x)
    raise ValueError(f"unsupported activation {activation!r}")


# ----------------------------------------------------------------------------
# layers (NHWC, Keras semantics)
# ----------------------------------------------------------------------------
def same_padding(in_size: int, k: int, s: int) -> Tuple[int, int, int]:
    """TF 'SAME': out=ceil(in/s); pad_total=max((out-1)s+k-in,0); before=total//2."""
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    before = total // 2
    return out, before, total - before


def conv2d(x, w, b, stride=1, padding="same", activation="linear"):
    """Keras Conv2D, channels_last, dilation 1, groups 1.  w: (kh,kw,Cin,Cout)."""
    n, h, wd, cin = x.shape
    kh, kw, cin2, cout = w.shape
    assert cin == cin2
    if padding == "same":
        oh, pt, pb = same_padding(h, kh, stride)
        ow, pl, pr = same_padding(wd, kw, stride)
    else:
        oh, ow = (h - kh) // stride + 1, (wd - kw) // stride + 1
        pt = pb = pl = pr = 0
    xp = np.zeros((n, h + pt + pb, wd + pl + pr, cin), dtype=x.dtype)
    xp[:, pt:pt + h, pl:pl + wd, :] = x
    out = np.zeros((n, oh, ow, cout), dtype=x.dtype)
    for ky in range(kh):
        for kx in range(kw):
            patch = xp[:, ky:ky + (oh - 1) * stride + 1:stride,
                       kx:kx + (ow - 1) * stride + 1:stride, :]
            out += patch @ w[ky, kx].astype(x.dtype)
    out += b.astype(x.dtype)
    return _act(out, activation)


def conv2d_transpose(x, w, b, stride=2, padding="valid", activation="linear"):
    """Keras Conv2DTranspose, VALID.  w: (kh,kw,Cout,Cin); scatter form, no flip:
    out[s*i+a, s*j+b, co] += x[i,j,ci] * w[a,b,co,ci]   (SURVEY 8a row a13)."""
    if padding != "valid":
        raise ValueError("oracle restates VALID transposed conv only")
    n, h, wd, cin = x.shape
    kh, kw, cout, cin2 = w.shape
    assert cin == cin2
    oh, ow = (h - 1) * stride + kh, (wd - 1) * stride + kw
    out = np.zeros((n, oh, ow, cout), dtype=x.dtype)
    for a in range(kh):
        for bb in range(kw):
            contrib = x @ w[a, bb].astype(x.dtype).T  # (n,h,w,cout)
            out[:, a:a + (h - 1) * stride + 1:stride,
                bb:bb + (wd - 1) * stride + 1:stride, :] += contrib
    out += b.astype(x.dtype)
    return _act(out, activation)


def dense(x, w, b, activation="linear"):
    return _act(x @ w.astype(x.dtype) + b.astype(x.dtype), activation)


# ----------------------------------------------------------------------------
# the two sub-models and their composition
# ----------------------------------------------------------------------------
def encoder_forward(x: np.ndarray, wts: Dict[str, np.ndarray], dtype=np.float64,
                    return_all: bool = False):
    """encoder_10: (N,10,10,1) -> (N,50).  sr-ae-conv.ipynb:c162-169."""
    x = np.asarray(x, dtype=dtype)
    a1 = conv2d(x, wts["conv2d/kernel"], wts["conv2d/bias"], 2, "same", "swish")
    a2 = conv2d(a1, wts["conv2d_1/kernel"], wts["conv2d_1/bias"], 1, "same", "swish")
    f = a2.reshape(a2.shape[0], -1)  # Flatten, NHWC order (h*5+w)*128+c
    a3 = dense(f, wts["dense/kernel"], wts["dense/bias"], "swish")
    z = dense(a3, wts["latent_vector/kernel"], wts["latent_vector/bias"], "linear")
    if return_all:
        return z, [a1, a2, a3, z]
    return z


def decoder_forward(z: np.ndarray, wts: Dict[str, np.ndarray], dtype=np.float64,
                    return_all: bool = False):
    """decoder_400: (N,50) -> (N,400,400,1).  sr-ae-conv.ipynb:c277-287."""
    z = np.asarray(z, dtype=dtype)
    acts = []
    h = dense(z, wts["dense_1/kernel"], wts["dense_1/bias"], "swish")
    acts.append(h)
    h = h.reshape(-1, 12, 12, 256)
    for i, name in enumerate(DECODER_LAYERS[1:6]):
        h = conv2d_transpose(h, wts[f"{name}/kernel"], wts[f"{name}/bias"], 2,
                             "valid", "swish")
        acts.append(h)
    y = conv2d(h, wts["output_image_400/kernel"], wts["output_image_400/bias"], 1,
               "same", "linear")
    acts.append(y)
    if return_all:
        return y, acts
    return y


def superres_forward(x, enc_w, dec_w, dtype=np.float64):
    """SuperResolutionAE.call: decoder_hr(encoder_lr(x)).  PyCFD...:686-689."""
    return decoder_forward(encoder_forward(x, enc_w, dtype), dec_w, dtype)


# ----------------------------------------------------------------------------
# pre / post processing
# ----------------------------------------------------------------------------
def standardize_with_stats(arr, mean, std):
    """PyCFD_ML_accelerated.py:665-668 (std==0 -> 1e-8)."""
    std = 1e-8 if std == 0 else std
    return (arr - mean) / std


def inverse_standardize(arr, mean, std):
    """PyCFD_ML_accelerated.py:671-673."""
    return arr * std + mean


def parse_stats(path: str) -> Dict[str, float]:
    """Stats txt: 'key value' lines, '#' comments.  PyCFD...:787-797.
    Lines that do not split into exactly two tokens are silently skipped."""
    stats: Dict[str, float] = {}
    with open(path, "r") as f:
        for line in f:
            if line.strip().startswith("#") or not line.strip():
                continue
            parts = line.strip().split()
            if len(parts) == 2:
                stats[parts[0]] = float(parts[1])
    return stats


def component_stats(stats: Dict[str, float], lr_dim: int, hr_dim: int):
    """PyCFD...:800-809; raises KeyError on a missing key like the reference."""
    lr = {c: (stats[f"mean{lr_dim}_{c}"], stats[f"std{lr_dim}_{c}"]) for c in COMPONENTS}
    hr = {c: (stats[f"mean{hr_dim}_{c}"], stats[f"std{hr_dim}_{c}"]) for c in COMPONENTS}
    return lr, hr


def adaptive_blend(x_lr_raw: np.ndarray, mean_tr: float, std_tr: float,
                   blend_factor: float = 0.3) -> Tuple[float, float]:
    """bfs_ml_accelerated.py:1091-1097.  np.mean/np.std of the float32 field
    (population std), blended with the training stats in Python floats."""
    input_mean = np.mean(x_lr_raw)
    input_std = np.std(x_lr_raw)
    mean = (1 - blend_factor) * mean_tr + blend_factor * input_mean
    std = (1 - blend_factor) * std_tr + blend_factor * max(input_std, 1e-8)
    return float(mean), float(std)


def nan_guard(arr: np.ndarray) -> Tuple[np.ndarray, int, int]:
    """PyCFD...:869-876: any NaN/Inf -> zero-fill, report counts."""
    nan_count = int(np.isnan(arr).sum())
    inf_count = int(np.isinf(arr).sum())
    if nan_count or inf_count:
        arr = np.nan_to_num(arr, nan=0.0, posinf=0.0, neginf=0.0)
    return arr, nan_count, inf_count


def ml_super_resolution(coarse_fields: Dict[str, np.ndarray], lr_dim: int, hr_dim: int,
                        stats: Dict[str, float], enc_w, dec_w,
                        use_adaptive_normalization: bool = False,
                        blend_factor: float = 0.3, dtype=np.float32,
                        net_dtype=None) -> Dict[str, np.ndarray]:
    """The per-component loop of PyCFD...:841-876 / bfs...:1080-1127 (without the
    spline resampling, which is host-side scipy in both reference and build).

    ``dtype`` is the dtype of the pre/post arithmetic (float32 in the reference);
    ``net_dtype`` the dtype the network itself is evaluated in (default = dtype).
    """
    net_dtype = net_dtype or dtype
    stats_lr, stats_hr = component_stats(stats, lr_dim, hr_dim)
    out = {}
    for c in COMPONENTS:
        x = np.asarray(coarse_fields[c]).astype(np.float32)
        mean_lr, std_lr = stats_lr[c]
        mean_hr, std_hr = stats_hr[c]
        if use_adaptive_normalization:
            mean_lr, std_lr = adaptive_blend(x, mean_lr, std_lr, blend_factor)
        xn = standardize_with_stats(x.astype(dtype), mean_lr, std_lr).astype(dtype)
        y = superres_forward(xn[None, ..., None], enc_w, dec_w, net_dtype)[0, ..., 0]
        y = inverse_standardize(y.astype(dtype), mean_hr, std_hr).astype(dtype)
        y, _, _ = nan_guard(y)
        out[c] = y
    return out


# ----------------------------------------------------------------------------
# synthetic weights (the decoder .h5 files are absent from the reference
# checkout: .MISSING_LARGE_BLOBS:29-34)
# ----------------------------------------------------------------------------
def _glorot_uniform(rng, shape, fan_in, fan_out):
    limit = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-limit, limit, size=shape).astype(np.float32)


def _fans(shape, transpose=False):
    if len(shape) == 2:
        return shape[0], shape[1]
    rf = shape[0] * shape[1]
    if transpose:  # (kh,kw,Cout,Cin)
        return rf * shape[3], rf * shape[2]
    return rf * shape[2], rf * shape[3]


def synthetic_weights(shapes: Dict[str, Sequence[int]], seed: int, bias_scale: float = 0.1,
                      init: str = "preserve") -> Dict[str, np.ndarray]:
    """Seeded synthetic weights with non-zero biases (so bias handling is
    exercised).  init="glorot": Keras' default initialiser (model_config attr).
    init="preserve" (default): uniform with variance 2.4/fan_eff, fan_eff = the
    number of inputs that actually reach one output ((k/s)^2*Cin for a stride-s
    transposed conv) -- keeps activations O(1) through eleven swish layers the
    way trained weights do, so every layer's nonlinearity and rounding matters
    to the output; Glorot weights shrink the signal ~2x per layer."""
    rng = np.random.default_rng(seed)
    w = {}
    for name, shape in shapes.items():
        tr = name.startswith("conv2d_transpose")
        fi, fo = _fans(shape, transpose=tr)
        if init == "glorot":
            k = _glorot_uniform(rng, shape, fi, fo)
        else:
            fan_eff = fi / 4.0 if (tr and shape[0] == 2) else (fi / 2.25 if tr else fi)
            limit = math.sqrt(3.0 * 2.4 / fan_eff)
            k = rng.uniform(-limit, limit, size=shape).astype(np.float32)
        w[f"{name}/kernel"] = k
        nb = shape[2] if tr else shape[-1]
        w[f"{name}/bias"] = (bias_scale * rng.standard_normal(nb)).astype(np.float32)
    return w


def synthetic_decoder(seed: int = 1, init: str = "preserve") -> Dict[str, np.ndarray]:
    return synthetic_weights(DECODER_SHAPES, seed, init=init)


def synthetic_encoder(seed: int = 2, init: str = "preserve") -> Dict[str, np.ndarray]:
    return synthetic_weights(ENCODER_SHAPES, seed, init=init)


def rel_l2(y: np.ndarray, ref: np.ndarray) -> float:
    """Tolerance metric of SURVEY 8c: ||y-ref||_2 / ||ref||_2 per sample, max."""
    y = np.asarray(y, dtype=np.float64).reshape(y.shape[0], -1)
    ref = np.asarray(ref, dtype=np.float64).reshape(ref.shape[0], -1)
    num = np.linalg.norm(y - ref, axis=1)
    den = np.linalg.norm(ref, axis=1)
    return float(np.max(num / np.maximum(den, 1e-300)))


MACS_PER_SAMPLE = 140_024_128  # SURVEY 8a totals
FLOPS_PER_SAMPLE = 2 * MACS_PER_SAMPLE


# ---------------------------------------------------------------------------
# hand-off into the solver state (the step right after the SR call)
# ---------------------------------------------------------------------------
def inject_and_apply_bc(hr_fields, bc_types, bc_values, left_profiles=None):
    """PyCFD_ML_accelerated.py:936-943: Var[k,1:-1,1:-1] = field.T on a zeroed (3,nx+2,ny+2) float64 state, then
    `apply_bc_configured` (PyCFD...:118-146: left/right for j in 1..ny, top/bottom for i in 1..nx; Dirichlet
    2*value - inner, Neumann = inner), then the BFS left-boundary override (bfs_ml_accelerated.py:524-562) when
    `left_profiles[k]` (row-wise Dirichlet values) is given.  Plain loops, as the reference's njit code."""
    ny, nx = np.asarray(hr_fields["u"]).shape
    Var = np.zeros((3, nx + 2, ny + 2), np.float64)
    for k, c in enumerate(("u", "v", "p")):
        Var[k, 1:-1, 1:-1] = np.asarray(hr_fields[c]).T
        t, v = bc_types[k], bc_values[k]
        for j in range(1, ny + 1):
            Var[k, 0, j] = 2 * v[0] - Var[k, 1, j] if t[0] == 0 else Var[k, 1, j]
            Var[k, nx + 1, j] = 2 * v[1] - Var[k, nx, j] if t[1] == 0 else Var[k, nx, j]
        for i in range(1, nx + 1):
            Var[k, i, ny + 1] = 2 * v[2] - Var[k, i, ny] if t[2] == 0 else Var[k, i, ny]
            Var[k, i, 0] = 2 * v[3] - Var[k, i, 1] if t[3] == 0 else Var[k, i, 1]
        if left_profiles is not None and left_profiles.get(k) is not None:
            for j in range(1, ny + 1):
                Var[k, 0, j] = 2.0 * left_profiles[k][j - 1] - Var[k, 1, j]
    return Var


def bfs_inlet_profiles(ny, dy, step_height, h, Ub):
    """bfs_ml_accelerated.py:524-562 as row-wise Dirichlet values for u (k=0) and v (k=1)."""
    u = np.zeros(ny)
    for j in range(1, ny + 1):
        y = (j - 0.5) * dy
        if y >= step_height:
            yp = min(max(y - step_height, 0.0), h)
            u[j - 1] = 6.0 * Ub * (yp / h) * (1.0 - (yp / h))
    return {0: u, 1: np.zeros(ny)}
