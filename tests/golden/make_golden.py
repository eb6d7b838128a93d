"""Regenerates tests/golden/golden_vectors.npz.

These vectors are produced by the float64 CPU oracle (oracle/sr_oracle.py), NOT
by the reference: TensorFlow/Keras are not installable in this image and the
reference's decoder weights are absent (.MISSING_LARGE_BLOBS:29-34), so no
reference output can be generated (SURVEY.md 8c -> "parity unpinned").  They
pin the oracle against regressions and give the GPU tests fixed targets:

  latent_*     encoder_10 latents for the 5 real coarse cases x (u,v,p), real
               multiBC encoder weights, LDC standardisation and BFS adaptive blend
  probe_*      64 fixed pixels + L2 norm + sum of the 400x400 output of
               real-encoder + synthetic decoder (seed 1) for 3 samples
  enc_checksum float64 sums of the encoder tensors as read by libsrcfd's HDF5 reader
  latent3_*    the same latents for ALL THREE trained encoder files of the reference, each with its own statistics file
               (keys latent3_<set>, x3_<set>; sets: conftest.ENCODER_SETS), + per-file tensor checksums enc3_checksum_<set>

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import srcfd_amd  # noqa: E402
from conftest import COARSE, ENCODER_H5, ENCODER_SETS, GOLDEN, STATS_TXT  # noqa: E402
from oracle import sr_oracle as o  # noqa: E402


def main():
    enc = srcfd_amd.SRModel.load_h5(ENCODER_H5, None, device=-1).weights()
    dec = o.synthetic_decoder(1)
    stats = o.parse_stats(STATS_TXT)
    lr, hr = o.component_stats(stats, 10, 400)
    xs, xs_blend = [], []
    for name, fn in COARSE.items():
        case = srcfd_amd.read_coarse_fields(os.path.join(GOLDEN, fn))
        for c in o.COMPONENTS:
            x = case[c].astype(np.float32)
            xs.append(o.standardize_with_stats(x, *lr[c]).astype(np.float32))
            mb, sb = o.adaptive_blend(x, lr[c][0], lr[c][1], 0.3)
            xs_blend.append(o.standardize_with_stats(x, mb, sb).astype(np.float32))
    x = np.stack(xs)[..., None]
    xb = np.stack(xs_blend)[..., None]
    out = {"x_std": x, "x_blend": xb,
           "latent_std": o.encoder_forward(x, enc, np.float64),
           "latent_blend": o.encoder_forward(xb, enc, np.float64)}
    rng = np.random.default_rng(1234)
    probe_idx = rng.integers(0, 400, size=(64, 2))
    probe_idx[:8] = [[0, 0], [0, 399], [399, 0], [399, 399], [1, 1], [24, 25], [199, 200], [398, 398]]
    y = o.superres_forward(x[:3], enc, dec, np.float64)[..., 0]
    out["probe_idx"] = probe_idx
    out["probe_val"] = y[:, probe_idx[:, 0], probe_idx[:, 1]]
    out["probe_l2"] = np.sqrt((y ** 2).sum(axis=(1, 2)))
    out["probe_sum"] = y.sum(axis=(1, 2))
    names = sorted(enc)
    out["enc_names"] = np.array(names)
    out["enc_checksum"] = np.array([enc[k].astype(np.float64).sum() for k in names])
    for key, (h5, txt) in ENCODER_SETS.items():
        enc3 = srcfd_amd.SRModel.load_h5(os.path.join(GOLDEN, h5), None, device=-1).weights()
        lr3, _ = o.component_stats(o.parse_stats(os.path.join(GOLDEN, txt)), 10, 400)
        xs3 = []
        for name, fn in COARSE.items():
            case = srcfd_amd.read_coarse_fields(os.path.join(GOLDEN, fn))
            for c in o.COMPONENTS:
                xs3.append(o.standardize_with_stats(case[c].astype(np.float32), *lr3[c]).astype(np.float32))
        x3 = np.stack(xs3)[..., None]
        out[f"x3_{key}"] = x3
        out[f"latent3_{key}"] = o.encoder_forward(x3, enc3, np.float64)
        out[f"enc3_checksum_{key}"] = np.array([enc3[k].astype(np.float64).sum() for k in sorted(enc3)])
    np.savez_compressed(os.path.join(GOLDEN, "golden_vectors.npz"), **out)
    print({k: getattr(v, "shape", None) for k, v in out.items()})


if __name__ == "__main__":
    main()
