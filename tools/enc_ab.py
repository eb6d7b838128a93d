"""A/B of the one-launch encoder (enc16) against the layer-by-layer chain (SRCFD_ENC=0): rel-L2 of both against the 16-bit
emulation and the float64 oracle on the 15 real coarse inputs.  Diagnostic; run on the GPU box: python tools/enc_ab.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import srcfd_amd as srcfd
from oracle import sr_oracle as oracle, sr_oracle_lowp as lp
from conftest import ENCODER_H5, STATS_TXT, GOLDEN, COARSE

enc = srcfd.SRModel.load_h5(ENCODER_H5, None, device=-1).weights()
dec = oracle.synthetic_decoder(1)
lr, _ = srcfd.load_stats(STATS_TXT, 10, 400)
xs = []
for v in COARSE.values():
    case = srcfd.read_coarse_fields(os.path.join(GOLDEN, v))
    for c in "uvp":
        xs.append(((case[c].astype(np.float32) - lr[c][0]) / lr[c][1]).astype(np.float32))
x = np.stack(xs)[..., None]
f64 = oracle.superres_forward(x, enc, dec, np.float64)
for kind in ("f16", "bf16"):
    emu = lp.superres_forward_lowp(x, enc, dec, kind)
    m = srcfd.SRModel.from_weights(enc, dec, device=0)
    m.precision = kind
    for e in ("1", "0"):
        os.environ["SRCFD_ENC"] = e
        y = m.predict(x)
        per = [float(np.linalg.norm(y[i] - emu[i]) / np.linalg.norm(emu[i])) for i in range(len(y))]
        print(f"{kind} SRCFD_ENC={e}: vs emulation {oracle.rel_l2(y, emu):.3e}  vs f64 {oracle.rel_l2(y, f64):.3e}  per sample vs emulation: "
              + " ".join(f"{v:.1e}" for v in per), flush=True)
    os.environ.pop("SRCFD_ENC")
