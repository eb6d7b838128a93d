import sys, importlib, numpy as np
sys.path.insert(0, '.')
srcfd = importlib.import_module("sr-for-cfd_amd")
from oracle import sr_oracle as o
from oracle.sr_oracle_lowp import round_bf16
rng = np.random.default_rng(0)
def trunc(a, terms):
    a = a.astype(np.float32); rest = a.copy(); tot = np.zeros_like(a)
    for i in range(terms):
        t = (rest.view(np.uint32) & np.uint32(0xffff0000)).view(np.float32); tot += t; rest = (rest - t).astype(np.float32)
    return tot
for (k, cin, cout, hw) in ((3, 256, 128, 12), (2, 128, 64, 25)):
    w = (rng.standard_normal((k, k, cout, cin)) * 0.05).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
    x = rng.standard_normal((3, hw, hw, cin)).astype(np.float32)
    for tag, xx, ww in (("full", x, w), ("x,w bf16-exact", trunc(x, 1), trunc(w, 1)), ("x 2 terms, w 1", trunc(x, 2), trunc(w, 1)), ("x 1, w 2 terms", trunc(x, 1), trunc(w, 2)),
                        ("x 3, w 1", x, trunc(w, 1)), ("x 1, w 3", trunc(x, 1), w), ("x 2, w 2", trunc(x, 2), trunc(w, 2))):
        m = srcfd.SRModel.from_layers([dict(kind="conv2d_transpose", name="ct", k=k, stride=2, same=False, act="linear", w=ww, b=b)], (hw, hw, cin), device=0)
        ref = o.conv2d_transpose(xx.astype(np.float64), ww.astype(np.float64), b.astype(np.float64), 2, "valid", "linear")
        out = {}
        for p in ("fp32", "fp32x3"):
            m.precision = p
            m.set_profiling(True); y = m.predict(xx); names = [n for n, _ in m.get_profile()]; m.set_profiling(False)
            out[p] = o.rel_l2(y, ref)
        print(f"k={k} {tag:18s} fp32 {out['fp32']:.2e}  x3 {out['fp32x3']:.2e}  {names[:2]}")
