"""Training-data I/O: simulation_result schema reader, per-BC Reynolds split,
component-wise statistics and the stats-file writer, against plain numpy on the same arrays."""
import importlib

import numpy as np
import pytest


@pytest.fixture()
def sim_files(srcfd, tmp_path):
    ds = importlib.import_module("sr-for-cfd_amd.datasets")
    h5 = importlib.import_module("sr-for-cfd_amd.h5")
    rng = np.random.default_rng(11)
    truth = {}
    paths = []
    for fname, bc, res in (("simulation_result.h5", "single_lid(u_top=1)", (100, 200, 800)),
                           ("simulation_result_double_lid.h5", "double_lid(u_top=1,u_bottom=1)", (100, 800))):
        w = h5.H5Writer()
        for Re in res:
            for n in (10, 50, 40):  # 50 is a mesh the loader must ignore
                f = {c: rng.standard_normal((n, n)) * (1 + "uvp".index(c)) + 0.1 * Re / 100 for c in "uvp"}
                ds.append_solution(w, Re, n, f, bc, "case")
                truth[(bc, Re, n)] = f
        if bc.startswith("single"):  # a Reynolds number with only the coarse mesh: not paired, not loaded
            ds.append_solution(w, 900, 10, {c: rng.standard_normal((10, 10)) for c in "uvp"}, bc)
        p = str(tmp_path / fname)
        w.save(p)
        paths.append(p)
    return paths, truth


def test_loader_pairs_every_reynolds_and_component(srcfd, sim_files):
    ds = importlib.import_module("sr-for-cfd_amd.datasets")
    paths, truth = sim_files
    x_lr, x_hr, res, comps, bcs = ds.load_paired_reynolds_multi(paths + ["/no/such/file.h5"], 10, 40)
    assert x_lr.shape == (15, 10, 10, 1) and x_hr.shape == (15, 40, 40, 1) and x_lr.dtype == np.float32
    assert res.tolist() == [100] * 3 + [200] * 3 + [800] * 3 + [100] * 3 + [800] * 3
    assert comps.tolist() == list("uvp") * 5
    assert set(bcs[:9]) == {"single_lid(u_top=1)"} and set(bcs[9:]) == {"double_lid(u_top=1,u_bottom=1)"}
    np.testing.assert_array_equal(x_hr[4, ..., 0], truth[("single_lid(u_top=1)", 200, 40)]["v"].astype(np.float32))
    np.testing.assert_array_equal(x_lr[14, ..., 0], truth[("double_lid(u_top=1,u_bottom=1)", 800, 10)]["p"].astype(np.float32))


def test_split_stats_and_stats_file(srcfd, sim_files, tmp_path):
    ds = importlib.import_module("sr-for-cfd_amd.datasets")
    paths, _ = sim_files
    cfg = {"single_lid(u_top=1)": {"train": [100, 200], "test": [800], "evaluate": [800]},
           "double_lid(u_top=1,u_bottom=1)": {"train": "ALL", "test": [800], "evaluate": [800]}}
    d = ds.prepare_training_set(paths, 10, 40, cfg)
    assert sorted(d["res_train"].tolist()) == sorted([100] * 3 + [200] * 3 + [100] * 3 + [800] * 3)
    assert sorted(d["res_test"].tolist()) == [800] * 6 and d["reynolds_to_evaluate"] == [800]
    x_lr, x_hr, res, comps, bcs = ds.load_paired_reynolds_multi(paths, 10, 40)
    tr, te, _ = ds.split_by_reynolds(res, bcs, cfg)
    for c in "uvp":
        m = comps[tr] == c
        raw = x_hr[tr][m]
        mean, std = float(np.mean(raw, dtype=np.float64)), float(np.std(raw, dtype=np.float64))
        assert d["stats_hr"][c] == (mean, std)
        np.testing.assert_array_equal(d["x_hr_train"][m], (raw - mean) / std)
        # test data is standardised with the TRAINING statistics
        mt = comps[te] == c
        np.testing.assert_array_equal(d["x_lr_test"][mt], (x_lr[te][mt] - d["stats_lr"][c][0]) / d["stats_lr"][c][1])
    # standardised training set has zero mean / unit variance per component
    assert abs(float(np.mean(d["x_lr_train"][comps[tr] == "u"], dtype=np.float64))) < 1e-6
    p = str(tmp_path / "standardization_stats_10to40_test.txt")
    ds.save_component_stats(p, 10, 40, d["stats_lr"], d["stats_hr"])
    lr, hr = srcfd.load_stats(p, 10, 40)
    assert lr == d["stats_lr"] and hr == d["stats_hr"]  # repr round trip is exact


def test_dummy_recipe_when_nothing_loads(srcfd):
    ds = importlib.import_module("sr-for-cfd_amd.datasets")
    x_lr, x_hr, res, comps, bcs = ds.load_paired_reynolds_multi(["/no/such/file.h5"], 10, 40)
    assert x_lr.shape == (60, 10, 10, 1) and x_hr.shape == (60, 40, 40, 1) and set(bcs) == {"dummy"}
    np.testing.assert_allclose(x_lr[7, 2, 3, 0], x_hr[7, 8:12, 12:16, 0].mean(), rtol=1e-6)
    with pytest.raises(ValueError):
        ds.dummy_pairs(10, 45)
