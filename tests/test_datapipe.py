"""CPU: the host data path (bsms_gnn_amd.datapipe) against golden vectors recorded from the reference's own datapipes
(datasets/base.py, airfoil.py, cylinder_flow.py; tests/golden/make_golden_r2.py -> tests/golden/datapipe.npz), plus
self-consistency checks of the cache format and the loaders."""
import pickle
from types import SimpleNamespace

import numpy as np
import pytest
import torch
from scipy.spatial import Delaunay


@pytest.fixture(scope="module")
def dp():
    import __graft_entry__
    __graft_entry__.build()
    import bsms_gnn_amd.datapipe as dp
    return dp


def synthetic_traj(n, T, seed, node_types=(0, 0, 0, 4, 5)):
    rng = np.random.default_rng(seed)
    pts = rng.random((n, 2)).astype(np.float32)
    cells = Delaunay(pts).simplices.astype(np.int64)
    return {"cells": np.repeat(cells[None], T, 0), "mesh_pos": np.repeat(pts[None], T, 0),
            "node_type": np.repeat(rng.choice(node_types, (1, n, 1)).astype(np.float32), T, 0),
            "velocity": rng.standard_normal((T, n, 2)).astype(np.float32),
            "density": rng.standard_normal((T, n, 1)).astype(np.float32)}


def cfg(consist, depth=2, gamma=1.0):
    return SimpleNamespace(field_names=["node_type", "cells", "mesh_pos", "density", "velocity"],
                           output_field_names=["velocity", "density"], mesh_type="tri", unet_depth=depth,
                           consist_mesh=consist, noise_level=[10, 10, 0.01], noise_gamma=gamma)


def test_packing_masks_and_noise(dp):
    traj = synthetic_traj(60, 4, 0)
    c = cfg(True, gamma=0.8)
    reader = dp.SingleTrajReader(c, traj)
    assert len(reader) == 3
    clean_in, clean_tar, mask = dp.proc_data(c, reader[1], dp.MASKS["airfoil"], train=False)
    # [velocity(2), density(1), mesh_pos(2), node_type(1)], target = next frame's outputs
    assert clean_in.shape == (60, 6) and clean_tar.shape == (60, 3)
    assert torch.equal(clean_in[:, :2], torch.tensor(traj["velocity"][1])) and torch.equal(clean_in[:, 3:5], torch.tensor(traj["mesh_pos"][1]))
    assert torch.equal(clean_tar[:, 2:], torch.tensor(traj["density"][2]))
    assert torch.equal(mask, (torch.tensor(traj["node_type"][1]) == 0).float())
    cyl = dp.MASKS["cylinder_flow"](torch.tensor(traj["node_type"][1]))
    assert float(cyl.sum()) == float(((traj["node_type"][1] == 0) | (traj["node_type"][1] == 5)).sum())
    gen = torch.Generator().manual_seed(3)
    noisy_in, noisy_tar, _ = dp.proc_data(c, reader[1], dp.MASKS["airfoil"], train=True, tc_rng=gen)
    noise = noisy_in[:, :3] - clean_in[:, :3]
    assert torch.all(noise[mask[:, 0] == 0] == 0)                            # Dirichlet / boundary nodes stay clean
    assert torch.equal(noisy_in[:, 3:], clean_in[:, 3:])                     # mesh_pos and node_type untouched
    torch.testing.assert_close(noisy_tar - clean_tar, 0.2 * noise, rtol=1e-5, atol=1e-6)   # (1 - gamma) * noise
    live = noise[mask[:, 0] == 1]
    assert 5 < float(live[:, 0].std()) < 20 and float(live[:, 2].std()) < 0.05             # per-channel noise levels


def test_cache_format_round_trip_and_reference_cache(dp, tmp_path):
    traj = synthetic_traj(80, 3, 1)
    c = cfg(True)
    r1 = dp.SingleTrajReader(c, traj, cache_dir=str(tmp_path))
    path = tmp_path / "mmesh_layer_2.dat"
    assert path.exists()
    blob = pickle.load(open(path, "rb"))
    assert set(blob) == {"m_gs", "m_ids"} and blob["m_gs"][0].dtype == torch.long and len(blob["m_ids"]) == 2
    # a cache written in the reference's format is picked up verbatim
    fake = {"m_gs": [torch.zeros(2, 1, dtype=torch.long)] * 3, "m_ids": [torch.zeros(1, dtype=torch.long)] * 2}
    pickle.dump(fake, open(path, "wb"))
    r2 = dp.SingleTrajReader(c, traj, cache_dir=str(tmp_path))
    assert r2.m_gs[0].shape == (2, 1) and r1.m_gs[0].shape[1] > 100
    # variable meshes: one cache per trajectory file name
    np.savez(tmp_path / "traj7.npz", **traj)
    dp.SingleTrajReader(cfg(False), str(tmp_path / "traj7.npz"))
    assert (tmp_path / "traj7.npz_mmesh_layer_2.dat").exists()


def test_loaders_feed_the_model_layouts(dp):
    c_con, c_var = cfg(True), cfg(False)
    same = synthetic_traj(50, 3, 2)
    ds = dp.TrajectoryDataset(c_con, [same, same], dataset="airfoil", mode="train", seed=1)
    batch = next(iter(dp.make_loader(ds, 3)))
    node_in, node_tar, mask, m_gs, m_ids = batch
    assert node_in.shape == (3, 50, 6) and mask.shape == (3, 50, 1) and m_gs[0].shape[0] == 3 and m_ids[0].shape[0] == 3
    assert torch.equal(m_gs[1][0], m_gs[1][2])                                # one shared mesh, stacked (model uses [0])
    ds = dp.TrajectoryDataset(c_var, [synthetic_traj(40, 2, 5), synthetic_traj(55, 2, 6)], dataset="cylinder_flow", mode="val")
    levels = next(iter(dp.make_loader(ds, 2)))
    assert [d.num_nodes for d in levels][0] == 95 and levels[0].x.shape == (95, 6) and levels[-1].face is None
    assert int(levels[0].edge_index.max()) < 95 and int(levels[1].edge_index.max()) < levels[1].num_nodes
    assert int(levels[0].face.max()) < 95 and levels[1].num_nodes == levels[0].face.numel()
    roll = next(iter(dp.TrajectoryDataset(c_con, [same], mode="rollout")))
    assert roll[0].shape == (2, 50, 6) and roll[1].shape == (2, 50, 3)       # whole trajectory, no noise


# ------------------------------------------------------------------------- pinned against the reference datapipes
def _golden_traj(z, name):
    return {k: z.np(f"in/{name}/{k}") for k in ("cells", "mesh_pos", "node_type", "velocity", "density")}


@pytest.mark.parametrize("tag,mode", [("air_train", "train"), ("air_val", "val")])
def test_consistent_mesh_yields_match_reference(dp, tag, mode):
    """airfoilDataPipe, consistent mesh: packed fields, mask, seeded noise (train) and the frame order (the reference
    shuffles frames in EVERY mode) are bit-equal to what the reference yields (datasets/base.py:238-273,300-323)."""
    from conftest import load_golden
    z = load_golden("datapipe")
    ds = dp.TrajectoryDataset(cfg(True, gamma=0.8), [_golden_traj(z, "trajA")], dataset="airfoil", mode=mode,
                              seed=int(z.np(f"{tag}/seed")))
    items = list(ds)
    assert len(items) == int(z.np(f"{tag}/n")) == 3
    es, ids = z.levels("air_train/mesh")
    for i, (x, y, m, gs, mids) in enumerate(items):
        assert x.dtype == y.dtype == torch.float32
        assert torch.equal(x, z.t(f"{tag}/{i}/x")) and torch.equal(y, z.t(f"{tag}/{i}/y")) and torch.equal(m, z.t(f"{tag}/{i}/mask"))
        assert all(torch.equal(a, b) for a, b in zip(mids, ids)) and torch.equal(gs[0], es[0])
        assert all(set(map(tuple, a.T.tolist())) == set(map(tuple, b.T.tolist())) for a, b in zip(gs[1:], es[1:]))


def test_rollout_yield_matches_reference(dp):
    from conftest import load_golden
    z = load_golden("datapipe")
    (x, y, m, gs, ids), = list(dp.TrajectoryDataset(cfg(True, gamma=0.8), [_golden_traj(z, "trajA")], mode="rollout", seed=1))
    assert torch.equal(x, z.t("air_roll/x")) and torch.equal(y, z.t("air_roll/y")) and torch.equal(m, z.t("air_roll/mask"))


def test_variable_mesh_yields_match_reference(dp):
    """cylinderDataPipe, consist_mesh: false: trajectory order, frame order, noise and the per-level Data lists
    (num_nodes of level l = len(m_ids[l-1]); `face` carries m_ids and is absent on the last level, base.py:325-349)."""
    from conftest import load_golden
    z = load_golden("datapipe")
    files = [str(f) for f in z.np("cyl_train/files")]
    ds = dp.TrajectoryDataset(cfg(False, gamma=1.0), [_golden_traj(z, f) for f in files], dataset="cylinder_flow",
                              mode="train", seed=int(z.np("cyl_train/seed")))
    items = list(ds)
    assert len(items) == int(z.np("cyl_train/n")) == 5
    for i, levels in enumerate(items):
        assert len(levels) == int(z.np(f"cyl_train/{i}/n_levels"))
        assert torch.equal(levels[0].x, z.t(f"cyl_train/{i}/x")) and torch.equal(levels[0].y, z.t(f"cyl_train/{i}/y"))
        assert torch.equal(levels[0].mask, z.t(f"cyl_train/{i}/mask"))
        assert torch.equal(levels[0].edge_index, z.t(f"cyl_train/{i}/0/edge_index"))
        for l, d in enumerate(levels):
            assert d.num_nodes == int(z.np(f"cyl_train/{i}/{l}/num_nodes"))
            assert (d.face is not None) == bool(z.np(f"cyl_train/{i}/{l}/has_face"))
            if d.face is not None:
                assert torch.equal(d.face, z.t(f"cyl_train/{i}/{l}/face"))
            ref_e = z.np(f"cyl_train/{i}/{l}/edge_index")
            assert set(map(tuple, d.edge_index.numpy().T.tolist())) == set(map(tuple, ref_e.T.tolist()))
