"""Host-side data path feeding the hot path (reference: src/datasets/base.py, airfoil.py, cylinder_flow.py).

What is mirrored: per-trajectory multi-level mesh with the reference's on-disk cache format
(`[<file>_]mmesh_layer_{L}.dat`, a pickle of {"m_gs": [LongTensor [2,E_l]], "m_ids": [LongTensor]} -- caches written
by the reference load unchanged, base.py:98-122), field packing `[outputs..., mesh_pos, node_type]` (base.py:238-252),
the dataset masks (airfoil: type 0; cylinder: type 0 or 5), the training-noise injection (base.py:257-273) and the
per-sample packing for consistent / variable meshes (base.py:319-351).

Parity status: PINNED.  The reference datapipe imports h5py, torchdata and torch_geometric, which this image lacks;
tests/golden/make_golden_r2.py imports it with three stub modules (an h5py.File backed by .npz, an empty
IterDataPipe base, an attribute-bag torch_geometric.data.Data) and records what `airfoilDataPipe` / `cylinderDataPipe`
yield on small synthetic trajectories -- packed fields, masks, the seeded training noise, the (always shuffled) frame
order, rollout frames and the per-level Data lists -- into tests/golden/datapipe.npz; tests/test_datapipe.py requires
bit-equal output from this module.  Not pinned (third-party, absent): PyG's Batch collation (pinned by the
batched == per-graph equivalence fixture instead) and real HDF5 decoding.  Trajectories are read from `.npz` files
or in-memory dicts; `.h5` files are read when h5py is importable.
"""
import glob
import os
import pickle

import numpy as np
import torch

from .graph import LevelData, collate_variable_meshes
from .hierarchy import BistrideMultiLayerGraph, to_flat_edge

MASKS = {  # valid-for-loss node types (airfoil.py:23, cylinder_flow.py:24)
    "airfoil": lambda node_type: (node_type == 0).float(),
    "cylinder_flow": lambda node_type: torch.logical_or(node_type == 0, node_type == 5).float(),
}


def load_fields(path_or_dict, field_names):
    """One trajectory as {name: tensor [T,N,c]} ("cells" stays an int array), base.py:35-41."""
    if isinstance(path_or_dict, dict):
        raw = path_or_dict
    elif str(path_or_dict).endswith(".npz"):
        raw = np.load(path_or_dict)
    else:
        import h5py  # optional dependency
        with h5py.File(path_or_dict, "r") as f:
            raw = {k: np.array(f[k]) for k in field_names}
    out = {}
    for name in field_names:
        arr = np.asarray(raw[name])
        out[name] = arr if name == "cells" else torch.tensor(arr, dtype=torch.float)
    return out


class SingleTrajReader:
    """base.py:13-126."""

    def __init__(self, cfg, source, mode="train", cache_dir=None):
        self.cfg, self.mode = cfg, mode
        self.fields = load_fields(source, cfg.field_names)
        self.cells = self.fields["cells"][0]
        self.L = self.fields["mesh_pos"].shape[0] - 1          # the last frame has no target
        name = os.path.basename(source) if isinstance(source, (str, os.PathLike)) else "mem"
        self.cache_dir = cache_dir or (os.path.dirname(source) if isinstance(source, (str, os.PathLike)) else None)
        self.cache_file = None if self.cache_dir is None else os.path.join(
            self.cache_dir, f"{'' if cfg.consist_mesh else name + '_'}mmesh_layer_{cfg.unet_depth}.dat")
        self.m_gs, self.m_ids = self._multi_mesh(self.fields["mesh_pos"][0].clone().numpy())

    def __len__(self):
        return self.L

    def __getitem__(self, idx):
        return {k: v[idx] for k, v in self.fields.items()}, {k: v[idx + 1] for k, v in self.fields.items()}

    def _multi_mesh(self, mesh_pos):
        if self.cache_file and os.path.isfile(self.cache_file):
            with open(self.cache_file, "rb") as f:
                m = pickle.load(f)
            return m["m_gs"], m["m_ids"]
        flat = to_flat_edge(self.cells, self.cfg.mesh_type)
        _, m_es, m_ids = BistrideMultiLayerGraph(flat, self.cfg.unet_depth, mesh_pos.shape[0], mesh_pos).get_multi_layer_graphs()
        m_gs = [torch.tensor(np.asarray(g), dtype=torch.long) for g in m_es]
        m_ids = [torch.tensor(np.asarray(i), dtype=torch.long) for i in m_ids]
        if self.cache_file:
            with open(self.cache_file, "wb") as f:
                pickle.dump({"m_gs": m_gs, "m_ids": m_ids}, f)
        return m_gs, m_ids


def proc_data(cfg, data, mask_fn, train, tc_rng=None):
    """base.py:219-275: pack fields, build the mask, inject training noise.  Returns (node_in, node_tar, node_mask)."""
    fields_inp, fields_tar = data
    keys_out = list(cfg.output_field_names)
    node_in = torch.cat([fields_inp[k] for k in [*keys_out, "mesh_pos", "node_type"]], dim=-1)
    node_tar = torch.cat([fields_tar[k] for k in keys_out], dim=-1)
    node_mask = mask_fn(fields_inp["node_type"])
    if train:
        std = node_tar.new_ones(node_tar.shape, dtype=torch.float64)
        std[..., :] = torch.tensor(np.array(list(cfg.noise_level)))
        noise = torch.normal(mean=node_tar.new_zeros(node_tar.shape), std=std, generator=tc_rng)
        noise = torch.where((node_mask == 0).bool(), torch.zeros_like(noise), noise)   # Dirichlet nodes stay clean
        node_in = node_in.clone()
        node_in[..., : noise.shape[-1]] += noise
        node_tar = node_tar + (1.0 - cfg.noise_gamma) * noise
    return node_in, node_tar, node_mask


def pack_levels(node_in, node_tar, node_mask, m_gs, m_ids):
    """Variable-mesh sample: one LevelData per level (the PyG `Data` list of base.py:325-349)."""
    out = [LevelData(m_gs[0], node_in.shape[-2], face=m_ids[0] if m_ids else None, x=node_in, y=node_tar, mask=node_mask)]
    for i in range(1, len(m_gs)):
        out.append(LevelData(m_gs[i], m_ids[i - 1].shape[0], face=m_ids[i] if i < len(m_gs) - 1 else None))
    return out


class TrajectoryDataset(torch.utils.data.IterableDataset):
    """base.py:128-357 without the torchdata dependency: iterates over trajectories and frames, yields the
    consistent-mesh tuple or a per-level LevelData list; `mode="rollout"` yields whole trajectories.  Like the
    reference, trajectories AND frames are shuffled in every mode (its `if self.rng is not None` guards are always
    true, base.py:300-314), frames stay ordered only in rollout mode; noise is injected in train mode only.
    `seed` is what the reference derives as train_seed + 1000 * worker_id + 10^6 * base_seed (base.py:176-183)."""

    def __init__(self, cfg, sources, dataset="airfoil", mode="train", seed=0, cache_dir=None):
        self.cfg, self.sources, self.mode, self.cache_dir = cfg, list(sources), mode, cache_dir
        self.mask_fn = MASKS[dataset]
        self.rng = np.random.default_rng(seed)
        self.tc_rng = torch.Generator().manual_seed(seed)

    @classmethod
    def from_dir(cls, cfg, data_dir, **kw):
        return cls(cfg, sorted(glob.glob(os.path.join(data_dir, "*.npz")) + glob.glob(os.path.join(data_dir, "*.h5"))), **kw)

    def __iter__(self):
        train, rollout = self.mode == "train", self.mode == "rollout"
        order = list(range(len(self.sources)))
        self.rng.shuffle(order)
        for si in order:
            reader = SingleTrajReader(self.cfg, self.sources[si], self.mode, self.cache_dir)
            t_ids = np.arange(len(reader))
            if rollout:
                yield (*proc_data(self.cfg, reader[t_ids], self.mask_fn, False), reader.m_gs, reader.m_ids)
                continue
            self.rng.shuffle(t_ids)
            for ti in t_ids:
                node_in, node_tar, node_mask = proc_data(self.cfg, reader[int(ti)], self.mask_fn, train, self.tc_rng)
                if self.cfg.consist_mesh:
                    yield node_in, node_tar, node_mask, reader.m_gs, reader.m_ids
                else:
                    yield pack_levels(node_in, node_tar, node_mask, reader.m_gs, reader.m_ids)


def collate_consistent(samples):
    """default_collate of the consistent-mesh tuples (train.py:50): stack everything along a new batch axis."""
    node_in = torch.stack([s[0] for s in samples])
    node_tar = torch.stack([s[1] for s in samples])
    mask = torch.stack([s[2] for s in samples])
    m_gs = [torch.stack([s[3][l] for s in samples]) for l in range(len(samples[0][3]))]
    m_ids = [torch.stack([s[4][l] for s in samples]) for l in range(len(samples[0][4]))]
    return node_in, node_tar, mask, m_gs, m_ids


def make_loader(dataset, batch_size):
    collate = collate_consistent if dataset.cfg.consist_mesh else collate_variable_meshes
    return torch.utils.data.DataLoader(dataset, batch_size=batch_size, collate_fn=collate)
