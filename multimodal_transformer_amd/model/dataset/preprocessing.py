"""Batch construction with the reference's interface (``Transformer/model/dataset/preprocessing.py:79-270``), vectorised
(SURVEY 8 f2): the per-(node, timestep) Python loop of ``get_node_timestep_data`` becomes ONE gather + standardise kernel
(``mmt_batch_gather``) over a packed table of the scene's node states that is uploaded once per scene.

``get_timesteps_data`` keeps the reference's signature and return value -- ``(batch, nodes, out_timesteps)`` with
``batch = (first_history_index, x, y, x_st, y_st, None, None, None, lanes, map)`` -- so it plugs into
``Trajectron(get_timesteps_data=...)``; the tensors already live on the GPU.  ``scene`` is anything with ``.nodes`` whose
elements have ``.type``, ``.first_timestep``, ``.is_robot`` and ``.data[:, state]`` (the reference's Scene / Node do).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ... import _lib
from ..._lib import MmtError

STANDARDIZE_DIVISOR = 80.0        # environment.py:61: (array - mean) / 80, the configured std is ignored


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _state_columns(state_of_type):
    return [(e, d) for e, dims in state_of_type.items() for d in dims]


class SceneTable:
    """The state columns of every node of one type in a scene, packed node after node (fp64, as the reference holds them),
    plus first / last timestep per node -- resident on the device."""

    def __init__(self, tables, firsts, nodes, device, lane_tables=None):
        self.nodes = list(nodes)
        lengths = np.array([len(t) for t in tables], dtype=np.int64)
        if len(tables) == 0:
            raise MmtError("SceneTable: no nodes")
        self.ncol = int(tables[0].shape[1])
        self.first_h = np.asarray(firsts, dtype=np.int32)
        self.last_h = (self.first_h + lengths - 1).astype(np.int32)
        self.row_off_h = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.int64)
        flat = np.ascontiguousarray(np.concatenate(tables, axis=0), dtype=np.float64)
        self.flat_h = flat                      # host copy: fp64 map points / headings for the patch extraction
        self.device = torch.device(device)
        self.table = torch.from_numpy(flat).to(self.device)
        self.row_off = torch.from_numpy(self.row_off_h).to(self.device)
        self.first = torch.from_numpy(self.first_h).to(self.device)
        self.last = torch.from_numpy(self.last_h).to(self.device)
        # lane tables (Node.lanes_point, node.py:119-121): one (n_lanes, n_points, 2) block per state row, metres
        self.lanes = None
        if lane_tables is not None:
            lt = [np.asarray(t, dtype=np.float64) for t in lane_tables]
            if any(t.ndim != 4 or t.shape[1:] != lt[0].shape[1:] or t.shape[3] != 2 or len(t) < n for t, n in zip(lt, lengths)):
                raise MmtError("SceneTable: every node needs a lane table of shape (timesteps, n_lanes, n_points, 2)")
            self.n_lanes, self.n_points = int(lt[0].shape[1]), int(lt[0].shape[2])
            flat_l = np.ascontiguousarray(np.concatenate([t[:n] for t, n in zip(lt, lengths)], axis=0))
            self.lanes = torch.from_numpy(flat_l).to(self.device)

    @classmethod
    def from_scene(cls, scene, node_type, state_of_type, device="cuda", return_robot=True):
        """One host pass over the nodes (``node.data[:, state]``: environment/data_structures.py DoubleHeaderNumpyArray)."""
        nodes = [n for n in scene.nodes if n.type == node_type and (return_robot or not getattr(n, "is_robot", False))]
        tables = [np.asarray(n.data[:, state_of_type], dtype=np.float64) for n in nodes]
        lanes = [getattr(n, "lanes_point", None) for n in nodes]
        return cls(tables, [int(n.first_timestep) for n in nodes], nodes, device,
                   lane_tables=lanes if nodes and all(l is not None for l in lanes) else None)


def present_nodes(first, last, timesteps, min_history_timesteps=0, min_future_timesteps=0):
    """Scene.present_nodes (environment/scene.py:103-140) for all nodes at once: returns (node_idx, ts) int32 arrays in the
    order the reference's nested loops visit them -- timesteps in the order a node first makes them appear, nodes in scene order."""
    ts = np.asarray(timesteps).reshape(-1)
    mask = (first[:, None] <= ts[None, :] - min_history_timesteps) & (ts[None, :] + min_future_timesteps <= last[:, None])
    cols = np.nonzero(mask.any(axis=0))[0]
    if cols.size == 0:
        return np.zeros(0, np.int32), np.zeros(0, np.int32)
    first_node = mask[:, cols].argmax(axis=0)
    cols = cols[np.lexsort((cols, first_node))]          # dict insertion order of scene.py:133-138
    node_idx = [np.nonzero(mask[:, j])[0] for j in cols]
    out_ts = [np.full(len(n), ts[j]) for n, j in zip(node_idx, cols)]
    return np.concatenate(node_idx).astype(np.int32), np.concatenate(out_ts).astype(np.int32)


def gather(table: SceneTable, node_idx, ts, max_ht, max_ft, pred_cols=(0, 1)):
    """(first_history_index, x, y, x_st, y_st) device tensors for the (node, timestep) pairs: preprocessing.py:100-123."""
    B = len(node_idx)
    dev = table.device
    if dev.type != "cuda":
        raise MmtError("batch construction runs on the GPU (no CPU fallback): build the SceneTable with device='cuda'")
    ni = torch.from_numpy(np.ascontiguousarray(node_idx, dtype=np.int32)).to(dev)
    tt = torch.from_numpy(np.ascontiguousarray(ts, dtype=np.int32)).to(dev)
    npred = len(pred_cols)
    fhi = torch.empty(B, dtype=torch.int64, device=dev)
    x = torch.empty(B, max_ht + 1, table.ncol, dtype=torch.float32, device=dev)
    y = torch.empty(B, max_ft, npred, dtype=torch.float32, device=dev)
    x_st, y_st = torch.empty_like(x), torch.empty_like(y)
    pc = (C.c_int32 * npred)(*[int(c) for c in pred_cols])
    lib = _lib.load()
    p = lambda t: C.c_void_p(t.data_ptr())
    _lib.check(lib.mmt_batch_gather(p(table.table), p(table.row_off), p(table.first), p(table.last), p(ni), p(tt), B, int(max_ht),
                                    int(max_ft), table.ncol, pc, npred, STANDARDIZE_DIVISOR, p(fhi), p(x), p(y), p(x_st), p(y_st),
                                    _stream()), "mmt_batch_gather")
    return fhi, x, y, x_st, y_st


def gather_lanes(table: SceneTable, node_idx, ts, max_ft):
    """[lane_input (B,Ln,P,2) f32, lane_label (B,max_ft,Ln) f32, lane_t_mask (B,Ln) bool] on the device: the lane tuple of
    preprocessing.py:170-202 after collate (:43-45) and the stacking of trajectron.py:62-65."""
    if table.lanes is None:
        raise MmtError("lane batches need Node.lanes_point on every node of the scene (Node.get_lane_id, node.py:119-121)")
    B, dev = len(node_idx), table.device
    ni = torch.from_numpy(np.ascontiguousarray(node_idx, dtype=np.int32)).to(dev)
    tt = torch.from_numpy(np.ascontiguousarray(ts, dtype=np.int32)).to(dev)
    Ln, P = table.n_lanes, table.n_points
    lane_input = torch.empty(B, Ln, P, 2, dtype=torch.float32, device=dev)
    lane_label = torch.empty(B, max_ft, Ln, dtype=torch.float32, device=dev)
    lane_mask = torch.empty(B, Ln, dtype=torch.uint8, device=dev)
    lib = _lib.load()
    p = lambda t: C.c_void_p(t.data_ptr())
    _lib.check(lib.mmt_lane_gather(p(table.table), p(table.lanes), p(table.row_off), p(table.first), p(table.last), p(ni), p(tt), B,
                                   int(max_ft), table.ncol, Ln, P, STANDARDIZE_DIVISOR, p(lane_input), p(lane_label), p(lane_mask),
                                   _stream()), "mmt_lane_gather")
    return [lane_input, lane_label, lane_mask.bool()]


def _table_for(scene, node_type, state_of_type, device, return_robot):
    cache = scene.__dict__.setdefault("_mmt_scene_tables", {})
    key = (str(node_type), tuple(_state_columns(state_of_type)), str(device), bool(return_robot))
    if key not in cache:
        cache[key] = SceneTable.from_scene(scene, node_type, state_of_type, device, return_robot)
    return cache[key]


def get_timesteps_data(env, scene, t, node_type, state, pred_state, edge_types, min_ht, max_ht, min_ft, max_ft, hyperparams,
                       device="cuda"):
    """Inputs for ALL nodes of ``node_type`` present at the timesteps ``t`` (preprocessing.py:208-270).  Presence needs
    ``min_ht`` history and ``max_ft`` future steps (:239-240; ``min_ft`` is accepted and unused, like in the reference)."""
    if hyperparams.get("incl_robot_node"):
        raise MmtError("incl_robot_node batches are not part of the Transformer hot path (no README config sets it)")
    st, ps = state[node_type], pred_state[node_type]
    if list(ps.keys())[0] != "position":
        raise MmtError("only position prediction states are supported (the reference's configs predict position)")
    cols = _state_columns(st)
    if cols[:2] != [("position", "x"), ("position", "y")]:
        raise MmtError("the state must start with position x, y")
    pred_cols = [cols.index(c) for c in _state_columns(ps)]
    table = _table_for(scene, node_type, st, device, return_robot=not hyperparams.get("incl_robot_node"))
    node_idx, out_ts = present_nodes(table.first_h, table.last_h, np.asarray(t), min_ht, max_ft)
    if node_idx.size == 0:
        return None
    fhi, x, y, x_st, y_st = gather(table, node_idx, out_ts, max_ht, max_ft, pred_cols)
    nodes = [table.nodes[i] for i in node_idx]
    maps = lanes = None
    if hyperparams.get("map_cnn_encoding") or hyperparams.get("map_vit_encoding"):
        cur = table.flat_h[table.row_off_h[node_idx] + (out_ts - table.first_h[node_idx])]     # x[-1] in fp64 (:150,166)
        maps = _map_patches(scene, node_type, len(nodes), cur, hyperparams, str(x.device))
    elif hyperparams.get("lane_cnn_encoding"):
        lanes = gather_lanes(table, node_idx, out_ts, max_ft)
    return (fhi, x, y, x_st, y_st, None, None, None, lanes, maps), nodes, [int(v) for v in out_ts]


def get_node_timestep_data(env, scene, t, node, state, pred_state, edge_types, max_ht, max_ft, hyperparams, scene_graph=None,
                           device="cuda"):
    """One batch element (preprocessing.py:79-205), as a batch of one."""
    st = state[node.type]
    table = _table_for(scene, node.type, st, device, True)
    idx = next(i for i, n in enumerate(table.nodes) if n is node or n == node)
    cols = _state_columns(st)
    pred_cols = [cols.index(c) for c in _state_columns(pred_state[node.type])]
    fhi, x, y, x_st, y_st = gather(table, np.array([idx]), np.array([int(t)]), max_ht, max_ft, pred_cols)
    lane_tuple = map_tuple = None
    if hyperparams.get("map_cnn_encoding") or hyperparams.get("map_vit_encoding"):
        # (scene map, current position, heading, patch size): cropped per batch by collate (preprocessing.py:147-169, :32-40)
        me = hyperparams["map_encoder"]["cnn_param"]
        cur = table.flat_h[table.row_off_h[idx] + (int(t) - table.first_h[idx])]
        heading = None
        if "heading_state_index" in me:
            hsi = me["heading_state_index"]
            heading = float(-np.arctan2(cur[hsi[1]], cur[hsi[0]]) * 180 / np.pi if isinstance(hsi, (list, tuple))
                            else -cur[hsi] * 180 / np.pi)
        map_tuple = (scene.map[node.type], cur[:2].copy(), heading, me["patch_size"])
    elif hyperparams.get("lane_cnn_encoding"):
        lane_tuple = tuple(v[0] for v in gather_lanes(table, np.array([idx]), np.array([int(t)]), max_ft))
    return fhi[0], x[0], y[0], x_st[0], y_st[0], None, None, None, lane_tuple, map_tuple


def restore(data):
    """preprocessing.py:9-19: structures that a DataLoader worker serialised with dill come back; anything else passes through."""
    if isinstance(data, bytes):
        import dill
        return dill.loads(data)
    return data


def collate(batch):
    """``collate_fn`` for per-node elements of get_node_timestep_data (preprocessing.py:22-55): tensors are stacked; a 4-tuple
    element is (scene map, point, heading, patch size) and becomes the batch of cropped map patches (one ``mmt_map_crop`` launch); a
    3-tuple element is the lane tuple and becomes three per-node tuples; other sequences are collated field by field; dict
    elements become a dict of lists (dill-serialised inside a DataLoader worker, see ``restore``)."""
    import collections.abc as abc
    if len(batch) == 0:
        return batch
    first = batch[0]
    if first is None:
        return None
    if isinstance(first, abc.Sequence) and not isinstance(first, (str, bytes)):
        if len(first) == 4:
            maps, points, headings, patch = zip(*batch)
            rotation = None if headings[0] is None else torch.tensor(headings)
            return maps[0].get_cropped_maps_from_scene_map_batch(maps, scene_pts=torch.tensor(np.array(points)),
                                                                 patch_size=patch[0], rotation=rotation)
        if len(first) == 3:
            return [field for field in zip(*batch)]
        return [collate(field) for field in zip(*batch)]
    if isinstance(first, abc.Mapping):
        merged = {key: [item[key] for item in batch] for key in first}
        if torch.utils.data.get_worker_info():
            import dill
            return dill.dumps(merged)
        return merged
    return torch.utils.data.default_collate(batch)


def _map_patches(scene, node_type, n, cur, hyperparams, device):
    """preprocessing.py:147-169 + collate (:32-40): rotate by minus the heading, crop around the current position.
    ``cur`` (n, ncol) fp64 = the nodes' current states."""
    me = hyperparams["map_encoder"]["cnn_param"]
    heading = None
    if "heading_state_index" in me:
        hsi = me["heading_state_index"]
        if isinstance(hsi, (list, tuple)):
            heading = -np.arctan2(cur[:, hsi[1]], cur[:, hsi[0]]) * 180 / np.pi
        else:
            heading = -cur[:, hsi] * 180 / np.pi
        heading = torch.tensor(heading)
    scene_map = scene.map[node_type]
    return scene_map.get_cropped_maps_from_scene_map_batch([scene_map] * n, scene_pts=torch.tensor(cur[:, :2]),
                                                           patch_size=me["patch_size"], rotation=heading, device=device)
