"""Batch construction on the GPU: mmt_batch_gather through the reference-interface mirror
(multimodal_transformer_amd/model/dataset/preprocessing.py) == the reference's own outputs (tests/golden/batch.npz) and
== the numpy oracle on a larger seeded scene.  Bit-exact, NaN padding included."""
import json
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STATE = {"VEHICLE": {"position": ["x", "y"], "velocity": ["x", "y"], "acceleration": ["x", "y"], "heading": ["°", "d°"]}}
PRED = {"VEHICLE": {"position": ["x", "y"]}}
HP = {"incl_robot_node": False, "map_cnn_encoding": False, "map_vit_encoding": False, "lane_cnn_encoding": False}


class _Data:
    """Stand-in for the reference's DoubleHeaderNumpyArray: ``data[:, state]`` -> the state's columns (all 8, in order)."""

    def __init__(self, arr):
        self.arr = arr

    def __getitem__(self, key):
        rows, state = key
        assert list(state.keys()) == list(STATE["VEHICLE"].keys())
        return self.arr[rows]


class _Node:          # hashable like the reference's Node (node.py:47-60)
    def __init__(self, i, first, table, lanes=None):
        self.type, self.id, self.first_timestep, self.is_robot, self.data = "VEHICLE", str(i), int(first), False, _Data(table)
        self.lanes_point = lanes

    def get(self, tr_scene, state):       # Node.get for the position state (node.py:101-117): inclusive range, NaN padding
        assert state == {"position": ["x", "y"]}
        t0, t1 = int(tr_scene[0]), int(tr_scene[1])
        out = np.full((t1 - t0 + 1, 2), np.nan)
        last = self.first_timestep + len(self.data.arr) - 1
        for j, t in enumerate(range(t0, t1 + 1)):
            if self.first_timestep <= t <= last:
                out[j] = self.data.arr[t - self.first_timestep, 0:2]
        return out


def _scene(tables, firsts, lanes=None):
    nodes = [_Node(i, f, t, None if lanes is None else lanes[i]) for i, (t, f) in enumerate(zip(tables, firsts))]
    return types.SimpleNamespace(nodes=nodes, map=None)


def _split(table, lengths):
    off = np.concatenate([[0], np.cumsum(lengths)])
    return [table[off[i]:off[i + 1]] for i in range(len(lengths))]


def test_get_timesteps_data_equals_reference_outputs():
    from multimodal_transformer_amd.model.dataset import get_timesteps_data, get_node_timestep_data
    z = np.load(os.path.join(GOLD, "batch.npz"))
    meta = json.load(open(os.path.join(GOLD, "batch.json")))
    scene = _scene(_split(z["table"], z["lengths"]), z["first"])
    for tag, m in meta.items():
        batch, nodes, out_ts = get_timesteps_data(env=None, scene=scene, t=z[f"{tag}::t"], node_type="VEHICLE", state=STATE,
                                                  pred_state=PRED, edge_types=None, min_ht=m["min_ht"], max_ht=m["max_ht"],
                                                  min_ft=m["min_ft"], max_ft=m["max_ft"], hyperparams=HP)
        assert [int(n.id) for n in nodes] == z[f"{tag}::nodes"].tolist() and out_ts == z[f"{tag}::timesteps"].tolist()
        fhi, x, y, x_st, y_st = batch[:5]
        assert all(b is None for b in batch[5:]) and x.is_cuda
        assert np.array_equal(fhi.cpu().numpy(), z[f"{tag}::first_history_index"])
        for got, key in ((x, "x"), (y, "y"), (x_st, "x_st"), (y_st, "y_st")):
            want = z[f"{tag}::{key}"]
            g = got.cpu().numpy()
            assert g.dtype == want.dtype and np.array_equal(g, want, equal_nan=True), (tag, key)
    assert get_timesteps_data(env=None, scene=scene, t=np.array([47]), node_type="VEHICLE", state=STATE, pred_state=PRED,
                              edge_types=None, min_ht=1, max_ht=8, min_ft=12, max_ft=12, hyperparams=HP) is None
    # a single element through the per-node entry point
    i = int(z["one_ts::nodes"][3])
    one = get_node_timestep_data(None, scene, 20, scene.nodes[i], STATE, PRED, None, 8, 12, HP)
    assert np.array_equal(one[3].cpu().numpy(), z["one_ts::x_st"][3], equal_nan=True)


def test_large_scene_equals_oracle():
    from multimodal_transformer_amd.model.dataset import get_timesteps_data
    from oracle import batch as ob
    rng = np.random.default_rng(5)
    N, T = 400, 120
    firsts = rng.integers(0, 90, size=N)
    lengths = np.array([rng.integers(1, T - f + 1) for f in firsts])
    tables = [rng.normal(size=(l, 8)) * 300.0 for l in lengths]
    scene = _scene(tables, firsts)
    ts = np.arange(0, T)
    batch, nodes, out_ts = get_timesteps_data(env=None, scene=scene, t=ts, node_type="VEHICLE", state=STATE, pred_state=PRED,
                                              edge_types=None, min_ht=2, max_ht=8, min_ft=12, max_ft=12, hyperparams=HP)
    table = np.concatenate(tables)
    row_off = np.concatenate([[0], np.cumsum(lengths)[:-1]])
    (fhi, x, y, x_st, y_st), on, ot = ob.timesteps_data(table, row_off, firsts, firsts + lengths - 1, ts, 2, 8, 12)
    assert [int(n.id) for n in nodes] == on.tolist() and out_ts == ot.tolist() and len(out_ts) > 2000
    assert np.array_equal(batch[0].cpu().numpy(), fhi)
    for got, want in zip(batch[1:5], (x, y, x_st, y_st)):
        assert np.array_equal(got.cpu().numpy(), want, equal_nan=True)


def test_predict_from_a_scene_through_trajectron():
    """The vectorised batch builder plugged into the Trajectron mirror: scene -> {timestep: {node: (1,1,ph,2)}}."""
    from helpers import TINY_TR
    from multimodal_transformer_amd.config import make_hyperparams
    from multimodal_transformer_amd.model import ModelRegistrar
    from multimodal_transformer_amd.model.trajectron import Trajectron
    from multimodal_transformer_amd.model.dataset import get_timesteps_data
    from multimodal_transformer_amd.weights import apply_recipe
    z = np.load(os.path.join(GOLD, "batch.npz"))
    scene = _scene(_split(z["table"], z["lengths"]), z["first"])
    dev = torch.device("cuda:0")
    hp = make_hyperparams("basic", history=8, horizon=12, transformer=TINY_TR)
    hp.update(HP)
    env = types.SimpleNamespace(NodeType=["VEHICLE"], get_edge_types=lambda: [], robot_type=None)
    reg = ModelRegistrar("/tmp", dev)
    tr = Trajectron(reg, hp, None, dev, get_timesteps_data=get_timesteps_data)
    tr.set_environment(env)
    apply_recipe(reg.model_dict, seed=3)
    hist, _ = tr.predict(scene, np.arange(10, 14), 12, min_future_timesteps=12, min_history_timesteps=1)
    assert sorted(hist.keys()) == [10, 11, 12, 13]
    n_total = sum(len(v) for v in hist.values())
    assert n_total > 0 and all(a.shape == (1, 1, 12, 2) and np.isfinite(a).all() for v in hist.values() for a in v.values())


def test_lane_batches_equal_reference_outputs():
    """Lane_encoding_CNN batches (lane points / 80, lane mask, nearest-lane labels of preprocessing.py:170-202) from
    mmt_lane_gather == what the reference's get_timesteps_data returned for the same scene (tests/golden/batch_lane.npz):
    bit-exact; and the batch drives Trajectron.predict end to end."""
    from multimodal_transformer_amd.model.dataset import get_timesteps_data, get_node_timestep_data
    z = np.load(os.path.join(GOLD, "batch.npz"))
    zl = np.load(os.path.join(GOLD, "batch_lane.npz"))
    meta = json.load(open(os.path.join(GOLD, "batch_lane.json")))
    scene = _scene(_split(z["table"], z["lengths"]), z["first"], _split(zl["lanes"], z["lengths"]))
    hp = dict(HP, lane_cnn_encoding=True)
    for tag, m in meta.items():
        batch, nodes, out_ts = get_timesteps_data(env=None, scene=scene, t=zl[f"{tag}::t"], node_type="VEHICLE", state=STATE,
                                                  pred_state=PRED, edge_types=None, min_ht=m["min_ht"], max_ht=m["max_ht"],
                                                  min_ft=m["max_ft"], max_ft=m["max_ft"], hyperparams=hp)
        assert [int(n.id) for n in nodes] == zl[f"{tag}::nodes"].tolist() and out_ts == zl[f"{tag}::timesteps"].tolist()
        lane_input, lane_label, lane_mask = batch[8]
        assert lane_input.is_cuda and lane_mask.dtype == torch.bool
        assert np.array_equal(batch[3].cpu().numpy(), zl[f"{tag}::x_st"], equal_nan=True)
        assert np.array_equal(lane_input.cpu().numpy(), zl[f"{tag}::lane_input"])
        assert np.array_equal(lane_label.cpu().numpy(), zl[f"{tag}::lane_label"])
        assert np.array_equal(lane_mask.cpu().numpy(), zl[f"{tag}::lane_t_mask"])
    i = int(zl["lane_range::nodes"][5])
    one = get_node_timestep_data(None, scene, int(zl["lane_range::timesteps"][5]), scene.nodes[i], STATE, PRED, None, 8, 12, hp)
    assert np.array_equal(one[8][1].cpu().numpy(), zl["lane_range::lane_label"][5])
    # end to end: lane batches from the scene through Trajectron.predict (three lane trajectories per node)
    from helpers import TINY_TR
    from multimodal_transformer_amd.config import make_hyperparams
    from multimodal_transformer_amd.model import ModelRegistrar
    from multimodal_transformer_amd.model.trajectron import Trajectron
    from multimodal_transformer_amd.weights import apply_recipe
    dev = torch.device("cuda:0")
    hpm = make_hyperparams("lane_cnn", history=8, horizon=12, transformer=TINY_TR)
    hpm.update(hp)
    env = types.SimpleNamespace(NodeType=["VEHICLE"], get_edge_types=lambda: [], robot_type=None)
    reg = ModelRegistrar("/tmp", dev)
    tr = Trajectron(reg, hpm, None, dev, get_timesteps_data=get_timesteps_data)
    tr.set_environment(env)
    apply_recipe(reg.model_dict, seed=3)
    hist, lane = tr.predict(scene, np.arange(10, 13), 12, min_future_timesteps=12, min_history_timesteps=1)
    assert sorted(lane.keys()) == [10, 11, 12]
    assert all(len(per_node) == 3 and all(np.isfinite(a).all() for a in per_node.values())
               for v in lane.values() for per_node in v.values())


def test_batch_statistics_equal_reference_outputs():
    """compute_batch_statistics / lane_compute_batch_statistics with the reference's signatures, reduced on the GPU == what the
    reference's own functions returned for the same prediction dicts (tests/golden/evalstats.npz): ADE / FDE within 1e-3 m
    (north star), same order; and Trajectron.evaluate (device-resident) == the dict route."""
    from multimodal_transformer_amd.evaluation import compute_batch_statistics, lane_compute_batch_statistics
    z = np.load(os.path.join(GOLD, "batch.npz"))
    ze = np.load(os.path.join(GOLD, "evalstats.npz"))
    scene = _scene(_split(z["table"], z["lengths"]), z["first"])
    pred, lane_pred = {}, {}
    for i, (n, t) in enumerate(zip(ze["nodes"], ze["timesteps"])):
        node = scene.nodes[int(n)]
        pred.setdefault(int(t), {})[node] = ze["pred"][i][None, None]
        lane_pred.setdefault(int(t), {})[node] = {k: ze["lane_pred"][i, k][None, None] for k in range(3)}
    one = compute_batch_statistics(pred, 0.5, max_hl=8, ph=12, node_type_enum=["VEHICLE"], map=None, prune_ph_to_future=False)
    lane = lane_compute_batch_statistics(lane_pred, 0.5, max_hl=8, ph=12, node_type_enum=["VEHICLE"], map=None,
                                         prune_ph_to_future=False)
    assert len(one["VEHICLE"]["ade"]) == len(ze["ade"]) and one["VEHICLE"]["kde"] == [0] * len(ze["ade"])
    for got, want in ((one["VEHICLE"]["ade"], ze["ade"]), (one["VEHICLE"]["fde"], ze["fde"]),
                      (lane["VEHICLE"]["ade"], ze["lane_ade"]), (lane["VEHICLE"]["fde"], ze["lane_fde"])):
        assert np.max(np.abs(np.array(got) - want)) <= 1e-4        # (bar: 1e-3 m)
    with pytest.raises(NotImplementedError):
        compute_batch_statistics(pred, 0.5, max_hl=8, ph=12, node_type_enum=["VEHICLE"], kde=True)
    # device-resident route == predict + dicts + statistics
    from helpers import TINY_TR
    from multimodal_transformer_amd.config import make_hyperparams
    from multimodal_transformer_amd.model import ModelRegistrar
    from multimodal_transformer_amd.model.trajectron import Trajectron
    from multimodal_transformer_amd.model.dataset import get_timesteps_data
    from multimodal_transformer_amd.weights import apply_recipe
    dev = torch.device("cuda:0")
    hp = make_hyperparams("basic", history=8, horizon=12, transformer=TINY_TR)
    hp.update(HP)
    env = types.SimpleNamespace(NodeType=["VEHICLE"], get_edge_types=lambda: [], robot_type=None)
    reg = ModelRegistrar("/tmp", dev)
    tr = Trajectron(reg, hp, None, dev, get_timesteps_data=get_timesteps_data)
    tr.set_environment(env)
    apply_recipe(reg.model_dict, seed=3)
    ts = np.arange(10, 14)
    hist, _ = tr.predict(scene, ts, 12, min_future_timesteps=12, min_history_timesteps=1)
    want = compute_batch_statistics(hist, 0.5, max_hl=8, ph=12, node_type_enum=["VEHICLE"])["VEHICLE"]
    got = tr.evaluate(scene, ts, 12, min_history_timesteps=1)["VEHICLE"]
    assert len(got["ade"]) == len(want["ade"]) > 0
    assert np.max(np.abs(got["ade"] - np.array(want["ade"]))) <= 1e-3 and np.max(np.abs(got["fde"] - np.array(want["fde"]))) <= 1e-3


def test_collate_of_per_node_elements_equals_the_vectorised_batch():
    """preprocessing.collate / restore (preprocessing.py:9-55) over get_node_timestep_data elements == get_timesteps_data for the
    same (node, timestep) pairs: state tensors, lane tuples, and the cropped map patches (4-tuple branch -> one mmt_map_crop launch)."""
    from multimodal_transformer_amd.environment import GeometricMap
    from multimodal_transformer_amd.model.dataset import collate, get_node_timestep_data, get_timesteps_data, restore
    z = np.load(os.path.join(GOLD, "batch.npz"))
    zl = np.load(os.path.join(GOLD, "batch_lane.npz"))
    zm = np.load(os.path.join(GOLD, "mapcrop.npz"))
    scene = _scene(_split(z["table"], z["lengths"]), z["first"], _split(zl["lanes"], z["lengths"]))
    scene.map = {"VEHICLE": GeometricMap(zm["raster0"], zm["homography0"])}
    ts = np.array([20, 21])
    map_hp = {"map_encoder": {"cnn_param": {"heading_state_index": 6, "patch_size": [50, 10, 50, 90]}}}
    for extra in ({}, {"lane_cnn_encoding": True}, dict(map_hp, map_cnn_encoding=True)):
        hp = dict(HP, **extra)
        batch, nodes, out_ts = get_timesteps_data(env=None, scene=scene, t=ts, node_type="VEHICLE", state=STATE, pred_state=PRED,
                                                  edge_types=None, min_ht=1, max_ht=8, min_ft=12, max_ft=12, hyperparams=hp)
        items = [get_node_timestep_data(None, scene, t, n, STATE, PRED, None, 8, 12, hp) for n, t in zip(nodes, out_ts)]
        got = collate(items)
        assert len(got) == 10 and all(g is None for g in got[5:8])
        for a, b in zip(got[:5], batch[:5]):
            assert torch.equal(a, b) or (torch.isnan(a) == torch.isnan(b)).all() and torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))
        if hp["lane_cnn_encoding"]:
            pts, label, mask = got[8]
            assert len(pts) == len(nodes) and torch.equal(torch.stack(list(pts)), batch[8][0])
            assert torch.equal(torch.stack(list(label)), batch[8][1]) and torch.equal(torch.stack(list(mask)), batch[8][2])
        else:
            assert got[8] is None
        if hp["map_cnn_encoding"]:
            assert got[9].shape == (len(nodes), 3, 100, 100) and torch.equal(got[9], batch[9])
        else:
            assert got[9] is None
    assert collate([]) == [] and collate([None, None]) is None
    d = collate([{"a": 1, "b": 2}, {"a": 3, "b": 4}])
    assert d == {"a": [1, 3], "b": [2, 4]} and restore(d) is d
    import dill
    assert restore(dill.dumps(d)) == d
