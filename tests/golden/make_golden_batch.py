"""Golden vectors for batch construction, produced by the REAL reference:
``model/dataset/preprocessing.py`` ``get_timesteps_data`` / ``get_node_timestep_data`` on a synthetic scene built from the
reference's own ``environment`` classes (Environment, Scene, Node, DoubleHeaderNumpyArray).

Run in THIS container only (``/root/reference`` does not travel):  python tests/golden/make_golden_batch.py

Shims, all for things the image lacks or that moved in Python 3.10 (none of them takes part in the arithmetic):
  * ``collections.Sequence`` alias (environment/data_structures.py:3 imports it from its pre-3.10 place);
  * ``orjson`` (environment.py:1; only ``dumps`` as a memo key) and ``ncls`` (node.py:5; only MultiNode uses it);
  * ``vit_pytorch`` as in make_golden.py (model/__init__ imports mgcvae).
Nothing of the reference is copied: the file holds the synthetic node tables and the tensors the reference returned.
"""
import collections
import collections.abc
import json
import os
import sys
import types

import numpy as np
import pandas as pd
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/Transformer"

collections.Sequence = collections.abc.Sequence
orjson = types.ModuleType("orjson")
orjson.dumps = lambda o: json.dumps(o, sort_keys=True, default=str).encode()
sys.modules["orjson"] = orjson
ncls = types.ModuleType("ncls")
ncls.NCLS = object
sys.modules["ncls"] = ncls
vit = types.ModuleType("vit_pytorch")
vit.ViT = type("ViT", (torch.nn.Module,), {})
sys.modules["vit_pytorch"] = vit
sys.path.insert(0, REF)

from environment import Environment, Scene, Node  # noqa: E402  (reference)
from model.dataset.preprocessing import get_timesteps_data  # noqa: E402  (reference)

STATE = {"VEHICLE": {"position": ["x", "y"], "velocity": ["x", "y"], "acceleration": ["x", "y"], "heading": ["°", "d°"]}}
PRED = {"VEHICLE": {"position": ["x", "y"]}}
COLS = pd.MultiIndex.from_tuples([(e, d) for e, dims in STATE["VEHICLE"].items() for d in dims])
HP = {"incl_robot_node": False, "map_cnn_encoding": False, "map_vit_encoding": False, "lane_cnn_encoding": False}


def main():
    rng = np.random.default_rng(11)
    std = {"VEHICLE": {e: {d: {"mean": 0.0, "std": 80.0 if e == "position" else 15.0} for d in dims}
                       for e, dims in STATE["VEHICLE"].items()}}
    env = Environment(node_type_list=["VEHICLE"], standardization=std)
    env.attention_radius = {(env.NodeType.VEHICLE, env.NodeType.VEHICLE): 30.0}
    T_scene, N = 48, 37
    scene = Scene(timesteps=T_scene, dt=0.5, name="synthetic")
    tables, firsts = [], []
    for n in range(N):
        first = int(rng.integers(0, 30))
        length = int(rng.integers(2, T_scene - first + 1))
        data = rng.normal(size=(length, 8)) * np.array([200., 200., 8., 8., 2., 2., 3., 0.5]) + np.array([500., 900., 0, 0, 0, 0, 0, 0])
        node = Node(node_type=env.NodeType.VEHICLE, node_id=str(n), data=pd.DataFrame(data, columns=COLS), first_timestep=first)
        scene.nodes.append(node)
        tables.append(data)
        firsts.append(first)
    out = {"first": np.array(firsts, dtype=np.int32), "lengths": np.array([len(t) for t in tables], dtype=np.int32),
           "table": np.concatenate(tables, axis=0)}
    cases = {"one_ts": (np.array([20]), 1, 8, 12, 12), "range": (np.arange(6, 40), 1, 8, 12, 12),
             "short_hist": (np.arange(0, 47), 0, 4, 3, 3), "eval_like": (np.arange(2, 30), 7, 8, 12, 12)}
    meta = {}
    for tag, (ts, min_ht, max_ht, min_ft, max_ft) in cases.items():
        res = get_timesteps_data(env=env, scene=scene, t=ts, node_type=env.NodeType.VEHICLE, state=STATE, pred_state=PRED,
                                 edge_types=env.get_edge_types(), min_ht=min_ht, max_ht=max_ht, min_ft=min_ft, max_ft=max_ft,
                                 hyperparams=HP)
        batch, nodes, out_ts = res
        fhi, x, y, x_st, y_st = batch[:5]
        assert all(b is None for b in batch[5:])
        out[f"{tag}::t"] = ts
        out[f"{tag}::first_history_index"] = fhi.numpy()
        for k, v in (("x", x), ("y", y), ("x_st", x_st), ("y_st", y_st)):
            out[f"{tag}::{k}"] = v.numpy()
        out[f"{tag}::nodes"] = np.array([int(nd.id) for nd in nodes], dtype=np.int32)
        out[f"{tag}::timesteps"] = np.array(out_ts, dtype=np.int32)
        meta[tag] = dict(min_ht=min_ht, max_ht=max_ht, min_ft=min_ft, max_ft=max_ft, B=len(nodes))
    assert get_timesteps_data(env=env, scene=scene, t=np.array([47]), node_type=env.NodeType.VEHICLE, state=STATE,
                              pred_state=PRED, edge_types=env.get_edge_types(), min_ht=1, max_ht=8, min_ft=12, max_ft=12,
                              hyperparams=HP) is None                        # nothing present -> None (preprocessing.py:268-269)
    np.savez_compressed(os.path.join(HERE, "batch.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "batch.json"), "w"), indent=1)
    print(meta)
    lane_golden(env, scene, tables, firsts, rng)
    stats_golden(env, scene, rng)


def _reference_evaluation():
    """Transformer/evaluation/evaluation.py, loaded from its file.  Its module-level imports pull in seaborn (visualization) and
    the nuScenes devkit (utils.lane_utils), which the image lacks and which compute_batch_statistics never touches: ``utils`` is
    served by the reference's own utils/trajectory_utils.py (also loaded from its file), ``visualization`` by an empty stub."""
    import importlib.util
    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    sys.modules["utils"] = load("utils", os.path.join(REF, "utils", "trajectory_utils.py"))          # reference
    sys.modules["visualization"] = types.ModuleType("visualization")
    return load("ref_evaluation", os.path.join(REF, "evaluation", "evaluation.py"))                  # reference


def stats_golden(env, scene, rng):
    """compute_batch_statistics / lane_compute_batch_statistics (evaluation.py:64-163) on synthetic prediction dicts
    ({t: {node: (1,1,ph,2)}} and {t: {node: {lane: (1,1,ph,2)}}}, trajectron.py:216-221) over the same scene."""
    ev = _reference_evaluation()
    ph, max_hl = 12, 8
    vt = env.NodeType.VEHICLE
    pos = {"position": ["x", "y"]}
    pred, lane_pred = {}, {}
    node_ids, tss, arrs, lane_arrs = [], [], [], []
    for t in (8, 15, 20, 31):
        for node in scene.present_nodes(np.array([t]), type=vt, min_history_timesteps=1, min_future_timesteps=ph).get(t, []):
            fut = node.get(np.array([t + 1, t + ph]), pos)
            a = (fut + rng.normal(size=fut.shape) * 2.0).astype(np.float32)[None, None]
            pred.setdefault(t, {})[node] = a
            la = [(fut + rng.normal(size=fut.shape) * 3.0).astype(np.float32)[None, None] for _ in range(3)]
            lane_pred.setdefault(t, {})[node] = {k: la[k] for k in range(3)}
            node_ids.append(int(node.id)), tss.append(t), arrs.append(a[0, 0]), lane_arrs.append(np.stack([x[0, 0] for x in la]))
    one = ev.compute_batch_statistics(pred, scene.dt, max_hl=max_hl, ph=ph, node_type_enum=env.NodeType, map=None,
                                      prune_ph_to_future=False)
    lane = ev.lane_compute_batch_statistics(lane_pred, scene.dt, max_hl=max_hl, ph=ph, node_type_enum=env.NodeType, map=None,
                                            prune_ph_to_future=False)
    out = {"nodes": np.array(node_ids, dtype=np.int32), "timesteps": np.array(tss, dtype=np.int32), "pred": np.stack(arrs),
           "lane_pred": np.stack(lane_arrs), "ade": np.array(one[vt]["ade"]), "fde": np.array(one[vt]["fde"]),
           "lane_ade": np.array(lane[vt]["ade"]), "lane_fde": np.array(lane[vt]["fde"]),
           "kde": np.array(one[vt]["kde"], dtype=np.float64), "obs_viols": np.array(one[vt]["obs_viols"], dtype=np.float64)}
    np.savez_compressed(os.path.join(HERE, "evalstats.npz"), **out)
    print("evalstats", len(node_ids), float(out["ade"].mean()), float(out["lane_ade"].mean()))


def lane_golden(env, scene, tables, firsts, rng):
    """Lane batches (preprocessing.py:170-202 + collate :43-45): every node gets a lane table (timesteps x 3 lanes x 10 points
    x 2, metres, as Node.get_lane_id stores it); absent lanes are the constant -100 m point (lane_utils.py:279-282)."""
    lanes = []
    for node, data in zip(scene.nodes, tables):
        n = len(data)
        lp = np.empty((n, 3, 10, 2))
        for i in range(n):
            for l in range(3):
                if rng.random() < 0.25:
                    lp[i, l] = -100.0                                    # absent lane
                else:
                    d = rng.normal(size=2)
                    start = data[i, :2] + rng.normal(size=2) * 6.0
                    lp[i, l] = start + np.outer(np.linspace(0.1, 1.0, 10), d / np.linalg.norm(d)) * rng.uniform(3, 30)
        node.get_lane_id(lp, [["a", "b", "c"]] * n)
        lanes.append(lp)
    out = {"lanes": np.concatenate(lanes, axis=0)}
    hp = dict(HP, lane_cnn_encoding=True)
    meta = {}
    for tag, (ts, min_ht, max_ht, max_ft) in {"lane_range": (np.arange(6, 40), 1, 8, 12), "lane_short": (np.arange(0, 47), 0, 4, 3)}.items():
        batch, nodes, out_ts = get_timesteps_data(env=env, scene=scene, t=ts, node_type=env.NodeType.VEHICLE, state=STATE,
                                                  pred_state=PRED, edge_types=env.get_edge_types(), min_ht=min_ht, max_ht=max_ht,
                                                  min_ft=max_ft, max_ft=max_ft, hyperparams=hp)
        lane_pts, y_lane, lane_mask = batch[8]
        out[f"{tag}::t"] = ts
        out[f"{tag}::lane_input"] = torch.tensor(np.array(lane_pts), dtype=torch.float32).numpy()     # trajectron.py:62-63
        out[f"{tag}::lane_label"] = torch.stack(y_lane).numpy()                                        # trajectron.py:64
        out[f"{tag}::lane_t_mask"] = torch.stack(lane_mask).numpy()                                    # trajectron.py:65
        out[f"{tag}::x_st"] = batch[3].numpy()
        out[f"{tag}::nodes"] = np.array([int(nd.id) for nd in nodes], dtype=np.int32)
        out[f"{tag}::timesteps"] = np.array(out_ts, dtype=np.int32)
        meta[tag] = dict(min_ht=min_ht, max_ht=max_ht, max_ft=max_ft, B=len(nodes))
    np.savez_compressed(os.path.join(HERE, "batch_lane.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "batch_lane.json"), "w"), indent=1)
    print(meta)


if __name__ == "__main__":
    main()
