"""Batch-tuple unpacking and per-node-type dispatch (reference ``Transformer/model/trajectron.py:7-223``).

Only the thin caller of the hot path is mirrored.  Scene -> batch construction
(``model/dataset/preprocessing.py``) is CPU data preparation outside this build's scope
(SURVEY.md §8 f2); ``predict`` therefore takes an injectable ``get_timesteps_data`` callable with the
reference's signature, or use ``predict_batch`` on an already collated batch.
"""
import numpy as np
import torch

from .. import ops
from .mgcvae import MultimodalGenerativeCVAE


class Trajectron(object):
    def __init__(self, model_registrar, hyperparams, log_writer, device, get_timesteps_data=None):
        self.hyperparams = hyperparams
        self.log_writer = log_writer
        self.device = device
        self.curr_iter = 0
        self.model_registrar = model_registrar
        self.node_models_dict = dict()
        self.nodes = set()
        self.env = None
        self.get_timesteps_data = get_timesteps_data
        self.min_ht = hyperparams["minimum_history_length"]
        self.max_ht = hyperparams["maximum_history_length"]
        self.ph = hyperparams["prediction_horizon"]
        self.state = hyperparams["state"]
        self.state_length = {k: int(np.sum([len(v) for v in s.values()])) for k, s in self.state.items()}
        self.pred_state = hyperparams["pred_state"]

    def set_environment(self, env):
        self.env = env
        self.node_models_dict.clear()
        edge_types = env.get_edge_types() if hasattr(env, "get_edge_types") else []
        for node_type in env.NodeType:
            if node_type == "PEDESTRIAN":          # trajectron.py:42-43
                continue
            if node_type in self.pred_state.keys():
                self.node_models_dict[node_type] = MultimodalGenerativeCVAE(
                    env, node_type, self.model_registrar, self.hyperparams, self.device, edge_types,
                    log_writer=self.log_writer)

    def set_curr_iter(self, curr_iter):
        self.curr_iter = curr_iter

    def _unpack(self, batch):
        (_, x_t, y_t, x_st_t, y_st_t, _, _, _robot, lanes, map) = batch
        lane_input = None
        if self.hyperparams["lane_cnn_encoding"]:
            # reference batches: a tuple of (Ln,P,2) arrays (trajectron.py:62-63); the vectorised builder: one device tensor
            # ... or, from collate, a tuple of per-node device tensors
            if torch.is_tensor(lanes[0]):
                lane_input = lanes[0]
            elif len(lanes[0]) and torch.is_tensor(lanes[0][0]):
                lane_input = torch.stack(list(lanes[0]))
            else:
                lane_input = torch.as_tensor(np.array(lanes[0]), dtype=torch.float32)
            lane_input = lane_input.to(self.device)
        dev = lambda t: t.to(self.device) if isinstance(t, torch.Tensor) else None
        return dev(x_t), dev(y_t), dev(x_st_t), dev(y_st_t), lane_input, dev(map)

    def train_loss(self, batch, node_type):
        x, y, x_st, y_st, lanes, map = self._unpack(batch)
        return self.node_models_dict[node_type].train_loss(
            inputs=x, inputs_st=x_st, inputs_lane=lanes, lane_label=None, lane_t_mask=None, labels=y, labels_st=y_st,
            map=map, prediction_horizon=self.ph)

    def eval_loss(self, batch, node_type):
        x, y, x_st, y_st, lanes, map = self._unpack(batch)
        loss, reg = ops.checked(lambda: self.node_models_dict[node_type].eval_loss(
            node_type, inputs=x, inputs_st=x_st, inputs_lane=lanes, lane_label=None, lane_t_mask=None, labels=y,
            labels_st=y_st, map=map, prediction_horizon=self.ph))
        return loss.cpu().detach().numpy(), reg.cpu().detach().numpy()

    def predict_batch(self, batch, node_type, ph):
        x, _, x_st, _, lanes, map = self._unpack(batch)
        return self.node_models_dict[node_type].predict(
            inputs=x, inputs_st=x_st, inputs_lane=lanes, lane_t_mask=None, map=map, prediction_horizon=ph)

    def predict(self, scene, timesteps, ph, min_future_timesteps=0, min_history_timesteps=1):
        """trajectron.py:137-223: returns ({ts: {node: (1,1,ph,2)}}, {ts: {node: {lane: (1,1,ph,2)}}})."""
        if self.get_timesteps_data is None:
            raise NotImplementedError("scene batching is outside the hot path: construct Trajectron with "
                                      "get_timesteps_data=<reference preprocessing function> or call predict_batch")
        hist_dict, lane_dict = dict(), dict()
        for node_type in self.env.NodeType:
            if node_type not in self.pred_state or node_type not in self.node_models_dict:
                continue
            model = self.node_models_dict[node_type]
            batch = self.get_timesteps_data(
                env=self.env, scene=scene, t=timesteps, node_type=node_type, state=self.state,
                pred_state=self.pred_state, edge_types=model.edge_types, min_ht=min_history_timesteps,
                max_ht=self.max_ht, min_ft=min_future_timesteps, max_ft=min_future_timesteps,
                hyperparams=self.hyperparams)
            if batch is None:
                continue
            tensors, nodes, timesteps_o = batch
            # (the results are copied to the host below: the place to learn that a one-launch decode gave up, and to re-issue it)
            hist, lane = ops.checked(lambda: self.predict_batch(tensors, node_type, ph))
            if self.hyperparams["lane_cnn_encoding"]:
                arr = lane.cpu().numpy()                       # (B, L, ph, 2): one device->host copy
                for i, ts in enumerate(timesteps_o):
                    per_node = lane_dict.setdefault(ts, dict()).setdefault(nodes[i], dict())
                    for k in range(arr.shape[1]):
                        per_node[k] = arr[i, k][None, None]
            else:
                arr = hist.cpu().numpy()
                for i, ts in enumerate(timesteps_o):
                    hist_dict.setdefault(ts, dict())[nodes[i]] = arr[i][None, None]
        return hist_dict, lane_dict

    def evaluate(self, scene, timesteps, ph, min_history_timesteps=1):
        """The body of evaluate.py's scene loop (evaluate.py:99-148: predict with min_future_timesteps=ph, then
        compute_batch_statistics / lane_compute_batch_statistics) without leaving the device: batch construction, predict and
        the ADE / FDE reduction (best-of-lanes in lane mode) against the batch's own futures.  Returns
        ``{node_type: {'ade': (B,) array, 'fde': (B,) array, 'nodes': [...], 'timesteps': [...]}}`` in the reference's visiting order."""
        from ..evaluation import batch_ade_fde
        if self.get_timesteps_data is None:
            raise NotImplementedError("construct Trajectron with get_timesteps_data=<batch builder>")
        out = dict()
        for node_type in self.env.NodeType:
            if node_type not in self.pred_state or node_type not in self.node_models_dict:
                continue
            model = self.node_models_dict[node_type]
            batch = self.get_timesteps_data(
                env=self.env, scene=scene, t=timesteps, node_type=node_type, state=self.state, pred_state=self.pred_state,
                edge_types=model.edge_types, min_ht=min_history_timesteps, max_ht=self.max_ht, min_ft=ph, max_ft=ph,
                hyperparams=self.hyperparams)
            if batch is None:
                continue
            tensors, nodes, timesteps_o = batch
            hist, lane = ops.checked(lambda: self.predict_batch(tensors, node_type, ph))
            y = tensors[2].to(self.device)
            pred = lane if self.hyperparams["lane_cnn_encoding"] else hist
            # relative to the current position (column 0:2 of the last history row): keeps the fp32 reduction exact to ~1e-6 m
            now = tensors[1].to(self.device)[:, -1, 0:2]
            rel = pred - (now[:, None, None, :] if pred.dim() == 4 else now[:, None, :])
            ade, fde = batch_ade_fde(rel.contiguous(), (y - now[:, None, :]).contiguous())
            out[node_type] = {"ade": ade.cpu().numpy(), "fde": fde.cpu().numpy(), "nodes": nodes, "timesteps": timesteps_o}
        return out

