"""ADE / FDE reductions (reference ``Transformer/evaluation/evaluation.py:10-19,64-163``).

``batch_ade_fde`` is the device path: one kernel over (B,ph,2) instead of the reference's per-node
numpy loop.  ``compute_batch_statistics`` / ``lane_compute_batch_statistics`` keep the reference's signatures
(evaluate.py:132-148 calls them with the dicts ``Trajectron.predict`` returns) and run that reduction once per
call on the GPU; ``Trajectron.evaluate`` skips the dicts altogether.  ``compute_ade`` / ``compute_fde`` keep the
reference's per-node numpy signatures for callers that already hold host arrays.
"""
import numpy as np
import torch

from .. import ops


def compute_ade(predicted_trajs, gt_traj):
    return np.mean(np.linalg.norm(predicted_trajs - gt_traj, axis=-1), axis=-1).flatten()


def compute_fde(predicted_trajs, gt_traj):
    return np.linalg.norm(predicted_trajs[:, :, -1] - gt_traj[-1], axis=-1).flatten()


def batch_ade_fde(pred_m: torch.Tensor, gt: torch.Tensor):
    """pred_m (B,ph,2) or (B,L,ph,2) metres on the GPU, gt (B,ph,2) -> (ade (B), fde (B)); lanes: best-of-L."""
    if pred_m.dim() == 4:
        res = [batch_ade_fde(pred_m[:, k].contiguous(), gt) for k in range(pred_m.shape[1])]
        return (torch.stack([r[0] for r in res], -1).min(-1)[0], torch.stack([r[1] for r in res], -1).min(-1)[0])
    zero = torch.zeros(pred_m.shape[0], 2, dtype=torch.float32, device=pred_m.device)
    # finalize() un-standardises with *80 + node_pos; feed metres/80 and node_pos = 0 would lose bits, so the
    # kernel is asked for the standardised-units metric on the raw tensors (metres flag off = no rescale).
    _, ade, fde, _ = ops.finalize(pred_m.contiguous(), zero, gt.contiguous(), metres=False, want_pred=False,
                                  want_metrics=True)
    return ade, fde


_POSITION = {"position": ["x", "y"]}


def _futures(node, t, ph):
    """utils/trajectory_utils.py:28-29: the node's positions at t+1..t+ph, rows outside its lifetime dropped."""
    future = np.asarray(node.get(np.array([t + 1, t + ph]), _POSITION))
    return future[~np.isnan(future.sum(axis=1))]


def _reduce_groups(entries, device, lanes):
    """entries: [(pred (ph,2) or (L,ph,2), future (F,2))] -> (ade, fde) lists in entry order; one device reduction per distinct
    future length (all of them ph in evaluate.py, which asks predict for min_future_timesteps=ph)."""
    ade, fde = [None] * len(entries), [None] * len(entries)
    groups = {}
    for i, (_, fut) in enumerate(entries):
        groups.setdefault(len(fut), []).append(i)
    for F, idx in groups.items():
        # positions relative to the node's first future point, subtracted in fp64 like the reference's `pred - future`
        # (hundreds of metres of map offset would otherwise cost the fp32 reduction 1e-4 m)
        off = [entries[i][1][0] if F else 0.0 for i in idx]
        pred = np.stack([np.asarray(entries[i][0], dtype=np.float64) - o for i, o in zip(idx, off)])
        if pred.shape[-2] != F:
            raise ValueError(f"prediction horizon {pred.shape[-2]} != {F} future positions: the reference's "
                             "`predicted_trajs - gt_traj` does not broadcast either (pass prune_ph_to_future=True)")
        gt = np.stack([entries[i][1] - o for i, o in zip(idx, off)]).astype(np.float32)
        a, f = batch_ade_fde(torch.from_numpy(np.ascontiguousarray(pred, dtype=np.float32)).to(device), torch.from_numpy(gt).to(device))
        a, f = a.cpu().numpy(), f.cpu().numpy()
        for k, i in enumerate(idx):
            ade[i], fde[i] = a[k], f[k]
    return ade, fde


def _statistics(prediction_output_dict, ph, node_type_enum, kde, obs, prune_ph_to_future, device, lanes):
    if kde or obs:
        raise NotImplementedError("KDE-NLL and obstacle-violation statistics are outside the hot path (evaluate.py never asks for "
                                  "them: evaluation.py:64-73 defaults)")
    batch_error_dict = {nt: {"ade": [], "fde": [], "kde": [], "obs_viols": []} for nt in node_type_enum}
    entries, owners = [], []
    for t in prediction_output_dict.keys():                       # evaluation.py:83-84 / :133-136: timestep order, node order
        for node, out in prediction_output_dict[t].items():
            future = _futures(node, t, ph)
            if lanes:
                pred = np.stack([np.asarray(out[k])[0, 0] for k in out.keys()])            # (L,ph,2); samples x batch = 1 x 1
            else:
                pred = np.asarray(out)[0, 0]
            if prune_ph_to_future:                                # trajectory_utils.py:31-34
                pred = pred[..., :future.shape[0], :]
                if pred.shape[-2] == 0:
                    continue
            entries.append((pred, future))
            owners.append(node.type)
    if entries:
        ade, fde = _reduce_groups(entries, device, lanes)
        for nt, a, f in zip(owners, ade, fde):
            d = batch_error_dict[nt]
            d["ade"].append(a), d["fde"].append(f), d["kde"].append(0), d["obs_viols"].append(0)
    return batch_error_dict


def compute_batch_statistics(prediction_output_dict, dt, max_hl, ph, node_type_enum, kde=False, obs=False, map=None,
                             prune_ph_to_future=False, best_of=False, device="cuda"):
    """evaluation.py:64-107 for ``{t: {node: (1,1,ph,2)}}`` (one sample per node, so best_of changes nothing): per node
    ADE = mean_t ||pred - future||, FDE = ||pred[-1] - future[-1]||, computed for all nodes in one device reduction."""
    return _statistics(prediction_output_dict, ph, node_type_enum, kde, obs, prune_ph_to_future, device, lanes=False)


def lane_compute_batch_statistics(prediction_output_dict, dt, max_hl, ph, node_type_enum, kde=False, obs=False, map=None,
                                  prune_ph_to_future=False, best_of=True, device="cuda"):
    """evaluation.py:110-163 for ``{t: {node: {lane: (1,1,ph,2)}}}``: the minimum over the node's lanes (best_of=True, the
    only value evaluate.py uses)."""
    if not best_of:
        raise NotImplementedError("lane statistics are defined as best-of-lanes (evaluation.py:153-156 default)")
    return _statistics(prediction_output_dict, ph, node_type_enum, kde, obs, prune_ph_to_future, device, lanes=True)


def _pooled(batch_errors_list):
    """{(node type label, metric): values of every batch, in order} for the dicts compute_batch_statistics returns."""
    pooled = {}
    for errors in batch_errors_list:
        for node_type, metrics in errors.items():
            label = getattr(node_type, "name", str(node_type))
            for metric, values in metrics.items():
                pooled.setdefault((label, metric), []).extend(values)
    return {k: v for k, v in pooled.items() if len(v) > 0}


def print_batch_errors(batch_errors_list, namespace, curr_iter):
    """evaluation.py:216-230: mean and median of every metric over all batches, one line each."""
    for (label, metric), values in _pooled(batch_errors_list).items():
        print(f"{curr_iter}: {label}/{namespace}/{metric}_mean", np.mean(values))
        print(f"{curr_iter}: {label}/{namespace}/{metric}_median", np.median(values))


def log_batch_errors(batch_errors_list, log_writer, namespace, curr_iter, bar_plot=[], box_plot=[]):
    """evaluation.py:166-213: histogram + mean + median per metric into a SummaryWriter-like object.  The matplotlib bar / box
    plots of the reference are visualisation, outside this build (asking for them raises)."""
    if bar_plot or box_plot:
        raise NotImplementedError("bar / box plots are visualisation code outside the hot path")
    for (label, metric), values in _pooled(batch_errors_list).items():
        log_writer.add_histogram(f"{label}/{namespace}/{metric}", values, curr_iter)
        log_writer.add_scalar(f"{label}/{namespace}/{metric}_mean", np.mean(values), curr_iter)
        log_writer.add_scalar(f"{label}/{namespace}/{metric}_median", np.median(values), curr_iter)
