"""ORACLE (test infrastructure, never shipped in the product path).

CPU restatement, in plain numpy loops, of the reference's batch construction:
  * ``present_nodes``      -- environment/scene.py:103-140 (which nodes are in the scene at which query timestep, and the
                              order the reference visits them in: dict insertion order of the timesteps, scene order of nodes);
  * ``node_timestep_data`` -- model/dataset/preprocessing.py:79-123 with environment/node.py:77-117 (``Node.get``: NaN
                              padding outside the node's lifetime) and environment/environment.py:49-61 (``standardize``
                              subtracts the mean and divides by 80, whatever std is passed);
  * ``timesteps_data``     -- preprocessing.py:208-270 (+ ``collate``: stack).
  * ``lane_tuple``         -- preprocessing.py:170-202 (lane points / 80, lane mask, one-hot label of the nearest lane for
                              every future position) with collate :43-45 and trajectron.py:62-65 (stacking).
  * ``ade_fde`` / ``batch_statistics`` -- evaluation/evaluation.py:10-19 (compute_ade / compute_fde) and :64-163 (the per-node
                              loops of compute_batch_statistics / lane_compute_batch_statistics, best-of-lanes).
Pinned against the reference's own outputs in tests/golden/batch.npz / batch_lane.npz / evalstats.npz
(tests/test_oracle_golden.py).
"""
import numpy as np

SCALE = 80.0


def present_nodes(first, last, timesteps, min_ht, min_ft):
    """-> list of (timestep, [node indices]) in the reference's visiting order (scene.py:120-140)."""
    present = {}
    for n in range(len(first)):                                  # scene.py:122
        lower, upper = timesteps - min_ht, timesteps + min_ft    # scene.py:126-127
        mask = (first[n] <= lower) & (upper <= last[n])          # scene.py:128-129
        for j in np.nonzero(mask)[0]:
            present.setdefault(int(timesteps[j]), []).append(n)
    return list(present.items())


def node_get(rows, first, last, t0, t1, cols):
    """Node.get(np.array([t0, t1]), state): inclusive range, NaN outside [first, last] (node.py:101-117)."""
    length = t1 - t0 + 1
    out = np.full((length, len(cols)), np.nan)
    for j in range(length):
        tt = t0 + j
        if first <= tt <= last:
            out[j] = rows[tt - first][cols]
    return out


def node_timestep_data(rows, first, t, max_ht, max_ft, pred_cols):
    """One batch element (preprocessing.py:100-123); rows = the node's state table (fp64), position in columns 0, 1."""
    last = first + len(rows) - 1
    ncol = rows.shape[1]
    x = node_get(rows, first, last, t - max_ht, t, list(range(ncol)))          # :100,103
    y = node_get(rows, first, last, t + 1, t + max_ft, list(pred_cols))        # :101,104
    fhi = max(max_ht - (t - first), 0)                                         # :105
    rel = np.zeros(ncol)
    rel[0:2] = x[-1, 0:2]                                                      # :109-110
    x_st = np.where(np.isnan(x), np.nan, (x - rel) / SCALE)                    # :111-112 -> environment.py:61
    y_st = np.where(np.isnan(y), np.nan, (y - rel[0:2][:len(pred_cols)]) / SCALE)   # :114-116
    return fhi, x.astype(np.float32), y.astype(np.float32), x_st.astype(np.float32), y_st.astype(np.float32)


def timesteps_data(table, row_off, first, last, timesteps, min_ht, max_ht, max_ft, pred_cols=(0, 1)):
    """get_timesteps_data (preprocessing.py:236-270): presence needs min_ht history and max_ft future steps (:239-240)."""
    nodes, out_ts, elems = [], [], []
    for ts, present in present_nodes(first, last, np.asarray(timesteps), min_ht, max_ft):
        for n in present:
            rows = table[row_off[n]:row_off[n] + (last[n] - first[n] + 1)]
            elems.append(node_timestep_data(rows, int(first[n]), ts, max_ht, max_ft, pred_cols))
            nodes.append(n)
            out_ts.append(ts)
    if not out_ts:
        return None
    fhi = np.array([e[0] for e in elems], dtype=np.int64)
    return (fhi,) + tuple(np.stack([e[k] for e in elems]) for k in range(1, 5)), np.array(nodes), np.array(out_ts)


def lane_tuple(rows, lane_rows, first, t, max_ft):
    """(lane_input (Ln,P,2) f32, lane_label (max_ft,Ln) f32, lane_t_mask (Ln) bool) of one batch element.
    rows = the node's state table (fp64, position first), lane_rows = its lane table (timesteps, Ln, P, 2) in metres."""
    last = first + len(rows) - 1
    y = node_get(rows, first, last, t + 1, t + max_ft, [0, 1])
    y_st = np.where(np.isnan(y), np.nan, (y - rows[t - first][0:2]) / SCALE)           # fp64, :114-116
    pts = np.array(lane_rows[t - first]) / 80                                          # :190-192 (history_points_at = t - first)
    mask = np.linalg.norm(pts[:, 0] - pts[:, 1], axis=1) != 0                          # :193-194 (dtype=torch.bool)
    label = np.zeros((max_ft, pts.shape[0]), dtype=np.float32)
    for j, point in enumerate(y_st):                                                   # :196-200
        dist = []
        for i, lp in enumerate(pts):                                                   # get_gt_lane, :173-188
            if not mask[i]:
                dist.append(1e10)
                continue
            dist.append(np.abs(np.cross(lp[1] - lp[0], point - lp[0])) / np.abs(np.linalg.norm(lp[1] - lp[0])))
        label[j, int(np.argmin(dist))] = 1.0
    return pts.astype(np.float32), label, mask


def ade_fde(pred, future):
    """compute_ade / compute_fde (evaluation.py:10-19) of one (ph,2) prediction against the (ph,2) future."""
    err = np.linalg.norm(pred.astype(np.float64) - future, axis=-1)
    return float(np.mean(err)), float(np.linalg.norm(pred[-1].astype(np.float64) - future[-1]))


def batch_statistics(preds, futures):
    """preds: per node (ph,2) or (L,ph,2) (lanes: the minimum over L, evaluation.py:153-156); -> (ade, fde) arrays."""
    ade, fde = [], []
    for p, f in zip(preds, futures):
        per = [ade_fde(q, f) for q in (p if p.ndim == 3 else p[None])]
        ade.append(min(a for a, _ in per))
        fde.append(min(b for _, b in per))
    return np.array(ade), np.array(fde)
