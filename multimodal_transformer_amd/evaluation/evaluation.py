"""ADE / FDE reductions (reference ``Transformer/evaluation/evaluation.py:10-19,153-156``).

``batch_ade_fde`` is the device path: one kernel over (B,ph,2) instead of the reference's per-node
numpy loop.  ``compute_ade`` / ``compute_fde`` keep the reference's per-node numpy signatures for
callers that already hold host arrays (they are host utilities, not the hot path).
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
