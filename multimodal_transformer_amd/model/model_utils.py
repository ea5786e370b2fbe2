"""Small helpers of reference ``Transformer/model/model_utils.py`` that the hot path touches.

The masks are rebuilt inside the kernels (an index compare / a NaN test), so nothing in this
package calls the two mask helpers; they are kept, with the reference's semantics, for callers
that still build masks to pass along (``Encoder.forward`` accepts and ignores them).
"""
from enum import Enum

import torch


class ModeKeys(Enum):          # model_utils.py:11-14
    TRAIN = 1
    EVAL = 2
    PREDICT = 3


def generate_square_subsequent_mask(sz, device):
    """0 on and below the diagonal, -inf above (model_utils.py:141-145)."""
    return torch.triu(torch.full((sz, sz), float("-inf"), device=device), diagonal=1)


def generate_mask(src):
    """Key-padding mask: NaN in feature column 1 (model_utils.py:148-151)."""
    return torch.isnan(src)[:, :, 1]


def L2_norm(y_pred, labels):
    """model_utils.py:154-160: ``torch.norm(y_pred - labels, p=2, dim=-1)`` on the device (``mmt_l2_norm``); ``labels`` may carry one
    broadcast axis of length 1, as in the lane loss (mgcvae.py:603-606).  The losses themselves use the fused ``mmt_finalize``."""
    from .. import _train_kernels as K
    return K.l2_norm(y_pred.contiguous(), labels.contiguous())
