from .evaluation import (compute_ade, compute_fde, batch_ade_fde, compute_batch_statistics,  # noqa: F401
                         lane_compute_batch_statistics, log_batch_errors, print_batch_errors)
