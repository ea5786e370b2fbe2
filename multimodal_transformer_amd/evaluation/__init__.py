from .evaluation import compute_ade, compute_fde, batch_ade_fde  # noqa: F401
