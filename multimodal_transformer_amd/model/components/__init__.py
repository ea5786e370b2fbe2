from .gmm2d import GMM2D  # noqa: F401
