from .mgcvae import MultimodalGenerativeCVAE  # noqa: F401
from .model_registrar import ModelRegistrar  # noqa: F401
from .model_utils import ModeKeys  # noqa: F401
from .trajectron import Trajectron  # noqa: F401
