"""Drop-in module classes for the reference's ``nn_component`` package (same names, constructor
signatures and state_dict keys; reference ``Transformer/nn_component/__init__.py:1-4``).

The classes own parameters only.  Their ``forward`` methods launch the hand-written gfx950
kernels through ``multimodal_transformer_amd.ops``; nothing is ever computed by ATen, and a CPU
tensor or a missing libmmt_amd.so raises ``MmtError``.
"""
from .modules import (CNNMapEncoder, Decoder, Encoder, Lane_Decoder, Lane_Encoder, Mlp, PositionalEncoding,  # noqa: F401
                      Trajectory_Decoder, ViT)

__all__ = ["CNNMapEncoder", "Decoder", "Encoder", "Lane_Decoder", "Lane_Encoder", "Mlp", "PositionalEncoding",
           "Trajectory_Decoder", "ViT"]
