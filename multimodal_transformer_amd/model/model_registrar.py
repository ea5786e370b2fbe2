"""String-keyed module registry + checkpoint I/O (reference ``Transformer/model/model_registrar.py:11-80``).

The registry is the reference's plugin interface: whoever stores a module under a key owns that
sub-computation.  Checkpoint layout is kept: ``<model_dir>/model_registrar-<iter>.pt`` holds the
pickled ``nn.ModuleDict``.  ``load_models`` can also read checkpoints written by the reference
itself (classes pickled under ``nn_component.*`` / ``vit_pytorch.*``): ``install_reference_aliases``
maps those import paths onto this package's classes, whose forwards read weights by state_dict name
and therefore work on the reference's attribute layout.
"""
import os
import sys
import types

import torch
import torch.nn as nn


def get_model_device(model):
    return next(model.parameters()).device


def install_reference_aliases():
    """Make ``nn_component.*`` / ``vit_pytorch`` resolve to this package so reference pickles load."""
    from .. import nn_component as ours
    from ..nn_component import modules as m
    if "nn_component" in sys.modules and getattr(sys.modules["nn_component"], "__mmt_alias__", False):
        return
    pkg = types.ModuleType("nn_component")
    pkg.__path__ = []
    pkg.__mmt_alias__ = True
    layout = {
        "Transformer_model": ["Encoder", "Decoder", "Lane_Decoder", "Trajectory_Decoder", "PositionalEncoding"],
        "MLP": ["Mlp"], "map_encoder": ["CNNMapEncoder"], "Lane_encoder": ["Lane_Encoder"],
    }
    for sub, names in layout.items():
        mod = types.ModuleType(f"nn_component.{sub}")
        for n in names:
            setattr(mod, n, getattr(m, n))
            setattr(pkg, n, getattr(m, n))
        setattr(pkg, sub, mod)
        sys.modules.setdefault(f"nn_component.{sub}", mod)
    sys.modules.setdefault("nn_component", pkg)
    if "vit_pytorch" not in sys.modules:
        vp = types.ModuleType("vit_pytorch")
        vp.__path__ = []
        vv = types.ModuleType("vit_pytorch.vit")
        vp.ViT = vv.ViT = ours.ViT
        vp.vit = vv
        sys.modules["vit_pytorch"] = vp
        sys.modules["vit_pytorch.vit"] = vv


class ModelRegistrar(nn.Module):
    def __init__(self, model_dir, device):
        super().__init__()
        self.model_dict = nn.ModuleDict()
        self.model_dir = model_dir
        self.device = device

    def forward(self):
        raise NotImplementedError("ModelRegistrar only stores modules")

    def get_model(self, name, model_if_absent=None):
        if name in self.model_dict:
            return self.model_dict[name]
        if model_if_absent is None:
            raise ValueError(f"{name} was never initialized in this Registrar!")
        self.model_dict[name] = model_if_absent.to(self.device)
        return self.model_dict[name]

    def get_name_match(self, name):
        return nn.ModuleList([m for k, m in self.model_dict.items() if name in k])

    def get_all_but_name_match(self, name):
        return nn.ModuleList([m for k, m in self.model_dict.items() if name not in k])

    def print_model_names(self):
        print(self.model_dict.keys())

    def _path(self, it):
        return os.path.join(self.model_dir, "model_registrar-%d.pt" % it)

    def save_models(self, curr_iter):
        torch.save(self.model_dict, self._path(curr_iter))

    def load_models(self, iter_num):
        """Reads this package's own checkpoints and the reference's (pickled module objects; the file is
        trusted input exactly as in the reference, which calls torch.load on it unguarded)."""
        install_reference_aliases()
        self.model_dict.clear()
        self.model_dict = torch.load(self._path(iter_num), map_location=self.device, weights_only=False)

    def to(self, device):
        for _, model in self.model_dict.items():
            if get_model_device(model) != torch.device(device):
                model.to(device)
        self.device = device
        return self
