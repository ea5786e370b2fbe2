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


def _allowed_checkpoint_globals():
    """What a registrar checkpoint may reference besides tensors: torch.nn module classes, this package's module classes (also
    under the reference's import paths, see install_reference_aliases) and the activation functions nn.Transformer* layers hold."""
    import collections
    import inspect

    import torch.nn.functional as F

    from ..nn_component import modules as m
    allowed = {"collections.OrderedDict": collections.OrderedDict, "builtins.set": set,
               "torch.nn.functional.relu": F.relu, "torch.nn.functional.gelu": F.gelu,
               "torch.nn.functional.leaky_relu": F.leaky_relu}
    for name in dir(torch.nn.modules):
        sub = getattr(torch.nn.modules, name)
        if inspect.ismodule(sub):
            for cn, cls in vars(sub).items():
                if inspect.isclass(cls) and issubclass(cls, nn.Module) and cls.__module__ == sub.__name__:
                    allowed[f"{sub.__name__}.{cn}"] = cls
    ours = [n for n, c in vars(m).items() if inspect.isclass(c) and issubclass(c, nn.Module) and c.__module__ == m.__name__]
    layout = {"Transformer_model": ["Encoder", "Decoder", "Lane_Decoder", "Trajectory_Decoder", "PositionalEncoding"],
              "MLP": ["Mlp"], "map_encoder": ["CNNMapEncoder"], "Lane_encoder": ["Lane_Encoder"]}
    for n in ours:
        allowed[f"{m.__name__}.{n}"] = getattr(m, n)
    for sub, names in layout.items():
        for n in names:
            allowed[f"nn_component.{sub}.{n}"] = getattr(m, n)
    allowed["vit_pytorch.vit.ViT"] = allowed["vit_pytorch.ViT"] = m.ViT
    return allowed


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

    def load_models(self, iter_num, trusted=False):
        """Reads this package's own checkpoints and the reference's (``torch.save`` of the ``nn.ModuleDict``: pickled module
        objects, model_registrar.py:58-75).  The file is read with torch's restricted unpickler (``weights_only=True``) with
        exactly the module classes such a checkpoint is made of allow-listed; a file that references anything else is refused
        unless ``trusted=True`` (then, and only then, it is unpickled in full, like the reference does unguarded)."""
        install_reference_aliases()
        path = self._path(iter_num)
        allowed = _allowed_checkpoint_globals()
        from .._lib import MmtError
        try:
            needed = torch.serialization.get_unsafe_globals_in_checkpoint(path)
        except Exception as e:      # e.g. a legacy (non-zip) torch.save file: the scan itself cannot read it
            if not trusted:
                raise MmtError(f"{path}: cannot list the objects this checkpoint references ({type(e).__name__}: {e}); "
                               "pass trusted=True to unpickle it anyway (executes code from the file)") from e
            needed = None
        extra = sorted(n for n in needed if n not in allowed) if needed is not None else ["<unscanned>"]
        self.model_dict.clear()
        if not extra:
            with torch.serialization.safe_globals([(allowed[n], n) for n in needed]):
                self.model_dict = torch.load(path, map_location=self.device, weights_only=True)
        elif trusted:
            self.model_dict = torch.load(path, map_location=self.device, weights_only=False)
        else:
            raise MmtError(f"{path} references objects a registrar checkpoint has no business holding: {extra}; "
                           "pass trusted=True to unpickle it anyway (executes code from the file)")

    def to(self, device):
        for _, model in self.model_dict.items():
            if get_model_device(model) != torch.device(device):
                model.to(device)
        self.device = device
        return self
