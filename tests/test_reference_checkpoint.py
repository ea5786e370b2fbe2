"""ModelRegistrar.load_models on checkpoints written by the REFERENCE's ModelRegistrar.save_models
(Transformer/model/model_registrar.py:58-75; fixtures from tests/golden/make_golden_ckpt.py): the pickled reference modules
come back as this package's classes, every state_dict entry equal to the by-name recipe the reference modules were filled with,
through the restricted unpickler; a file that references anything else is refused."""
import os

import pytest
import torch

from conftest import GOLDEN
from multimodal_transformer_amd.model import ModelRegistrar
from multimodal_transformer_amd.weights import recipe_tensor


@pytest.mark.parametrize("mode", ["lane_cnn", "map_cnn"])
def test_reference_written_checkpoint_loads(mode):
    from multimodal_transformer_amd import nn_component as ours
    if not os.path.isdir(os.path.join(GOLDEN, "ref_ckpt_" + mode)):
        # .gpurunignore keeps the pickled reference ModuleDicts off the GPU box: nothing to load there
        pytest.skip("reference-written checkpoint fixtures are not shipped to this machine")
    reg = ModelRegistrar(os.path.join(GOLDEN, "ref_ckpt_" + mode), "cpu")
    reg.load_models(7)
    keys = set(reg.model_dict.keys())
    assert {"VEHICLE/node_history_encoder", "VEHICLE/decoder/transformer_decoder", "VEHICLE/decoder/MLP"} <= keys
    assert isinstance(reg.model_dict["VEHICLE/node_history_encoder"], ours.Encoder)
    assert isinstance(reg.model_dict["VEHICLE/decoder/transformer_decoder"], ours.Trajectory_Decoder)
    if mode == "lane_cnn":
        assert isinstance(reg.model_dict["VEHICLE/lane_encoder"], ours.Lane_Encoder)
        assert isinstance(reg.model_dict["VEHICLE/Lane/MLP_Softmax"], ours.Mlp)
    else:
        assert isinstance(reg.model_dict["VEHICLE/map_encoder"], ours.CNNMapEncoder)
    n = 0
    for name, mod in reg.model_dict.items():
        for pn, t in mod.state_dict().items():
            if pn.endswith("pos_encoder.pe"):
                assert t.shape[-1] == 32 and torch.isfinite(t).all()
                continue
            assert torch.equal(t, recipe_tensor(f"{name}.{pn}", t.shape, 0)), (name, pn)
            n += 1
    assert n > 40
    # same keys / shapes as a freshly constructed model of this package (drop-in state_dict layout)
    from multimodal_transformer_amd.config import make_hyperparams
    from multimodal_transformer_amd.model import MultimodalGenerativeCVAE

    class Env:
        robot_type = None
    hp = make_hyperparams(mode, history=4, horizon=3, transformer={"nlayers": 2, "nhead": 2, "in_dim": 32, "fdim": 64, "output_size": 32})
    fresh = ModelRegistrar("/tmp", "cpu")
    MultimodalGenerativeCVAE(Env(), "VEHICLE", fresh, hp, "cpu", [])
    assert set(fresh.model_dict.keys()) == keys
    for name in keys:
        a, b = fresh.model_dict[name].state_dict(), reg.model_dict[name].state_dict()
        assert {k: tuple(v.shape) for k, v in a.items()} == {k: tuple(v.shape) for k, v in b.items()}, name


class _Foreign(torch.nn.Module):          # a module class no registrar checkpoint is made of
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.ones(2))


def test_foreign_pickle_is_refused(tmp_path):
    from multimodal_transformer_amd._lib import MmtError
    torch.save(torch.nn.ModuleDict({"VEHICLE/x": _Foreign()}), tmp_path / "model_registrar-1.pt")
    reg = ModelRegistrar(str(tmp_path), "cpu")
    with pytest.raises(MmtError, match="no business"):
        reg.load_models(1)
    reg.load_models(1, trusted=True)          # explicit opt-in: full unpickle
    assert isinstance(reg.model_dict["VEHICLE/x"], _Foreign)


def test_legacy_format_checkpoint_needs_the_opt_in(tmp_path):
    """A non-zip (legacy) torch.save file cannot be scanned for its globals: refused with a clear error unless trusted=True,
    which then unpickles it in full (the scan used to raise before the trusted branch was reached)."""
    from multimodal_transformer_amd._lib import MmtError
    torch.save(torch.nn.ModuleDict({"VEHICLE/x": _Foreign()}), tmp_path / "model_registrar-2.pt",
               _use_new_zipfile_serialization=False)
    reg = ModelRegistrar(str(tmp_path), "cpu")
    with pytest.raises(MmtError, match="trusted=True"):
        reg.load_models(2)
    reg.load_models(2, trusted=True)
    assert isinstance(reg.model_dict["VEHICLE/x"], _Foreign)
