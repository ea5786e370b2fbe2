"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports what include/mmt_amd.h
declares; host-side argument validation fails loudly; no compute is launched here."""
import ctypes
import os
import re

import pytest
import torch

from multimodal_transformer_amd import _lib, ops
from multimodal_transformer_amd._lib import MmtError

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "mmt_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(mmt_[a-z0-9_]+)\s*\(", hdr)))


def test_header_and_binding_agree():
    names = _declared()
    assert names, "no declarations found"
    assert sorted(_lib.SIGNATURES) == names


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.mmt_version() >= 1
    assert lib.mmt_packed_linear_floats(768, 256) == 768 * 256
    assert lib.mmt_packed_linear_floats(2, 32) == 16 * 32
    assert lib.mmt_packed_linear_floats(256, 8) == 256 * 16


def test_struct_sizes_match_header_layout():
    # natural C layout of the header structs (LP64): catches a binding that drifted from the header
    assert ctypes.sizeof(_lib.Linear) == 24
    assert ctypes.sizeof(_lib.Norm) == 16
    assert ctypes.sizeof(_lib.EncLayer) == 4 * 24 + 2 * 16
    assert ctypes.sizeof(_lib.DecLayer) == 7 * 24 + 3 * 16 + 6 * 8
    assert ctypes.sizeof(_lib.Mlp) == 8 + 3 * 24 + 3 * 16 + 24
    assert ctypes.sizeof(_lib.Encoder) == 24 + 24 + 8 + 8 * ctypes.sizeof(_lib.EncLayer)


def test_cpu_tensors_are_rejected_not_silently_computed():
    with pytest.raises(MmtError):
        ops._chk(torch.zeros(2, 3), "x")
    from multimodal_transformer_amd.nn_component import Encoder
    enc = Encoder(ninp=8, in_dim=32, nhead=2, fdim=64, nlayers=1)
    with pytest.raises(MmtError):
        enc(torch.zeros(2, 5, 8), None, None)


def test_null_and_empty_arguments_return_einval():
    lib = _lib.load()
    enc = _lib.Encoder()
    rc = lib.mmt_encoder_forward(ctypes.byref(enc), None, 4, 9, None, None, None)
    assert rc == -1 and b"null" in lib.mmt_last_error()
    dec = _lib.Decoder()
    assert lib.mmt_decoder_workspace_floats(ctypes.byref(dec), 0, 9, 12) == 0
    rc = lib.mmt_decoder_forward(ctypes.byref(dec), None, None, None, None, 0, None, 0, None, 4, 9, 12, None, None,
                                 None, None)
    assert rc == -1


def test_batch_error_reporting_helpers(capsys):
    """evaluation.print_batch_errors / log_batch_errors (evaluation.py:166-230): pooled mean / median over the per-batch dicts
    that compute_batch_statistics returns; host logic only."""
    import numpy as np
    from multimodal_transformer_amd.evaluation import log_batch_errors, print_batch_errors

    class NT:
        name = "VEHICLE"
    nt = NT()
    batches = [{nt: {"ade": [1.0, 3.0], "fde": [2.0], "kde": []}}, {nt: {"ade": [5.0], "fde": [4.0, 9.0], "kde": []}}]
    print_batch_errors(batches, "eval", 7)
    out = capsys.readouterr().out.splitlines()
    assert out == ["7: VEHICLE/eval/ade_mean 3.0", "7: VEHICLE/eval/ade_median 3.0", "7: VEHICLE/eval/fde_mean 5.0",
                   "7: VEHICLE/eval/fde_median 4.0"]

    class Writer:
        def __init__(self):
            self.scalars, self.hists = {}, {}

        def add_scalar(self, tag, value, it):
            self.scalars[tag] = (float(value), it)

        def add_histogram(self, tag, values, it):
            self.hists[tag] = (list(values), it)
    w = Writer()
    log_batch_errors(batches, w, "eval", 7)
    assert w.scalars["VEHICLE/eval/ade_mean"] == (3.0, 7) and w.scalars["VEHICLE/eval/fde_median"] == (4.0, 7)
    assert w.hists["VEHICLE/eval/fde"] == ([2.0, 4.0, 9.0], 7) and "VEHICLE/eval/kde" not in w.hists
    with pytest.raises(NotImplementedError):
        log_batch_errors(batches, w, "eval", 7, bar_plot=["kde"])
    assert np.isclose(w.scalars["VEHICLE/eval/ade_median"][0], 3.0)
