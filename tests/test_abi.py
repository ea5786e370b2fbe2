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


def test_group_launch_planner_validates_jobs_and_lays_out_the_table():
    """mmt_gemm_group_plan / mmt_ln_grad_group_plan are host-only: sizes, the block map, and a job the group kernel does not take."""
    lib = _lib.load()
    jobs = (_lib.GemmArgs * 2)()
    for j, (M, N, K) in zip(jobs, ((64, 96, 1536), (36, 32, 1000))):        # dY^T X forms: A (M,K) and B (K,N) with unit stride in M / N
        j.A, j.sam, j.sak = 0x1000, 1, M
        j.B, j.sbk, j.sbn = 0x2000, N, 1
        j.C, j.ldc, j.M, j.N, j.K = 0x3000, N, M, N, K
        j.splitk, j.part_stride = 4, 1 << 20
    nbytes, nblocks = ctypes.c_size_t(0), ctypes.c_int(0)
    assert lib.mmt_gemm_group_plan(jobs, 2, None, 0, ctypes.byref(nbytes), ctypes.byref(nblocks)) == 0
    assert nblocks.value == (2 * 3 + 2 * 1) * 4 and nbytes.value == 2 * ctypes.sizeof(_lib.GemmArgs) + 16 * nblocks.value
    buf = (ctypes.c_uint8 * nbytes.value)()
    assert lib.mmt_gemm_group_plan(jobs, 2, buf, nbytes.value - 1, ctypes.byref(nbytes), ctypes.byref(nblocks)) != 0     # too small
    assert lib.mmt_gemm_group_plan(jobs, 2, buf, nbytes.value, ctypes.byref(nbytes), ctypes.byref(nblocks)) == 0
    assert bytes(buf[:2 * ctypes.sizeof(_lib.GemmArgs)]) == bytes(jobs)
    blocks = (ctypes.c_int32 * (4 * nblocks.value)).from_buffer_copy(bytes(buf[2 * ctypes.sizeof(_lib.GemmArgs):]))
    rows = [tuple(blocks[4 * i:4 * i + 4]) for i in range(nblocks.value)]
    # every tile exactly once; the jobs are dealt over 8 lanes (table entry 8 i + x = i-th tile of lane x: workgroups go round-robin
    # over the 8 XCDs, so a job's tiles share one L2) - here job 0 on lane 0, job 1 on lane 1, until job 1 runs out
    assert len(set(rows)) == len(rows) and sum(1 for r in rows if r[0] == 0) == 24
    assert set(rows) == {(0, x, y, z) for x in range(3) for y in range(2) for z in range(4)} | \
        {(1, 0, y, z) for y in range(2) for z in range(4)}
    assert rows[0] == (0, 0, 0, 0) and rows[1] == (1, 0, 0, 0) and [r[0] for r in rows[:16]] == [0, 1] * 8
    assert all(r[0] == 0 for r in rows[16:])
    jobs[1].relu = 1                                            # an epilogue: not a split-K weight-gradient job
    assert lib.mmt_gemm_group_plan(jobs, 2, None, 0, ctypes.byref(nbytes), ctypes.byref(nblocks)) != 0
    assert b"job 1" in lib.mmt_last_error()
    assert lib.mmt_gemm_group(None, 2, 8, None) != 0 and lib.mmt_ln_grad_group(None, 1, 1, None) != 0
    ln = (_lib.LnGradJob * 1)()
    ln[0].dy, ln[0].xhat, ln[0].gw, ln[0].gb = 0x1000, 0x2000, 0x3000, 0x4000
    ln[0].ld, ln[0].part_stride, ln[0].M, ln[0].N, ln[0].splitk = 256, 1 << 20, 1536, 256, 4
    assert lib.mmt_ln_grad_group_plan(ln, 1, None, 0, ctypes.byref(nbytes), ctypes.byref(nblocks)) == 0
    assert nblocks.value == 16 * 4 and nbytes.value == ctypes.sizeof(_lib.LnGradJob) + 16 * 64
    ln[0].splitk = 0
    assert lib.mmt_ln_grad_group_plan(ln, 1, None, 0, ctypes.byref(nbytes), ctypes.byref(nblocks)) != 0
