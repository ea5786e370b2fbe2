"""CPU tier, world_size 2 over gloo: the data-parallel training step (equal shards, one flat all-reduce, clip after
averaging) must reproduce the single-process step on the concatenated batch.  Kernels are the torch stand-ins of
tests/fake_kernels.py (host logic test); the RCCL path on GPUs runs the same code with backend nccl."""
import os
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    import fake_kernels
    from multimodal_transformer_amd import _train_kernels as K, training
    fake_kernels.install()
    from helpers import TINY_TR
    from multimodal_transformer_amd.config import make_hyperparams
    from multimodal_transformer_amd.model import ModelRegistrar, MultimodalGenerativeCVAE
    from multimodal_transformer_amd.synthetic import make_batch
    from multimodal_transformer_amd.weights import apply_recipe

    class Env:
        robot_type = None

    def build():
        hp = make_hyperparams("basic", history=4, horizon=3, transformer=TINY_TR)
        hp["autoregressive"] = True
        reg = ModelRegistrar("/tmp", "cpu")
        model = MultimodalGenerativeCVAE(Env(), "VEHICLE", reg, hp, "cpu", [])
        apply_recipe(reg.model_dict, seed=9)
        return model, reg

    B = 6
    batches = [make_batch(B, 4, 3, seed=30 + i) for i in range(2)]
    shard = slice(rank * B // world, (rank + 1) * B // world)
    model, reg = build()
    tr = training.Trainer(model, B // world, 5, 3, use_graph=False, dropout=False)
    assert tr.world == world and tr.overlap            # two buckets: the decoder side reduced beside the encoder's backward
    assert tr.engine.early_range[1] > tr.engine.early_range[0]
    for b in batches:
        tr.step(torch.from_numpy(b["x_st"][shard]), torch.from_numpy(b["y_st"][shard]))
    ref_model, ref_reg = build()
    ref = training.Trainer(ref_model, B, 5, 3, use_graph=False, distributed=False, dropout=False)
    for b in batches:
        ref.step(torch.from_numpy(b["x_st"]), torch.from_numpy(b["y_st"]))
    worst = 0.0
    for (n, p), (_, q) in zip(reg.named_parameters(), ref_reg.named_parameters()):
        worst = max(worst, float((p - q).abs().max()))
    # gradients of the averaged shards equal the full-batch gradient up to fp32 summation order
    g_worst = float((tr.engine.flat_grad / world - ref.engine.flat_grad).abs().max())
    np.save(os.path.join(out_dir, f"r{rank}.npy"), np.array([worst, g_worst, float(tr.state[0])]))
    dist.destroy_process_group()


def test_two_rank_step_equals_single_process_step(tmp_path):
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        worst, g_worst, step = np.load(tmp_path / f"r{r}.npy")
        # first AdamW steps move noise-gradient elements by up to 2*lr regardless of the summation order
        assert worst <= 2.1 * 4e-4 * 2, worst
        assert g_worst <= 1e-6, g_worst
        assert step == 3.0
