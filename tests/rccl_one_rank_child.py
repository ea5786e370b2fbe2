"""Child process of test_one_rank_rccl_all_reduce_between_graph_replays (not collected by pytest): the training step with a
one-rank RCCL group -- two backward graphs with the decoder-side bucket reduced beside the second one, and round 1's one-bucket
form -- against a run without a group.  Then the ORDERED teardown: every Trainer closed (Trainer.close():
synchronise, drop the last Work, the timing events and the captured graphs), barrier, synchronise, destroy_process_group, and a
normal interpreter exit -- the exit code is the real one; the parent test prints this process's stderr when it is not 0 (an
in-process run under pytest's fd capture had swallowed the C++ side's message of the one abort seen in round 2)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(rdzv):
    import torch.distributed as dist
    from multimodal_transformer_amd import training
    from multimodal_transformer_amd.config import make_hyperparams
    from multimodal_transformer_amd.synthetic import make_batch
    from multimodal_transformer_amd.weights import apply_recipe
    from test_gpu_training import _model
    dev = torch.device("cuda:0")

    def run(reduce, overlap=True):
        hp = make_hyperparams("basic", history=8, horizon=12)
        model, reg = _model(hp, dev)
        apply_recipe(reg.model_dict, seed=2)
        tr = training.Trainer(model, 128, 9, 12, use_graph=True, always_reduce=reduce, overlap=overlap)
        assert tr.always_reduce == reduce and tr.overlap == (reduce and overlap)
        tr.time_allreduce = reduce
        losses = []
        for it in range(4):
            b = make_batch(128, 8, 12, seed=20 + it)
            losses.append(float(tr.step(torch.from_numpy(b["x_st"]).to(dev), torch.from_numpy(b["y_st"]).to(dev))))
        if reduce:
            assert tr.allreduce_ms() is not None and tr.allreduce_ms() >= 0.0
        out = losses, torch.cat([p.detach().reshape(-1) for p in reg.parameters()]).clone()
        tr.close()
        return out

    plain = run(False)
    dist.init_process_group("nccl", init_method=f"file://{rdzv}", rank=0, world_size=1, device_id=dev)
    dist.barrier()
    reduced = run(True)                    # two backward graphs, the decoder-side bucket reduced beside the second one
    one_bucket = run(True, overlap=False)  # round 1's form: one bucket behind the whole backward
    torch.cuda.synchronize()
    assert plain[0] == reduced[0] and torch.equal(plain[1], reduced[1])
    assert plain[0] == one_bucket[0] and torch.equal(plain[1], one_bucket[1])
    print("one-rank RCCL step: identical", plain[0], flush=True)
    dist.barrier()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print("teardown: destroy_process_group returned", flush=True)


if __name__ == "__main__":
    main(sys.argv[1])
    print("child: leaving through the normal interpreter exit", flush=True)
