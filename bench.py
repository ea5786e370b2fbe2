"""Benchmark of the hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 128] [--mode basic] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one `predict` of BASELINE.json configs[0] (Basic transformer --autoregressive predict,
VEHICLE, 8-step history, 12-step horizon) on a synthetic nuScenes-shaped batch of 128 agents that is
already resident in HBM.  With N GPUs every rank runs its own batch of 128 (agents are independent:
no data-path collective; weak scaling) and `value` = N*128*K / max-over-ranks time.

The JSON line also carries
  roofline     for the dominant kernel (the KV-cached autoregressive decoder): algorithmic FLOPs per launch
               (109.91 MFLOP/agent for decoder+head, KV-cached) / its average launch
               duration measured with HIP events on the launch stream, against the dense fp32 MFMA peak;
  cpu_baseline the CPU oracle in reference form (stock torch.nn modules, full recompute per step) timed on
               this box's host cores on the same batch (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from multimodal_transformer_amd.config import make_hyperparams  # noqa: E402
from multimodal_transformer_amd.synthetic import make_batch  # noqa: E402
from multimodal_transformer_amd.weights import apply_recipe  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3           # MI355X_MICROARCH.md: dense fp32 matrix peak
# algorithmic MFLOP per agent, KV-cached (SURVEY.md §8d / BASELINE.md §3)
MFLOP_ENCODER, MFLOP_DECODER, MFLOP_HEAD = 57.16, 109.88, 0.03
MFLOP_CNN = 11.60


class _Env:
    robot_type = None


def build_model(mode, dev, history, horizon):
    from multimodal_transformer_amd.model import ModelRegistrar, MultimodalGenerativeCVAE
    hp = make_hyperparams(mode, history=history, horizon=horizon)
    reg = ModelRegistrar("/tmp", dev)
    model = MultimodalGenerativeCVAE(_Env(), "VEHICLE", reg, hp, dev, [])
    apply_recipe(reg.model_dict, seed=0)
    return hp, model, reg


def time_kernels(model, reg, tensors, ph, iters):
    """Per-kernel average durations (ms) with HIP events on the stream the kernels are launched on."""
    from multimodal_transformer_amd import ops
    x, x_st, lanes, maps = tensors
    enc = reg.model_dict["VEHICLE/node_history_encoder"]
    dec = reg.model_dict["VEHICLE/decoder/transformer_decoder"]
    head = reg.model_dict["VEHICLE/decoder/MLP"]
    out = {}

    def avg(fn):
        fn()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for a, b in evs:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in evs) / iters

    memory, pad = enc.forward_with_mask(x_st)
    start = x_st[:, -1, 0:2].contiguous()
    out["encoder_ms"] = avg(lambda: enc.forward_with_mask(x_st))
    map_feat = fusion = None
    if maps is not None:
        me = reg.model_dict["VEHICLE/map_encoder"]
        map_feat = me.forward_raw(maps)
        fusion = reg.model_dict["VEHICLE/fusion/hist_map"]
        out["map_encoder_ms"] = avg(lambda: me.forward_raw(maps))
    packed = dec.packed(head=head, map_fusion=fusion)
    out["decoder_ms"] = avg(lambda: ops.decoder_forward(packed, memory, pad, start, ph, map_feat=map_feat))
    return out


def cpu_baseline(mode, hp, batch, ph, max_seconds=25.0):
    """Oracle in reference form on the host cores: 2 warm-up + up to 5 timed predicts (bounded)."""
    from oracle import reference_form as rf          # checker / baseline only, never the product path
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    ref = rf.ReferenceFormModel(hp)
    apply_recipe(ref.m, seed=0)
    ref.m.eval()
    c = {k: torch.from_numpy(v) for k, v in batch.items()}
    B = c["x"].shape[0]
    times = []
    t_begin = time.perf_counter()
    with torch.no_grad():
        for i in range(7):
            t0 = time.perf_counter()
            ref.predict(c["x"], c["x_st"], c.get("lanes"), c.get("map"), ph)
            dt = time.perf_counter() - t0
            if i >= 2:
                times.append(dt)
            if time.perf_counter() - t_begin > max_seconds and len(times) >= 1:
                break
    times.sort()
    med = times[len(times) // 2]
    return {"value": B / med, "unit": "agents/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} timed predicts of the same B={B} batch after 2 warm-ups, median; "
                      f"oracle/reference_form.py (stock torch.nn modules, full recompute per AR step), "
                      f"eval mode, no_grad, anomaly detection off; os.cpu_count()={os.cpu_count()}"}


def cpu_train_baseline(hp, batch, ph, max_seconds=25.0):
    """Reference-form training step (stock torch.nn modules, autograd, clip, torch AdamW; dropout 0, anomaly
    detection off) on the host cores: 1 warm-up + up to 3 timed steps (bounded)."""
    from oracle import reference_form as rf          # baseline only
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    ref = rf.ReferenceFormModel(hp)
    apply_recipe(ref.m, seed=0)
    ref.m.train()
    for m in ref.m.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0
    c = {k: torch.from_numpy(v) for k, v in batch.items()}
    opt = torch.optim.AdamW(ref.m.parameters(), lr=hp["learning_rate"])
    times = []
    t_begin = time.perf_counter()
    for i in range(4):
        t0 = time.perf_counter()
        opt.zero_grad()
        loss = ref.train_loss(c["x"], c["x_st"], c["y"], c["y_st"], c.get("lanes"), c.get("map"), ph)
        loss.backward()
        torch.nn.utils.clip_grad_value_(ref.m.parameters(), hp["grad_clip"])
        opt.step()
        dt = time.perf_counter() - t0
        if i >= 1:
            times.append(dt)
        if time.perf_counter() - t_begin > max_seconds and times:
            break
    times.sort()
    med = times[len(times) // 2]
    return {"value": 1.0 / med, "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} timed fwd+bwd+clip+AdamW steps at B={c['x'].shape[0]} after 1 warm-up, median; "
                      "oracle/reference_form.py, dropout 0, anomaly detection off"}


def bench_train(model, hp, dev, B, history, ph, rank, world, steps, warmup, dist, with_map=False, with_lanes=False):
    """BASELINE config 2: Basic transformer training step (fwd + bwd + clip + AdamW), data-parallel with one flat
    RCCL gradient all-reduce per step, 128 agents per GPU (global batch 128 * world)."""
    from multimodal_transformer_amd.training import Trainer
    hp["autoregressive"] = True
    batches = [make_batch(B, history, ph, seed=100 + 10 * rank + i, with_map=with_map, with_lanes=with_lanes) for i in range(4)]
    dev_batches = [(torch.from_numpy(b["x_st"]).to(dev), torch.from_numpy(b["y_st"]).to(dev),
                    torch.from_numpy(b["map"]).to(dev) if with_map else None,
                    torch.from_numpy(b["lanes"]).to(dev) if with_lanes else None) for b in batches]
    tr = Trainer(model, B, history + 1, ph, use_graph=True)
    for i in range(warmup):
        tr.step(*dev_batches[i % 4])
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = tr.step(*dev_batches[i % 4])
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(loss).all()
    n_par = sum(p.numel() for _, p in tr.engine.named_params)
    return {"metric": "train steps/sec (fwd+bwd+clip+AdamW)", "value": steps / elapsed, "unit": "steps/s",
            "ms_per_step": elapsed / steps * 1e3, "steps": steps, "warmup": warmup, "per_gpu_batch": B,
            "global_batch": B * world, "agents_per_s": B * world * steps / elapsed,
            "parallelism": f"dp{world}, one flat fp32 all-reduce of {n_par} gradients per step (RCCL)" if world > 1
            else "single GPU", "dropout": 0.0, "last_loss": float(loss),
            "algorithmic_tflops": 3 * (MFLOP_ENCODER + MFLOP_DECODER + MFLOP_HEAD) * 1e6 * B * world * steps / elapsed / 1e12}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=128, help="agents per GPU per step")
    ap.add_argument("--mode", default="basic", choices=["basic", "map_cnn", "map_vit", "lane_cnn"])
    ap.add_argument("--history", type=int, default=8)
    ap.add_argument("--horizon", type=int, default=12)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sweep", action="store_true", help="also time larger per-GPU batches (extra keys)")
    ap.add_argument("--decoder", default="auto", choices=["auto", "persistent", "cluster", "cluster_launches"],
                    help="kernel family behind mmt_decoder_forward (include/mmt_amd.h)")
    ap.add_argument("--no-train", action="store_true", help="skip the training-step leg (extra key 'train')")
    ap.add_argument("--train-steps", type=int, default=30)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # MMT_BENCH_FORCE_DIST=1: take the multi-rank branch (RCCL group, barriers, MAX over ranks) with one rank - a rehearsal on one GPU
    distributed = world > 1 or os.environ.get("MMT_BENCH_FORCE_DIST") == "1"
    if args.gpus != world and distributed:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and not distributed:
        raise SystemExit("launch multi-GPU runs with torch.distributed.run (one rank per GPU)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from multimodal_transformer_amd import ops
    ops.decoder_select(args.decoder)
    B, ph = args.batch, args.horizon
    hp, model, reg = build_model(args.mode, dev, args.history, ph)
    # every rank gets its own shard of the global synthetic batch (seeded by rank): no collective on the data path
    batch = make_batch(B, args.history, ph, seed=1 + rank, with_map=(args.mode in ("map_cnn", "map_vit")),
                       with_lanes=(args.mode == "lane_cnn"))
    t = {k: torch.from_numpy(v).to(dev) for k, v in batch.items()}
    x, x_st, lanes, maps = t["x"], t["x_st"], t.get("lanes"), t.get("map")

    def step():
        return model.predict(x, x_st, lanes, None, maps, ph)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if distributed:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    res = out[1] if args.mode == "lane_cnn" else out[0]
    assert torch.isfinite(res).all()

    train = None
    if not args.no_train:       # every rank takes part (all-reduce inside)
        try:
            train = bench_train(model, hp, dev, B, args.history, ph, rank, world, args.train_steps,
                                max(3, min(args.warmup, 5)), dist if distributed else None, with_map=(args.mode in ("map_cnn", "map_vit")),
                                with_lanes=(args.mode == "lane_cnn"))
        except Exception as e:       # the headline metric must survive a failure of the extra leg
            train = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        kt = time_kernels(model, reg, (x, x_st, lanes, maps), ph, iters=min(50, max(5, args.steps)))
        dec_flops = (MFLOP_DECODER + MFLOP_HEAD) * 1e6 * B * (1 if args.mode != "lane_cnn" or model.lane_dedup else 3)
        achieved = dec_flops / (kt["decoder_ms"] * 1e-3) / 1e12
        n_dec_par = sum(p.numel() for k, m in reg.model_dict.items() if "decoder" in k or "fusion" in k
                        for p in m.parameters())
        dec_bytes = 4.0 * (n_dec_par + B * (args.history + 1) * hp["transformer"]["in_dim"] + B * ph * 2)
        cluster = args.decoder != "persistent" and B <= 1280 and args.history + 1 <= 16 and ph <= 16
        fused = cluster and args.decoder != "cluster_launches" and ((B + 3) // 4) * 8 <= 256 and not os.environ.get("MMT_NO_FUSED_DECODER")
        if fused:
            dec_kernel = ("mmt_dec_fused4_kernel (all %d phases of the decode in one launch, in-kernel cluster hand-offs) + "
                          "mmt_dec_ckv_kernel + mmt_dec_phase_kernel (final head)" % (3 * 6 * ph))
        elif cluster:
            dec_kernel = "mmt_dec_phase4_kernel / mmt_dec_phase_kernel x%d + mmt_dec_ckv_kernel (cluster decoder)" % (3 * 6 * ph + 1)
        else:
            dec_kernel = "mmt_decoder_kernel (persistent decoder)"
        traffic, traffic_src = None, None
        tf = os.path.join(ROOT, "profiles", "r01", "traffic_b128_basic_v4.json")
        if fused and args.mode == "basic" and B == 128 and ph == 12 and args.history == 8 and os.path.exists(tf):
            # HBM-side bytes per decode from separate rocprofv3 --pmc passes over this same command (tools/pmc_summary.py)
            traffic = json.load(open(tf))["decoder_bytes_per_predict"]
            traffic_src = "profiles/r01/traffic_b128_basic_v4.json (FETCH_SIZE x2 + WRITE_SIZE; Infinity-Cache hits included)"
        line = {
            "metric": "predicted agents/sec (12-step, batch 128)",
            "value": value, "unit": "agents/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic nuScenes-shaped batch (multimodal_transformer_amd/synthetic.py), "
                                    "random-init weights from the by-name recipe",
            "config": {"workload": f"{args.mode} transformer --autoregressive predict, VEHICLE, batch {B} per GPU, "
                                   f"{args.history}-step history, {ph}-step horizon",
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"batch-split x{world}, no collective"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "kernel": dec_kernel, "launch_unit": "one 12-step decode (all of its launches)",
                         "avg_launch_ms": kt["decoder_ms"], "algorithmic_flop_per_launch": dec_flops,
                         # the HBM view of the same launches (north_star asks for it): weights once + memory + outputs
                         "algorithmic_bytes_per_launch": dec_bytes,
                         "achieved_gbs_vs_hbm": dec_bytes / (kt["decoder_ms"] * 1e-3) / 1e9, "hbm_peak_gbs": 8000.0},
            "kernels_ms": kt,
        }
        # every kernel of the predict against both roofs (SURVEY.md section 8d figures; north star: HBM GB/s for the attention /
        # decoder / head kernels, MFMA utilisation for the map encoders): algorithmic FLOPs and bytes per launch / measured time
        n_enc_par = sum(p.numel() for k, m in reg.model_dict.items() if "node_history_encoder" in k for p in m.parameters()
                        if p.dim() < 3)
        enc_bytes = 4.0 * (n_enc_par + B * (args.history + 1) * (8 + hp["transformer"]["in_dim"]))
        kr = {"encoder": {"ms": kt["encoder_ms"], "flop": MFLOP_ENCODER * 1e6 * B, "bytes": enc_bytes},
              "decoder+head": {"ms": kt["decoder_ms"], "flop": dec_flops, "bytes": dec_bytes}}
        if "map_encoder_ms" in kt:
            n_map_par = sum(p.numel() for k, m in reg.model_dict.items() if "map_encoder" in k for p in m.parameters())
            kr["map_encoder"] = {"ms": kt["map_encoder_ms"], "flop": (MFLOP_CNN if args.mode == "map_cnn" else 179.4) * 1e6 * B,
                                 "bytes": 4.0 * (n_map_par + B * 3 * 100 * 100 + B * 32)}
        for v in kr.values():
            v["tflops"] = v["flop"] / (v["ms"] * 1e-3) / 1e12
            v["mfma_f32_frac"] = v["tflops"] / PEAK_F32_MFMA_TFLOPS
            v["gbs"] = v["bytes"] / (v["ms"] * 1e-3) / 1e9
            v["hbm_frac"] = v["gbs"] / 8000.0
        line["kernel_rooflines"] = kr
        if args.sweep and not distributed:
            sweep = {}
            for bb in (1024, 4096):
                bt = make_batch(bb, args.history, ph, seed=7, with_map=(args.mode in ("map_cnn", "map_vit")),
                                with_lanes=(args.mode == "lane_cnn"))
                tt_ = {k: torch.from_numpy(v).to(dev) for k, v in bt.items()}
                f = lambda: model.predict(tt_["x"], tt_["x_st"], tt_.get("lanes"), None, tt_.get("map"), ph)
                for _ in range(3):
                    f()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                n = 10
                for _ in range(n):
                    f()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t1) / n
                sweep[str(bb)] = {"agents_per_s": bb / dt, "ms_per_step": dt * 1e3,
                                  "mfma_frac": (MFLOP_ENCODER + MFLOP_DECODER + MFLOP_HEAD) * 1e6 * bb / dt / 1e12
                                  / PEAK_F32_MFMA_TFLOPS}
            line["batch_sweep"] = sweep
        line["train"] = train
        if not args.no_cpu_baseline and not distributed:
            line["cpu_baseline"] = cpu_baseline(args.mode, hp, batch, ph)
            if train is not None and "error" not in train:
                train["cpu_baseline"] = cpu_train_baseline(hp, batch, ph)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
