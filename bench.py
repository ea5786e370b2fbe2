"""Benchmark of the hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 128] [--mode basic] [--no-cpu-baseline]

`--gpus N` with N > 1 and no WORLD_SIZE in the environment launches the N ranks itself (a child
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py ...`, started before this
process has touched a GPU) and relays rank 0's JSON line; started under torch.distributed.run it is a rank.

A "step" is one `predict` of BASELINE.json configs[0] (Basic transformer --autoregressive predict, VEHICLE, 8-step history,
12-step horizon) on a synthetic nuScenes-shaped batch of 128 agents that is already resident in HBM.  With N GPUs every rank
runs its own batch of 128 (agents are independent: no data-path collective; weak scaling) and `value` = N*128*K / max-over-ranks
time.  The same JSON line carries
  roofline     for the dominant kernel (the KV-cached autoregressive decoder): algorithmic FLOPs per launch (SURVEY.md 8d:
               109.91 MFLOP/agent for decoder+head, KV-cached) / its average launch duration measured with HIP events on the
               launch stream, against the dense fp32 MFMA peak; `traffic` only from a PMC pass over THIS tree's kernels;
  cpu_baseline the CPU oracle in reference form (stock torch.nn modules, full recompute per step) timed on this box's host cores
               on the same batch (rank 0, N=1 only);
  scalings     the three scalings the north star names: weak (128 agents per GPU), strong (the 128-agent batch split N ways),
               and the Lane_encoding_CNN batch of 8192 split N ways;
  configs      BASELINE configs 3-5 at the bench batch (N=1 only): map_cnn, map_vit (parity unpinned), lane_cnn at 128 and 1024;
  train        the training step (config 2), with the gradient all-reduce timed separately.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from multimodal_transformer_amd.config import make_hyperparams  # noqa: E402
from multimodal_transformer_amd.synthetic import make_batch  # noqa: E402
from multimodal_transformer_amd.weights import apply_recipe  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3           # MI355X_MICROARCH.md: dense fp32 matrix peak
HBM_PEAK_GBS = 8000.0
# algorithmic MFLOP per agent, KV-cached (SURVEY.md §8d / BASELINE.md §3)
MFLOP_ENCODER, MFLOP_DECODER, MFLOP_HEAD = 57.16, 109.88, 0.03
MFLOP_CNN, MFLOP_VIT = 11.60, 179.4


class _Env:
    robot_type = None


def build_model(mode, dev, history, horizon):
    from multimodal_transformer_amd.model import ModelRegistrar, MultimodalGenerativeCVAE
    hp = make_hyperparams(mode, history=history, horizon=horizon)
    reg = ModelRegistrar("/tmp", dev)
    model = MultimodalGenerativeCVAE(_Env(), "VEHICLE", reg, hp, dev, [])
    apply_recipe(reg.model_dict, seed=0)
    return hp, model, reg


PREDICT_SOURCES = ("decoder_cluster.hip", "transformer.hip", "encoder_split.hip", "mmt_device.cuh", "mmt_gfx950.cuh")


def kernel_source_hash(names=PREDICT_SOURCES):
    """sha256 over the sources of the predict path's kernels (encoder + decoder; the training / context kernels live in other
    files): the traffic summary under profiles/ records it, so that a stale one is never quoted."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "multimodal_transformer_amd", "csrc")
    for name in sorted(names):
        with open(os.path.join(csrc, name), "rb") as f:
            h.update(name.encode())
            h.update(f.read())
    return h.hexdigest()[:16]


def time_kernels(model, reg, tensors, ph, iters):
    """Per-kernel average durations (ms) with HIP events on the stream the kernels are launched on."""
    from multimodal_transformer_amd import ops
    x, x_st, lanes, maps = tensors
    enc = reg.model_dict["VEHICLE/node_history_encoder"]
    dec = reg.model_dict["VEHICLE/decoder/transformer_decoder"]
    head = reg.model_dict["VEHICLE/decoder/MLP"]
    out = {}

    def avg(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for a, b in evs:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        # median: an iteration in which the host fell behind the GPU (its start event is stamped before the launches arrive)
        # or the clocks had not ramped yet is not the kernels' time
        ts = sorted(a.elapsed_time(b) for a, b in evs)
        return ts[len(ts) // 2]

    memory, pad = enc.forward_with_mask(x_st)
    start = x_st[:, -1, 0:2].contiguous()
    out["encoder_ms"] = avg(lambda: enc.forward_with_mask(x_st))
    map_feat = fusion = mem_feat = memfuse = None
    if maps is not None:
        me = reg.model_dict["VEHICLE/map_encoder"]
        map_feat = me.forward_raw(maps)
        fusion = reg.model_dict["VEHICLE/fusion/hist_map"]
        out["map_encoder_ms"] = avg(lambda: me.forward_raw(maps))
    if lanes is not None:
        le = reg.model_dict["VEHICLE/lane_encoder"]
        flat = lanes.contiguous().view(-1, 2, lanes.shape[2])
        mem_feat = le(flat).view(lanes.shape[0], lanes.shape[1], -1)[:, 0, :].contiguous()
        memfuse = reg.model_dict["VEHICLE/fusion/lane_hist"]
        out["lane_encoder_ms"] = avg(lambda: le(flat))
    packed = dec.packed(head=head, map_fusion=fusion, mem_fusion=memfuse) if memfuse is not None else dec.packed(head=head, map_fusion=fusion)
    out["decoder_ms"] = avg(lambda: ops.decoder_forward(packed, memory, pad, start, ph, map_feat=map_feat, mem_feat=mem_feat))
    out["decoder_path"] = ops.decoder_last_path()[1]
    return out


def kernel_rooflines(mode, B, history, ph, hp, reg, kt, n_decodes=1):
    """Every kernel of the predict against both roofs (SURVEY.md 8d figures): algorithmic FLOPs and bytes per launch / time."""
    n_enc_par = sum(p.numel() for k, m in reg.model_dict.items() if "node_history_encoder" in k for p in m.parameters() if p.dim() < 3)
    n_dec_par = sum(p.numel() for k, m in reg.model_dict.items() if "decoder" in k or "fusion" in k for p in m.parameters())
    d = hp["transformer"]["in_dim"]
    enc_bytes = 4.0 * (n_enc_par + B * (history + 1) * (8 + d))
    dec_flops = (MFLOP_DECODER + MFLOP_HEAD) * 1e6 * B * n_decodes
    dec_bytes = 4.0 * (n_dec_par + B * (history + 1) * d + B * ph * 2)
    kr = {"encoder": {"ms": kt["encoder_ms"], "flop": MFLOP_ENCODER * 1e6 * B, "bytes": enc_bytes},
          "decoder+head": {"ms": kt["decoder_ms"] * n_decodes, "flop": dec_flops, "bytes": dec_bytes}}
    if "map_encoder_ms" in kt:
        n_map_par = sum(p.numel() for k, m in reg.model_dict.items() if "map_encoder" in k for p in m.parameters())
        kr["map_encoder"] = {"ms": kt["map_encoder_ms"], "flop": (MFLOP_CNN if mode == "map_cnn" else MFLOP_VIT) * 1e6 * B,
                             "bytes": 4.0 * (n_map_par + B * 3 * 100 * 100 + B * 32)}
    for v in kr.values():
        v["tflops"] = v["flop"] / (v["ms"] * 1e-3) / 1e12
        v["mfma_f32_frac"] = v["tflops"] / PEAK_F32_MFMA_TFLOPS
        v["gbs"] = v["bytes"] / (v["ms"] * 1e-3) / 1e9
        v["hbm_frac"] = v["gbs"] / HBM_PEAK_GBS
    return kr, dec_flops, dec_bytes


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


class _Group:
    """Barrier / max-over-ranks helpers that also work with one rank (then they are no-ops)."""

    def __init__(self, dist, dev):
        self.dist, self.dev = dist, dev

    def sync(self):
        if self.dev.type == "cuda":
            torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        if self.dev.type == "cuda":
            torch.cuda.synchronize()

    def max(self, v):
        if self.dist is None:
            return v
        tt = torch.tensor([v], dtype=torch.float64, device=self.dev)
        self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
        return float(tt.item())


def timed(group, fn, steps, warmup):
    """W untimed + exactly K timed calls bracketed by barrier + synchronize on both sides; MAX over ranks."""
    out = None
    for _ in range(warmup):
        out = fn()
    group.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    group.sync()
    return group.max(time.perf_counter() - t0), out


def bench_train(model, hp, dev, B, history, ph, rank, world, steps, warmup, group, with_map=False, with_lanes=False,
                dropout=False):
    """BASELINE config 2: Basic transformer training step (fwd + bwd + clip + AdamW), data-parallel with one flat
    RCCL gradient all-reduce per step, 128 agents per GPU (global batch 128 * world).  ``dropout`` False: the dropout-free graph
    (what training parity and the CPU baseline are defined on); True: the modules' own probabilities (0.2 Transformer / PE, 0.5
    map feature, 0.1 ViT), masks drawn in-graph every step."""
    from multimodal_transformer_amd.training import Trainer
    hp["autoregressive"] = True
    batches = [make_batch(B, history, ph, seed=100 + 10 * rank + i, with_map=with_map, with_lanes=with_lanes) for i in range(4)]
    dev_batches = [(torch.from_numpy(b["x_st"]).to(dev), torch.from_numpy(b["y_st"]).to(dev),
                    torch.from_numpy(b["map"]).to(dev) if with_map else None,
                    torch.from_numpy(b["lanes"]).to(dev) if with_lanes else None) for b in batches]
    tr = Trainer(model, B, history + 1, ph, use_graph=True, dropout=None if dropout else False, seed=1234)
    if dropout and not tr.engine.drop.active:
        raise RuntimeError("bench_train: dropout requested but the modules carry no dropout probability")
    it = [0]

    def step():
        it[0] += 1
        return tr.step(*dev_batches[it[0] % 4])

    for _ in range(warmup):
        step()
    tr.time_allreduce = True
    elapsed, loss = timed(group, step, steps, 0)
    assert torch.isfinite(loss).all()
    n_par = sum(p.numel() for _, p in tr.engine.named_params)
    ar_ms, overlap = tr.allreduce_ms(), tr.overlap
    drop_info = None
    if dropout:
        drop_info = {"transformer_and_pe": float(tr.engine.enc.pos_encoder.dropout.p),
                     "mask_bytes_per_step": int(tr.engine.drop.pool.numel()),
                     "decode": (f"{_last_decoder_path()}, reading the step's keep masks"
                                if tr.engine._ar_decode_takes_masks() else
                                "step by step on the stash kernels (tokens depend on the masks)")}
    eng = tr.engine
    forms = {"decode": ("mmt_decoder_forward_stash (pipeline decode that keeps the backward's stash)" if eng._stash_decode_ok()
                        else "decode + all-steps recomputation of the stash"),
             "sweep": ("mmt_sweep_pipe_run (one weight-stationary pipeline launch)" if eng._sweep_pipe() is not None
                       else f"launch-per-kernel sweep ({getattr(eng, 'sweep_pipe_why', '')})")}
    tr.close()      # ordered teardown (graphs, events, the last Work) before anybody destroys the process group
    return {"metric": "train steps/sec (fwd+bwd+clip+AdamW)", "forms": forms, "value": steps / elapsed, "unit": "steps/s",
            "ms_per_step": elapsed / steps * 1e3, "steps": steps, "warmup": warmup, "per_gpu_batch": B,
            "global_batch": B * world, "agents_per_s": B * world * steps / elapsed,
            "allreduce_ms": ar_ms, "allreduce_bytes": 4 * n_par if world > 1 else 0,
            "allreduce_form": ("two buckets: the decoder-side gradients reduced asynchronously beside the history encoder's backward "
                               "graph, the rest behind it; allreduce_ms = what the all-reduce adds between backward and optimizer "
                               "(HIP events: first bucket issued -> last bucket reduced, minus that backward graph)")
                              if overlap else ("one bucket behind the backward" if world > 1 else None),
            "parallelism": f"dp{world}, flat fp32 gradient of {n_par} elements all-reduced per step (RCCL) in two buckets, parameters "
                           f"broadcast from rank 0 at start" if world > 1 else "single GPU",
            "dropout": drop_info if dropout else 0.0,
            "last_loss_rank0": float(loss),
            "algorithmic_tflops": 3 * (MFLOP_ENCODER + MFLOP_DECODER + MFLOP_HEAD) * 1e6 * B * world * steps / elapsed / 1e12}


def _last_decoder_path():
    from multimodal_transformer_amd import ops
    return ops.decoder_last_path()[1]


def predict_fn(model, mode, dev, B, history, ph, seed):
    batch = make_batch(B, history, ph, seed=seed, with_map=(mode in ("map_cnn", "map_vit")), with_lanes=(mode == "lane_cnn"))
    t = {k: torch.from_numpy(v).to(dev) for k, v in batch.items()}
    x, x_st, lanes, maps = t["x"], t["x_st"], t.get("lanes"), t.get("map")
    return (lambda: model.predict(x, x_st, lanes, None, maps, ph)), (x, x_st, lanes, maps), batch


def sync_predict_ms(model, mode, dev, B, history, ph, iters=100):
    """A SYNCHRONOUS caller (evaluate.py copies every prediction to the host, trajectron.py:216-221): wall time per predict +
    ``.cpu()``, eager and replayed from the captured HIP graph (``MultimodalGenerativeCVAE.predict_graphed``)."""
    fn, _, _ = predict_fn(model, mode, dev, B, history, ph, seed=9)
    pick = (lambda o: o[1]) if mode == "lane_cnn" else (lambda o: o[0])
    out = {}
    for name, graphed in (("eager", False), ("graph", True)):
        model.graph_predict = graphed
        try:
            for _ in range(5):
                pick(fn()).cpu()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                pick(fn()).cpu()
            out[name] = (time.perf_counter() - t0) / iters * 1e3
        finally:
            model.graph_predict = False
    return out


def config_report(mode, B, dev, history, ph, steps, group, train_steps=0):
    """One of BASELINE configs 3-5 on this GPU: whole-predict throughput + per-kernel times and rooflines (+ its training step:
    the 8-GPU run's single-GPU denominator)."""
    hp, model, reg = build_model(mode, dev, history, ph)
    fn, tensors, _ = predict_fn(model, mode, dev, B, history, ph, seed=5)
    elapsed, out = timed(group, fn, steps, 3)
    res = out[1] if mode == "lane_cnn" else out[0]
    assert torch.isfinite(res).all()
    kt = time_kernels(model, reg, tensors, ph, iters=min(20, steps))
    kr, _, _ = kernel_rooflines(mode, B, history, ph, hp, reg, kt, 1 if mode != "lane_cnn" or model.lane_dedup else 3)
    rep = {"workload": f"{mode} transformer --autoregressive predict, VEHICLE, batch {B}, {history}-step history, {ph}-step horizon",
           "value": B * steps / elapsed, "unit": "agents/s", "ms_per_step": elapsed / steps * 1e3, "steps": steps,
           "kernels_ms": kt, "kernel_rooflines": kr}
    if B <= 256:        # the same predict replayed from a HIP graph (model.graph_predict = True: a serving loop's form)
        try:
            model.graph_predict = True
            el, _ = timed(group, fn, steps, 5)
            rep["graph_replay_ms_per_step"] = el / steps * 1e3
        except Exception as e:
            rep["graph_replay_ms_per_step"] = f"{type(e).__name__}: {e}"
        finally:
            model.graph_predict = False
    if mode == "map_vit":
        rep["parity"] = "unpinned (vit_pytorch==0.33.2 is neither in the reference tree nor installable offline; checked against the CPU restatement only)"
    if mode == "lane_cnn":
        rep["note"] = "the reference decodes all three lanes with lane 0's feature (mgcvae.py:404): one decode, returned three times"
    if train_steps:
        try:
            t = bench_train(model, hp, dev, B, history, ph, 0, 1, train_steps, 3, group, with_map=(mode in ("map_cnn", "map_vit")),
                            with_lanes=(mode == "lane_cnn"))
            rep["train"] = {k: t[k] for k in ("value", "unit", "ms_per_step", "steps", "per_gpu_batch", "last_loss_rank0")}
        except Exception as e:
            rep["train"] = {"error": f"{type(e).__name__}: {e}"}
    return rep


def self_launch(n, argv):
    """Start the N ranks as a child `torch.distributed.run` (before this process touches a GPU) and relay its output."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


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
    ap.add_argument("--decoder", default="auto", choices=["auto", "persistent", "cluster", "cluster_launches", "cluster_fused", "cluster_pipeline"],
                    help="kernel family behind mmt_decoder_forward (include/mmt_amd.h)")
    ap.add_argument("--no-train", action="store_true", help="skip the training-step leg (extra key 'train')")
    ap.add_argument("--no-configs", action="store_true", help="skip BASELINE configs 3-5 (extra key 'configs')")
    ap.add_argument("--no-scalings", action="store_true", help="skip the strong / lane-8192 scalings (extra key 'scalings')")
    ap.add_argument("--train-steps", type=int, default=30)
    ap.add_argument("--rehearse", action="store_true",
                    help="CPU rehearsal of the multi-rank control flow (gloo, a stand-in step, no kernels): tests the launcher")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # MMT_BENCH_FORCE_DIST=1: take the multi-rank branch (RCCL group, barriers, MAX over ranks) with one rank - a rehearsal on one GPU
    distributed = world > 1 or os.environ.get("MMT_BENCH_FORCE_DIST") == "1"
    if args.gpus != world and distributed:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    if args.rehearse:
        dev = torch.device("cpu")
        if distributed:
            import torch.distributed as dist
            dist.init_process_group("gloo")
        group = _Group(dist, dev)
        elapsed, _ = timed(group, lambda: time.sleep(0.001), args.steps, args.warmup)
        if rank == 0:
            print(json.dumps({"metric": "predicted agents/sec (12-step, batch 128)", "rehearsal": True, "value": None, "unit": "agents/s",
                              "n_gpus": world, "world_size_seen_by_backend": dist.get_world_size() if dist is not None else 1,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                              "higher_is_better": True, "scaling": "weak", "data": "none (launcher rehearsal on CPU, gloo)"}), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    group = _Group(dist, dev)

    from multimodal_transformer_amd import ops
    ops.decoder_select(args.decoder)
    B, ph = args.batch, args.horizon
    hp, model, reg = build_model(args.mode, dev, args.history, ph)
    # every rank gets its own shard of the global synthetic batch (seeded by rank): no collective on the data path
    step, tensors, batch = predict_fn(model, args.mode, dev, B, args.history, ph, seed=1 + rank)
    elapsed, out = timed(group, step, args.steps, args.warmup)
    res = out[1] if args.mode == "lane_cnn" else out[0]
    assert torch.isfinite(res).all()
    ops.decoder_status()
    weak = {"per_gpu_batch": B, "global_batch": B * world, "value": world * B * args.steps / elapsed, "unit": "agents/s",
            "ms_per_step": elapsed / args.steps * 1e3, "scaling": "weak"}

    scalings = None
    if not args.no_scalings:       # every rank takes part
        scalings = {"weak": weak}
        sB = max(1, (B + world - 1) // world)
        if world > 1:
            fn_s, _, _ = predict_fn(model, args.mode, dev, sB, args.history, ph, seed=50 + rank)
            el_s, _ = timed(group, fn_s, args.steps, max(3, args.warmup // 4))
        else:
            el_s = elapsed
        scalings["strong_b%d" % B] = {"per_gpu_batch": sB, "global_batch": sB * world, "value": sB * world * args.steps / el_s,
                                      "unit": "agents/s", "ms_per_step": el_s / args.steps * 1e3, "scaling": "strong",
                                      "note": "the 128-agent batch split N ways: a 12-step decode is latency-bound (flat from 4 to 128 agents), "
                                              "so this does not scale - reported as measured"}
        try:
            lB = 8192 // world
            _, lmodel, _ = build_model("lane_cnn", dev, args.history, ph)
            fn_l, _, _ = predict_fn(lmodel, "lane_cnn", dev, lB, args.history, ph, seed=70 + rank)
            lsteps = max(5, min(20, args.steps))
            el_l, lout = timed(group, fn_l, lsteps, 2)
            assert torch.isfinite(lout[1]).all()
            scalings["lane_cnn_b8192"] = {"per_gpu_batch": lB, "global_batch": lB * world, "value": lB * world * lsteps / el_l,
                                          "unit": "agents/s", "ms_per_step": el_l / lsteps * 1e3, "steps": lsteps, "scaling": "strong",
                                          "decoder_path": ops.decoder_last_path()[1]}
            del lmodel, fn_l, lout
        except Exception as e:       # the headline metric must survive a failure of an extra leg
            scalings["lane_cnn_b8192"] = {"error": f"{type(e).__name__}: {e}"}

    train = None
    if not args.no_train:       # every rank takes part (all-reduce inside)
        try:
            train = bench_train(model, hp, dev, B, args.history, ph, rank, world, args.train_steps,
                                max(3, min(args.warmup, 5)), group, with_map=(args.mode in ("map_cnn", "map_vit")),
                                with_lanes=(args.mode == "lane_cnn"))
        except Exception as e:
            train = {"error": f"{type(e).__name__}: {e}"}
        if train is not None and "error" not in train:
            try:        # the same step as the reference trains it: dropout on
                td = bench_train(model, hp, dev, B, args.history, ph, rank, world, max(5, args.train_steps // 2), 3, group,
                                 with_map=(args.mode in ("map_cnn", "map_vit")), with_lanes=(args.mode == "lane_cnn"), dropout=True)
                train["with_dropout"] = {k: td[k] for k in ("value", "unit", "ms_per_step", "steps", "dropout", "allreduce_ms",
                                                            "last_loss_rank0")}
            except Exception as e:
                train["with_dropout"] = {"error": f"{type(e).__name__}: {e}"}
    train_big = None
    if not args.no_train and not distributed and args.mode == "basic" and B == 128:
        # BASELINE config 2's GLOBAL batch (1024 = 8 x 128) on ONE GPU: the denominator of the 8-GPU data-parallel run
        train_big = {}
        for key, drop in (("dropout_free", False), ("with_dropout", True)):
            try:
                _, bmodel, _ = build_model(args.mode, dev, args.history, ph)
                t = bench_train(bmodel, dict(hp), dev, 1024, args.history, ph, 0, 1, 8, 2, group, dropout=drop)
                train_big[key] = {k: t[k] for k in ("value", "unit", "ms_per_step", "steps", "per_gpu_batch", "agents_per_s",
                                                    "algorithmic_tflops", "last_loss_rank0")}
                del bmodel
            except Exception as e:
                train_big[key] = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        x, x_st, lanes, maps = tensors
        kt = time_kernels(model, reg, tensors, ph, iters=min(50, max(5, args.steps)))
        n_dec = 1 if args.mode != "lane_cnn" or model.lane_dedup else 3
        kr, dec_flops, dec_bytes = kernel_rooflines(args.mode, B, args.history, ph, hp, reg, kt, n_dec)
        achieved = dec_flops / (kt["decoder_ms"] * n_dec * 1e-3) / 1e12
        # HBM-side bytes per decode: only from a PMC pass (tools/pmc_summary.py) over kernels built from THIS source tree
        traffic, traffic_src = None, "no rocprofv3 --pmc summary for this source tree (profiles/r*/traffic_b128_basic.json absent or stale)"
        if args.mode == "basic" and B == 128 and ph == 12 and args.history == 8:
            for rnd in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):   # newest round first
                tf = os.path.join(ROOT, "profiles", rnd, "traffic_b128_basic.json")
                if not os.path.exists(tf):
                    continue
                tj = json.load(open(tf))
                if tj.get("kernel_source_hash") == kernel_source_hash() and tj.get("decoder_form") in kt["decoder_path"]:
                    traffic = tj["decoder_bytes_per_predict"]
                    traffic_src = (f"profiles/{rnd}/traffic_b128_basic.json: rocprofv3 --pmc FETCH_SIZE (x2 on gfx950) + WRITE_SIZE, separate passes "
                                   "over tools/decode_only.py on this source tree; counted at the L2's fabric side, Infinity-Cache hits included")
                    break
        line = {
            "metric": "predicted agents/sec (12-step, batch 128)",
            "value": weak["value"], "unit": "agents/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": weak["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic nuScenes-shaped batch (multimodal_transformer_amd/synthetic.py), "
                                    "random-init weights from the by-name recipe",
            "config": {"workload": f"{args.mode} transformer --autoregressive predict, VEHICLE, batch {B} per GPU, "
                                   f"{args.history}-step history, {ph}-step horizon",
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"batch-split x{world}, no collective"},
            "world_size_rccl": dist.get_world_size() if dist is not None else 1,
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kt["decoder_path"], "launch_unit": "one 12-step decode (all of its launches)",
                         "avg_launch_ms": kt["decoder_ms"], "algorithmic_flop_per_launch": dec_flops,
                         # the HBM view of the same launches (north_star asks for it): weights once + memory + outputs
                         "algorithmic_bytes_per_launch": dec_bytes,
                         "achieved_gbs_vs_hbm": dec_bytes / (kt["decoder_ms"] * 1e-3) / 1e9, "hbm_peak_gbs": HBM_PEAK_GBS,
                         "kernel_source_hash": kernel_source_hash()},
            "kernels_ms": kt, "kernel_rooflines": kr, "scalings": scalings,
        }
        if args.sweep and not distributed:
            sweep = {}
            for bb in (1024, 4096):
                f, _, _ = predict_fn(model, args.mode, dev, bb, args.history, ph, seed=7)
                el, _ = timed(group, f, 10, 3)
                sweep[str(bb)] = {"agents_per_s": bb * 10 / el, "ms_per_step": el / 10 * 1e3,
                                  "mfma_frac": (MFLOP_ENCODER + MFLOP_DECODER + MFLOP_HEAD) * 1e6 * bb * 10 / el / 1e12 / PEAK_F32_MFMA_TFLOPS}
            line["batch_sweep"] = sweep
        if not args.no_configs and not distributed:
            cfgs = {}
            for name, mode, bb in (("map_cnn_b128", "map_cnn", 128), ("map_vit_b128", "map_vit", 128),
                                   ("lane_cnn_b128", "lane_cnn", 128), ("lane_cnn_b1024", "lane_cnn", 1024)):
                try:
                    cfgs[name] = config_report(mode, bb, dev, args.history, ph, 20, group,
                                               train_steps=(10 if bb == 128 and mode != "map_vit" and not args.no_train else 0))
                except Exception as e:
                    cfgs[name] = {"error": f"{type(e).__name__}: {e}"}
            line["configs"] = cfgs
        line["train"] = train
        line["train_b1024"] = train_big
        if not distributed:
            try:
                sp = sync_predict_ms(model, args.mode, dev, B, args.history, ph)
                line["sync_ms_per_step"] = sp["graph"]
                line["sync_ms_per_step_eager"] = sp["eager"]
                line["sync_note"] = ("wall time of predict + .cpu() per call, one after the other (the caller evaluate.py is): "
                                     "'sync_ms_per_step' replays the predict from a HIP graph captured per shape "
                                     "(model.graph_predict = True), '_eager' issues the launches from Python every call")
            except Exception as e:
                line["sync_ms_per_step"] = None
                line["sync_note"] = f"{type(e).__name__}: {e}"
        if not args.no_cpu_baseline and not distributed:
            line["cpu_baseline"] = cpu_baseline(args.mode, hp, batch, ph)
            if train is not None and "error" not in train:
                train["cpu_baseline"] = cpu_train_baseline(hp, batch, ph)
        else:
            line["cpu_baseline"] = None
        line["teardown"] = ("ordered: every Trainer closed (graphs, events, last Work dropped after a device synchronise), barrier, "
                            "synchronise, destroy_process_group, normal interpreter exit - this line is printed first, the "
                            "process's exit code is the teardown's verdict") if dist is not None else "no process group"
        print(json.dumps(line), flush=True)
    if dist is not None:
        # Ordered teardown, exit code from the real exit path (round 2 left through os._exit here).  Every Trainer
        # was closed by bench_train; nothing of this process is in flight on the communicator any more.
        import gc
        del model, reg, step, tensors
        gc.collect()
        dist.barrier()
        torch.cuda.synchronize()
        dist.destroy_process_group()
        sys.stdout.flush()
        sys.stderr.flush()


if __name__ == "__main__":
    main()
