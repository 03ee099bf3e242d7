#!/usr/bin/env python3
"""CT-CLIP contrastive training-step benchmark (BASELINE.json metric: CT-volume-report pairs/s, 480x480x240 bf16).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

A step = CTClipTrainer.train_step on one synthetic batch already resident in HBM: zero_grad, CT-ViT + BERT forward,
similarity matrix + symmetric InfoNCE, backward, RCCL gradient all-reduce (N>1), fused clip(0.5)+Adam, loss.item().
Workload (BASELINE configs[1]): CT-ViT base (dim 512, 4+4 layers, 8x32 heads, codebook 8192) + BERT-base-shaped text
encoder (12x768, the shape of CXR-BERT), 480x480x240 bf16 volumes, 128-token reports, random-init weights.
Rank 0 prints ONE json line (see README / DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "ct-clip-ut_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (guides/MI355X_MICROARCH.md)
VIT = dict(dim=512, codebook_size=8192, image_size=480, patch_size=20, temporal_patch_size=10, spatial_depth=4,
           temporal_depth=4, dim_head=32, heads=8)                      # reference src/train_ctclip.py:19-29
TEXT = dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072, vocab_size=30522,
            max_position_embeddings=512, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)   # BertConfig / CXR-BERT defaults: the
                                                                      # reference trains with them on (model.train())


def build_model(vit_cfg, text_cfg, dim_latent=512):
    from transformers import BertConfig, BertModel
    from models.ctclip import CTCLIP
    from utils.ctvit import CTViT
    torch.manual_seed(0)
    grid = vit_cfg["image_size"] // vit_cfg["patch_size"]
    return CTCLIP(text_encoder=BertModel(BertConfig(**text_cfg)), image_encoder=CTViT(**vit_cfg),
                  dim_text=text_cfg["hidden_size"], dim_image=grid * grid * vit_cfg["dim"], dim_latent=dim_latent)


def synthetic_batch(B, depth, size, L, vocab, device, rank, dtype=torch.bfloat16):
    g = torch.Generator(device=device).manual_seed(1234 + rank)
    vol = (torch.randn(B, 1, depth, size, size, generator=g, device=device) * 0.5).clamp_(-1, 1).to(dtype)
    gt = torch.Generator().manual_seed(4321 + rank)
    ids = torch.randint(0, vocab, (B, L), generator=gt)
    lens = torch.randint(L // 4, L + 1, (B,), generator=gt)
    mask = (torch.arange(L)[None] < lens[:, None]).long()
    txt = {"input_ids": ids.to(device), "token_type_ids": torch.zeros_like(ids).to(device), "attention_mask": mask.to(device)}
    return vol, txt


def gemm_traffic(args):
    """HBM-side bytes per GEMM launch.  bench.py cannot collect PMC counters itself; the figure comes from the committed
    rocprofv3 passes of this same command (profiles/r01_i_hbm_traffic_b64.csv: separate FETCH_SIZE / WRITE_SIZE runs, read
    side doubled as MI355X_MICROARCH.md prescribes for gfx950) and is reported only for the configuration they were taken on."""
    path = os.path.join(ROOT, "profiles", "r01_i_hbm_traffic_b64.csv")
    if args.small or args.batch != 64 or not os.path.exists(path):
        return {"traffic": None}
    launches = total = 0.0
    for line in open(path):
        f = line.rsplit(",", 4)
        if len(f) == 5 and f[0].split("::")[-1].startswith(("gemm3_kernel", "gemm4_kernel", "gemm2_kernel", "gemm_bf16_kernel")):
            launches += float(f[1])
            total += float(f[1]) * float(f[4]) * 1e6
    if launches == 0:
        return {"traffic": None}
    return {"traffic": total / launches, "traffic_unit": "bytes per launch (mean over the family)",
            "traffic_source": "profiles/r01_i_hbm_traffic_b64.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)"}


def cpu_baseline(model, depth, size, L, vocab, reps=3):
    """The oracle (oracle/ctclip_oracle.py: op-for-op f32 restatement of the reference, pinned by golden vectors) timed
    on the host cores.  A full-depth production step takes many minutes on a CPU, so the sample is bounded: every DISTINCT
    stage of the step is run forward + backward for ONE pair at the production shape (480x480x240, L tokens) -- one
    untimed warm-up, then `reps` timed repetitions, median -- and the repeated layers are multiplied out:
    T = patch + 4*spatial + 4*temporal + pos_bias + vq + tail + text(12) + adam.  BASELINE config 1 (4 x 64^3, 32 tokens,
    2+2 / 2 layers, two whole training steps) is timed end to end as well.  Reported next to the GPU number; it is not an
    optimisation target."""
    import statistics
    from oracle import ctclip_oracle as O
    # the GPU box grants a 16-core share of a much larger host: size the pool to the share, not to os.cpu_count()
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(16, avail))
    torch.set_num_threads(cores)
    log = lambda m: print(f"[cpu_baseline] {m}", file=sys.stderr, flush=True)
    log(f"{cores} threads (affinity {avail}, os.cpu_count {os.cpu_count()})")
    st = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    P = "visual_transformer."
    vol, txt = synthetic_batch(1, depth, size, L, vocab, torch.device("cpu"), 0, dtype=torch.float32)

    def req(prefixes):
        out = dict(st)
        for k, v in st.items():
            if v.is_floating_point() and any(k.startswith(p) for p in prefixes) and not k.endswith(".beta"):
                out[k] = v.clone().requires_grad_(True)
        return out

    spent = [0.0]

    def timed(name, fn):
        fn()                                              # warm-up (allocator, thread pool, first-touch)
        ts = []
        for _ in range(reps):
            t0 = time.time()
            fn()
            ts.append(time.time() - t0)
        spent[0] += sum(ts)
        med = statistics.median(ts)
        log(f"{name}: median {med:.2f} s of {[round(t, 2) for t in ts]}")
        return med

    parts = {}
    # patch embedding fwd+bwd
    tok = [None]
    def f_patch():
        s1 = req([P + "to_patch_emb."])
        y = O.patch_embed(vol, s1, P + "to_patch_emb.", VIT["patch_size"], VIT["temporal_patch_size"])
        y.square().mean().backward()
        tok[0] = y.detach()
    parts["patch"] = timed("patch-embed fwd+bwd", f_patch)
    b, t, h, w, d = tok[0].shape
    # position bias MLP (dense (h*w)^2 rows as the reference evaluates it) fwd+bwd
    bias = [None]
    def f_bias():
        s2 = req([P + "spatial_rel_pos_bias."])
        y = O.cpb_bias(h, w, s2, P + "spatial_rel_pos_bias.")
        y.square().mean().backward()
        bias[0] = y.detach()
    parts["pos_bias"] = timed("position-bias MLP fwd+bwd", f_bias)
    # one spatial and one temporal transformer layer fwd+bwd (x1 each; multiplied by depth below)
    def f_spatial():
        s3 = req([P + "enc_spatial_transformer.layers.0.", P + "enc_spatial_transformer.norm_out"])
        xs = tok[0].reshape(b * t, h * w, d).clone().requires_grad_(True)
        O.transformer(xs, s3, P + "enc_spatial_transformer.", 1, VIT["heads"], (b, t, h, w), bias[0]).square().mean().backward()
    parts["spatial_layer"] = timed("1 spatial layer fwd+bwd", f_spatial)
    def f_temporal():
        s4 = req([P + "enc_temporal_transformer.layers.0.", P + "enc_temporal_transformer.norm_out"])
        xt = tok[0].reshape(b * h * w, t, d).clone().requires_grad_(True)
        O.transformer(xt, s4, P + "enc_temporal_transformer.", 1, VIT["heads"], (b, t, h, w), None).square().mean().backward()
    parts["temporal_layer"] = timed("1 temporal layer fwd+bwd", f_temporal)
    # VQ (forward + EMA update; straight-through backward is an identity-cost l2norm)
    q = [None]
    def f_vq():
        xq = tok[0].reshape(b, t * h * w, d).clone().requires_grad_(True)
        out, _, _, _ = O.vq_cosine(xq, st[P + "vq._codebook.embed"], st[P + "vq._codebook.cluster_size"], freeze_codebook=False)
        out.square().mean().backward()
        q[0] = out.detach().reshape(b, t, h, w, d)
    parts["vq"] = timed("VQ fwd(+EMA)+bwd", f_vq)
    # text encoder (all layers) + CLIP tail + loss, fwd+bwd
    def f_tail():
        s5 = req(["text_transformer.", "to_text_latent", "to_visual_latent", "temperature"])
        cls = O.bert_cls(txt["input_ids"], txt["token_type_ids"], txt["attention_mask"], s5, "text_transformer.",
                         TEXT["num_hidden_layers"], TEXT["num_attention_heads"])
        tl, il = O.clip_latents(cls, q[0], s5)
        O.symmetric_info_nce(O.sim_matrix(il, tl, s5["temperature"])).backward()
    parts["text_and_tail"] = timed("text encoder (12 layers) + latents + InfoNCE fwd+bwd", f_tail)
    # clip + Adam over every parameter
    params = [v for k, v in st.items() if v.is_floating_point() and "vq._codebook" not in k and not k.endswith(".beta")]
    def f_adam():
        for p_ in params:
            g = torch.zeros_like(p_)
            O.adam_step(p_.clone(), g, torch.zeros_like(p_), torch.zeros_like(p_), 1, 1.25e-5)
    parts["adam"] = timed(f"Adam over {sum(p_.numel() for p_ in params)/1e6:.1f} M parameters", f_adam)
    total = (parts["patch"] + parts["pos_bias"] + VIT["spatial_depth"] * parts["spatial_layer"]
             + VIT["temporal_depth"] * parts["temporal_layer"] + parts["vq"] + parts["text_and_tail"] + parts["adam"])
    out = {"value": 1.0 / total, "unit": "pairs/s", "cores": cores, "kind": "port",
           "sample": (f"1 pair at 480x480x240 fp32, L={L}: each distinct stage of train_step run fwd+bwd with the oracle, 1 warm-up + "
                      f"{reps} timed repetitions, median ({spent[0]:.1f} s of timed CPU work), layers multiplied out to 4+4+12 -> "
                      f"{total:.1f} s per pair-step"),
           "parts_s": {k: round(v, 3) for k, v in parts.items()}}
    out["config1"] = cpu_config1(O, reps, log)
    return out


def cpu_config1(O, reps, log):
    """BASELINE configs[0]: 4 synthetic 64^3 fp32 volumes + 32-token reports, 2+2-layer CT-ViT / 2-layer text encoder, two
    whole training steps (forward, loss, backward, clip 0.5, Adam) through the oracle on the host cores."""
    import statistics
    cfg = dict(VIT, image_size=64, patch_size=16, temporal_patch_size=16, spatial_depth=2, temporal_depth=2, codebook_size=512)
    tcfg = dict(TEXT, num_hidden_layers=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    model = build_model(cfg, tcfg)
    st0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    frozen = [k for k in st0 if k.endswith(".beta") or "vq._codebook." in k or not st0[k].is_floating_point()]
    data = []
    for s_ in range(2):
        vol, txt = synthetic_batch(4, 64, 64, 32, tcfg["vocab_size"], torch.device("cpu"), s_, dtype=torch.float32)
        data.append((txt, vol))
    ocfg = dict(cfg, text_layers=2, text_heads=tcfg["num_attention_heads"])
    O.train_steps(st0, data, ocfg, lr=1.25e-5, max_grad_norm=0.5, frozen=frozen)      # warm-up
    ts = []
    for _ in range(reps):
        t0 = time.time()
        losses, _, _ = O.train_steps(st0, data, ocfg, lr=1.25e-5, max_grad_norm=0.5, frozen=frozen)
        ts.append((time.time() - t0) / 2)
    med = statistics.median(ts)
    log(f"config 1 (4 x 64^3, 2+2 / 2 layers): median {med * 1e3:.1f} ms per step")
    return {"value": 4.0 / med, "unit": "pairs/s", "ms_per_step": 1e3 * med,
            "sample": f"two full training steps x {reps} repetitions after one warm-up, 4 pairs per step", "loss": losses[-1]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("CTCLIP_BENCH_BATCH", 64)), help="pairs per GPU")
    ap.add_argument("--text-len", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--local-negatives", action="store_true", help="BASELINE config 3: no embedding all-gather")
    ap.add_argument("--small", action="store_true", help="debug: 2+2-layer model on 64^3 volumes")
    args = ap.parse_args()

    from utils.CTClipTrainer import CTClipTrainer
    from ctclip_hip.lib import hip

    vit_cfg, text_cfg, depth, size = dict(VIT), dict(TEXT), 240, 480
    if args.small:
        vit_cfg.update(image_size=64, patch_size=16, temporal_patch_size=16, spatial_depth=2, temporal_depth=2,
                       codebook_size=512)
        text_cfg.update(num_hidden_layers=2)
        depth, size = 64, 64
    model = build_model(vit_cfg, text_cfg)
    model.gather_negatives = not args.local_negatives
    trainer = CTClipTrainer(model, batch_size=args.batch, results_folder=None)
    rt = trainer.accelerator
    world, rank, dev = rt.num_processes, rt.process_index, rt.device
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    vol, txt = synthetic_batch(args.batch, depth, size, args.text_len, text_cfg["vocab_size"], dev, rank)
    batch = (vol, txt)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    loss = None
    for _ in range(args.warmup):
        loss = trainer.train_step(batch)
    # dominant kernel: the bf16 MFMA GEMM family.  algorithmic work of a launch = 2*M*N*K.
    hip.time_kernel("gemm_bf16", lambda A, B, C, bias, resid, M, N, K, *rest: 2.0 * M * N * K)
    hip.time_kernel("gemm_bf16_geglu", lambda A, B, H, G, M, inner, K, *rest: 4.0 * M * inner * K)      # FF1 + fused GEGLU
    hip.time_kernel("gemm_bf16_geglu_bwd", lambda dY, W, H, S, M, inner, K, *rest: 2.0 * M * inner * K)  # FF2 dgrad + GEGLU
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.train_step(batch)          # returns the python float the reference's train_step returns (.item())
    sync()
    dt = time.perf_counter() - t0
    def gemm_timing():
        timed = hip.stop_timing()
        return {k: sum(timed[n][k] for n in ("ctclip_gemm_bf16", "ctclip_gemm_bf16_geglu", "ctclip_gemm_bf16_geglu_bwd"))
                for k in ("launches", "total_ms", "work")}

    def arm():
        hip.time_kernel("gemm_bf16", lambda A, B, C, bias, resid, M, N, K, *rest: 2.0 * M * N * K)
        hip.time_kernel("gemm_bf16_geglu", lambda A, B, H, G, M, inner, K, *rest: 4.0 * M * inner * K)      # FF1 + fused GEGLU
        hip.time_kernel("gemm_bf16_geglu_bwd", lambda dY, W, H, S, M, inner, K, *rest: 2.0 * M * inner * K)  # FF2 dgrad + GEGLU

    timing = gemm_timing()
    # The weight-gradient GEMMs run on a second stream next to the HBM-bound backward kernels, so inside the timed region
    # a GEMM launch shares the chip and its event-to-event duration is longer than the kernel alone.  Two extra, untimed
    # steps with that overlap switched off give the family's stand-alone rate as well.
    from ctclip_hip import ops as _ops
    side_was = _ops._side["on"]
    _ops._side["on"] = False
    arm()
    for _ in range(2):
        trainer.train_step(batch)
    sync()
    alone = gemm_timing()
    _ops._side["on"] = side_was
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    peak_mem = torch.cuda.max_memory_allocated() / 2 ** 30

    if rank == 0:
        pairs = args.batch * world * args.steps
        gemm_tflops = timing["work"] / (timing["total_ms"] * 1e-3) / 1e12 if timing["total_ms"] > 0 else 0.0
        out = {
            "metric": "CT-volume-report pairs/sec (480x480x240 bf16)", "value": pairs / dt, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ("debug-small" if args.small else
                                    "BASELINE configs[1]: CT-ViT base (512d, 4+4 layers, 8x32 heads, cb 8192) + BERT-base-shape "
                                    "text encoder, 480x480x240 bf16 volumes, 128-token reports, text dropout 0.1, full train_step"),
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world, "text_len": args.text_len,
                       "negatives": "local" if args.local_negatives or world == 1 else "global (all-gather)",
                       "parallelism": f"dp{world}", "peak_hbm_gib": round(peak_mem, 1), "final_loss": float(loss)},
            "roofline": {"kernel": "ctclip_gemm_bf16 family (gemm3_kernel 256x256x32 four-stage LDS-DMA ring for the k-major products, "
                                   "gemm4_kernel, the same tile with transposed operands, for weight gradients, gemm2/gemm_bf16 kernels for small grids)", "bound": "mfma", "achieved": gemm_tflops,
                         "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": gemm_tflops / PEAK_BF16_TFLOPS,
                         **gemm_traffic(args), "launches_per_step": timing["launches"] / args.steps,
                         "gemm_ms_per_step": timing["total_ms"] / args.steps,
                         "achieved_without_stream_overlap": alone["work"] / (alone["total_ms"] * 1e-3) / 1e12,
                         "note": "achieved: HIP-event durations inside the timed region, where weight-gradient GEMMs run "
                                 "concurrently on a second stream; *_without_stream_overlap: same launches, two extra "
                                 "untimed steps on one stream"},
        }
        if not args.no_cpu_baseline and world == 1 and not args.small:
            out["cpu_baseline"] = cpu_baseline(model, depth, size, args.text_len, text_cfg["vocab_size"])
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
