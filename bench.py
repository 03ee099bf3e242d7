#!/usr/bin/env python3
"""CT-CLIP contrastive training-step benchmark (BASELINE.json metric: CT-volume-report pairs/s, 480x480x240 bf16).

    python bench.py --gpus N --steps K --warmup W            # N > 1 from a plain shell starts its own N ranks (self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

A step = CTClipTrainer.train_step on one synthetic batch already resident in HBM: zero_grad, CT-ViT + BERT forward,
similarity matrix + symmetric InfoNCE, backward, RCCL gradient all-reduce (N>1), fused clip(0.5)+Adam, loss.item().
Workload (BASELINE configs[1]): CT-ViT base (dim 512, 4+4 layers, 8x32 heads, codebook 8192) + BERT-base-shaped text
encoder (12x768, the shape of CXR-BERT), 480x480x240 bf16 volumes, 128-token reports, random-init weights.
Rank 0 prints ONE json line (see README / DESIGN.md "Measurement").
"""
import argparse
import json, re
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


def pmc_traffic(args):
    """HBM-side bytes per launch from the committed rocprofv3 PMC passes of this same command (separate FETCH_SIZE /
    WRITE_SIZE runs, read side doubled as MI355X_MICROARCH.md prescribes for gfx950).  bench.py cannot collect PMC counters
    itself; the newest profiles/r*_hbm_traffic_b<batch>.csv is used, and only for the per-GPU batch it was taken on.
    -> (family mean for the GEMM kernels, {kernel-name prefix: bytes per launch})"""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_hbm_traffic_b{args.batch}.csv")))
    if args.small or not paths:
        return {"traffic": None}, {}
    path = paths[-1]
    launches = total = everything = 0.0
    per = {}
    steps_profiled = None
    for line in open(path):
        if line.startswith("#"):
            m = re.search(r"--steps (\d+) --warmup (\d+)", line)
            if m:                                  # + the two untimed single-stream steps every run appends
                steps_profiled = int(m.group(1)) + int(m.group(2)) + 2
            continue
        f = line.rsplit(",", 4)
        if len(f) != 5 or f[0].startswith("kernel"):
            continue
        name = f[0].split("::")[-1]
        try:
            per[name] = float(f[4]) * 1e6
        except ValueError:
            continue
        everything += float(f[1]) * float(f[4]) * 1e6
        if name.startswith(GEMM_FAMILY_KERNELS):
            launches += float(f[1])
            total += float(f[1]) * float(f[4]) * 1e6
    rel = os.path.relpath(path, ROOT)
    fam = {"traffic": (total / launches) if launches else None,
           "traffic_unit": "bytes per launch (mean over EVERY launch of the family's kernels in a step, both streams)",
           "traffic_source": f"{rel} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)"}
    if steps_profiled:
        per["__step_total__"] = everything / steps_profiled
        fam["traffic_per_step"] = total / steps_profiled
        fam["traffic_launches_per_step"] = launches / steps_profiled
    return fam, per


# kernels behind ctclip_gemm_bf16 / _geglu / _geglu_bwd / _headmajor / _lnbwd (rocprofv3 names without the namespace): the family
# `roofline` is reported on -- gemm3 (8 waves x 128 x 64) and gemm5 (4 waves x 128 x 128) for the k-major products, gemm4 for the
# weight gradients, gemm2 / gemm_bf16 for small grids
GEMM_FAMILY_KERNELS = ("gemm3_kernel", "gemm5_kernel", "gemm4_kernel", "gemm2_kernel", "gemm_bf16_kernel")


PEAK_HBM_GBPS = 8000.0         # MI355X HBM3E (guides/MI355X_MICROARCH.md; ~6300 achievable by a float4 copy)
FF_INNER, FF_INNER_PAD = 1365, 1408    # int(4 * 2/3 * 512) and its zero-padded width: flops are counted UNPADDED


def _unpad(n):
    return {FF_INNER_PAD: FF_INNER, 2 * FF_INNER_PAD: 2 * FF_INNER}.get(n, n)


def gemm_work_fns():
    """work functions for hip.time_kernel: algorithmic flops (2 M N K on the unpadded dims) and algorithmic HBM bytes
    (operands once + outputs once) of one launch of each entry point of the bf16 MFMA GEMM family."""
    # the q / kv / out projections of the CT-ViT attention blocks (src/utils/attention.py:118-124,140-182) and their gradients, by
    # shape: dim 512, inner 256 (BERT's are 768-wide and never match)
    proj = {(256, 512), (512, 512), (512, 256)}

    def plain(A, B, C, bias, resid, M, N, K, lda, ldb, ldc, ldr, akm, bkm, c_fp32, *rest):
        tag = None
        if akm and bkm and (N, K) in proj and M >= 13824:
            tag = "attn_proj"                      # forward projections and data gradients (same three shapes)
        elif not akm and not bkm and (M, N) in proj and K >= 13824:
            tag = "attn_proj_wgrad"
        return {"flops": 2.0 * _unpad(M) * _unpad(N) * _unpad(K), "tag": tag,
                "bytes": 2.0 * (M * K + N * K) + M * N * (4.0 if c_fp32 else 2.0) + (4.0 * M * N if resid is not None else 0.0)}

    def headmajor(A, B, C, M, N, K, *rest):        # kv projection / out-projection data gradient in the head-major layout
        return {"flops": 2.0 * M * N * K, "tag": "attn_proj", "bytes": 2.0 * (M * K + N * K) + 2.0 * M * N}

    def geglu(A, B, H, G, M, inner, K, *rest):                 # FF1 + fused GEGLU: h [M, 2 inner] and g [M, inner] written
        return {"flops": 4.0 * M * _unpad(inner) * K, "bytes": 2.0 * (M * K + 2 * inner * K) + 2.0 * M * 3 * inner}

    def geglu_bwd(dY, W, H, S, M, inner, K, *rest):            # FF2 dgrad + GEGLU backward: h read, d(h) written in place
        return {"flops": 2.0 * M * _unpad(inner) * K, "bytes": 2.0 * (M * K + inner * K) + 2.0 * M * 4 * inner}
    def lnbwd(A, B, dx, dx16, M, N, K, *rest):       # [rstd dq | dk | dv] [Wqg ; Wkv] with the LayerNorm backward in the epilogue
        return {"flops": 2.0 * M * N * K, "tag": "attn_proj",
                "bytes": 2.0 * (M * K + N * K) + M * N * (4.0 + 2.0 + 4.0) + (2.0 * M * N if dx16 is not None else 0.0)}
    def headnorm(A, B, C, inv, scale, M, N, K, lda, ldb, ldc, n_tok, heads, ncols, *rest):   # q / kv projection + cosine head-norm in the epilogue
        return {"flops": 2.0 * M * N * K, "tag": "attn_proj_hn", "bytes": 2.0 * (M * K + N * K) + 2.0 * M * N + 4.0 * M * ncols / 32}
    return {"gemm_bf16": plain, "gemm_bf16_geglu": geglu, "gemm_bf16_geglu_bwd": geglu_bwd, "gemm_bf16_headmajor": headmajor,
            "gemm_bf16_lnbwd": lnbwd, "gemm_bf16_headnorm": headnorm}


def other_work_fns():
    """the kernels north_star names beside the GEMMs: tubelet patch-embed (HBM), spatial attention (MFMA), VQ search."""
    def patch_fwd(vol, is16, w, b, A, mean, rstd, B, C, Dz, Hy, Wx, tp, p, ldA, *rest):
        # the gather + LayerNorm(4000) kernel ALONE: its algorithmic bytes are the volume read once; the normalised operand
        # it writes ([tokens, 4000] bf16) is traffic the SURVEY 8(d) figure does not grant -- reported next to it
        M = B * (Dz // tp) * (Hy // p) * (Wx // p)
        return {"tag": "tubelet_gather_ln_fwd", "bytes": float(B * C * Dz * Hy * Wx * (2 if is16 else 4)),
                "operand_bytes_written": float(M * ldA * 2)}

    def attn_fwd(q, k, v, o, lse, bias, mask, nseq, n, heads, dp, *rest):
        if n < 256:
            if dp == 32 and nseq >= 576:                        # temporal attention of the CT-ViT (n = 24); BERT's (d_head 64): not timed
                return {"tag": "temporal_attention_fwd", "flops": 4.0 * nseq * heads * n * n * dp, "bytes": 2.0 * 4 * nseq * n * heads * dp}
            return None
        return {"tag": "spatial_attention_fwd", "flops": 4.0 * nseq * heads * n * n * dp,
                "bytes": 2.0 * 4 * nseq * n * heads * dp}

    def attn_bwd(q, k, v, o, do, lse, delta, dq, dk, dv, bias, mask, dbias, rel, dtable, tsize, gh, gw, nseq, n, heads, dp, *rest):
        if n < 256:
            if dp == 32 and nseq >= 576:
                return {"tag": "temporal_attention_bwd", "flops": 10.0 * nseq * heads * n * n * dp, "bytes": 2.0 * 8 * nseq * n * heads * dp}
            return None
        return {"tag": "spatial_attention_bwd", "flops": 10.0 * nseq * heads * n * n * dp,
                "bytes": 2.0 * 8 * nseq * n * heads * dp}

    def attn_hm_fwd(q, k, v, o, lse, bias, shift, nseq, n, heads, *rest):
        return {"tag": "spatial_attention_fwd", "flops": 4.0 * nseq * heads * n * n * 32, "bytes": 2.0 * 4 * nseq * n * heads * 32}

    def attn_hm_bwd(q, k, v, o, do, lse, delta, dq, dk, dv, bias, dbias, rel, dtable, tsize, gh, gw, nseq, n, heads, *rest):
        return {"tag": "spatial_attention_bwd", "flops": 10.0 * nseq * heads * n * n * 32, "bytes": 2.0 * 8 * nseq * n * heads * 32}

    def vq(embed, x, pv, pi, ncodes, M, d, *rest):
        return {"tag": "vq_search", "flops": 2.0 * ncodes * M * d, "bytes": 2.0 * (M * d + ncodes * d) + 8.0 * 16 * M}
    return {"patch_ln_fwd": patch_fwd, "attn_fwd": attn_fwd, "attn_bwd": attn_bwd, "attn_hm_fwd": attn_hm_fwd,
            "attn_hm_bwd": attn_hm_bwd, "vq_topk": vq}


def attention_block_aggregate(timed2, tags, nsteps, vit_cfg, peaks):
    sd, td = vit_cfg["spatial_depth"], vit_cfg["temporal_depth"]
    plain = [(ms, w) for ms, w in timed2["ctclip_gemm_bf16"]["items"] if w.get("tag") == "attn_proj"]
    wgrad = [(ms, w) for ms, w in timed2["ctclip_gemm_bf16"]["items"] if w.get("tag") == "attn_proj_wgrad"]
    hmaj = list(timed2.get("ctclip_gemm_bf16_headmajor", {}).get("items", []))

    def per_step(items):
        n = len(items) // nsteps
        assert n * nsteps == len(items), (len(items), nsteps)
        return [items[i * n:(i + 1) * n] for i in range(nsteps)]

    acc = {k: {"ms": 0.0, "flops": 0.0, "launches": 0} for k in ("spatial_fwd", "temporal_fwd", "spatial_bwd", "temporal_bwd")}

    def add(key, items):
        for ms, w in items:
            acc[key]["ms"] += ms; acc[key]["flops"] += w["flops"]; acc[key]["launches"] += 1
    # launches per step.  ctclip_gemm_bf16_headnorm (round 5: the cosine head-norm in the projection's epilogue): q and kv of every
    # layer, forward only, spatial layers first.  ctclip_gemm_bf16: the out projections (forward: spatial then temporal) and the
    # temporal layers' out-projection data gradient (the spatial ones go through the head-major entry).  ctclip_gemm_bf16_lnbwd: the
    # q + kv data gradients of a layer as ONE launch with the LayerNorm backward in its epilogue, temporal layers first.
    lnb = list(timed2.get("ctclip_gemm_bf16_lnbwd", {}).get("items", []))
    hn = list(timed2.get("ctclip_gemm_bf16_headnorm", {}).get("items", []))
    if hn:
        for st in per_step(hn):
            assert len(st) == 2 * (sd + td), len(st)
            add("spatial_fwd", st[:2 * sd]); add("temporal_fwd", st[2 * sd:])
        for st in per_step(plain):
            assert len(st) == sd + 2 * td, len(st)
            add("spatial_fwd", st[:sd]); add("temporal_fwd", st[sd:sd + td]); add("temporal_bwd", st[sd + td:])
        for st in per_step(hmaj):                   # spatial out-projection data gradient (backward) only
            assert len(st) == sd, len(st)
            add("spatial_bwd", st)
    else:                                           # CTCLIP_HEADNORM_IN_GEMM=0: the round-4 launch pattern
        nf_s, nf_t = 2 * sd, 3 * td
        for st in per_step(plain):
            assert len(st) == nf_s + nf_t + td, len(st)
            add("spatial_fwd", st[:nf_s]); add("temporal_fwd", st[nf_s:nf_s + nf_t])
            add("temporal_bwd", st[nf_s + nf_t:])
        for st in per_step(hmaj):                   # spatial kv projection (forward), spatial out-projection data gradient (backward)
            assert len(st) == 2 * sd, len(st)
            add("spatial_fwd", st[:sd]); add("spatial_bwd", st[sd:])
    for st in per_step(lnb):                        # one per layer, temporal layers first
        assert len(st) == sd + td, len(st)
        add("temporal_bwd", st[:td]); add("spatial_bwd", st[td:])
    for st in per_step(wgrad):                      # three weight gradients per layer, temporal layers first
        assert len(st) == 3 * (sd + td), len(st)
        add("temporal_bwd", st[:3 * td]); add("spatial_bwd", st[3 * td:])
    for tag, key in (("spatial_attention_fwd", "spatial_fwd"), ("spatial_attention_bwd", "spatial_bwd"),
                     ("temporal_attention_fwd", "temporal_fwd"), ("temporal_attention_bwd", "temporal_bwd")):
        d = tags.get(tag)
        assert d is not None, tag
        acc[key]["ms"] += d["ms"]; acc[key]["flops"] += d["flops"]; acc[key]["launches"] += d["launches"]
    out = {"bound": "mfma", "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
           "what": "q + kv + out projections and score / PV products of the CT-ViT attention blocks (SURVEY 8(d): 22.65 GF per spatial "
                   "and 14.84 GF per temporal layer and pair, forward); backward = data + weight gradients of the projections and the "
                   "attention backward at 10 n^2 d flops per head; time = summed HIP-event durations of exactly those launches (the "
                   "backward's q + kv data-gradient launch also applies the block's LayerNorm backward in its epilogue)"}
    tot = {"fwd": [0.0, 0.0], "bwd": [0.0, 0.0]}
    for k, v in acc.items():
        tf = v["flops"] / (v["ms"] * 1e-3) / 1e12
        out[k] = {"achieved": tf, "frac": tf / PEAK_BF16_TFLOPS, "frac_of_measured_mfma": tf / peaks["mfma_bf16_tflops"],
                  "ms_per_step": v["ms"] / nsteps, "gflops_per_step": v["flops"] / nsteps / 1e9, "launches_per_step": v["launches"] / nsteps}
        tot[k[-3:]][0] += v["flops"]; tot[k[-3:]][1] += v["ms"]
    for d_, (fl, ms) in tot.items():
        out[d_] = {"achieved": fl / (ms * 1e-3) / 1e12, "frac": fl / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, "ms_per_step": ms / nsteps}
    fl, ms = tot["fwd"][0] + tot["bwd"][0], tot["fwd"][1] + tot["bwd"][1]
    out["achieved"] = fl / (ms * 1e-3) / 1e12
    out["frac"] = out["achieved"] / PEAK_BF16_TFLOPS
    return out


def patch_embed_chain(model, vol, reps=5):
    """SURVEY 8 row a1 / K1 as ONE unit (reference ctvit.py:44-52: Rearrange + LayerNorm(4000) + Linear(4000, 512) + LayerNorm(512)):
    HIP events around the whole forward chain and around forward + backward, against SURVEY 8(d)'s algorithmic bytes
    (volume + weight + 512-wide output = 128.9 MB per pair at bf16 input) -- NOT the bytes this implementation moves (it
    materialises the normalised [tokens, 4000] operand; `roofline.kernels.tubelet_gather_ln_fwd` shows that kernel alone)."""
    vit = model.visual_transformer
    B, C, Dz, Hy, Wx = vol.shape
    p, tp, dim = vit.patch_size, vit.temporal_patch_size, vit.to_patch_emb[2].weight.shape[0]
    M = B * (Dz // tp) * (Hy // p) * (Wx // p)
    F_ = C * tp * p * p
    alg = float(vol.numel() * vol.element_size() + F_ * dim * 2 + M * dim * 2)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    with torch.no_grad():
        vit.patch_embed(vol)
    fwd, both = [], []
    for _ in range(reps):
        e0, e1 = ev(), ev()
        with torch.no_grad():
            e0.record(); vit.patch_embed(vol); e1.record()
        torch.cuda.synchronize()
        fwd.append(e0.elapsed_time(e1))
    g = None
    for _ in range(reps):
        e0, e1 = ev(), ev()
        e0.record()
        y = vit.patch_embed(vol)
        if g is None:
            g = torch.ones_like(y)
        y.backward(g)
        e1.record()
        torch.cuda.synchronize()
        both.append(e0.elapsed_time(e1))
    model.zero_grad(set_to_none=False)
    f, fb = min(fwd), min(both)
    gbps = alg / (f * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbps / PEAK_HBM_GBPS,
            "ms_forward_chain": f, "ms_forward_plus_backward_chain": fb, "algorithmic_bytes_per_launch": alg,
            "what": "whole K1 forward chain (gather + LN(4000) kernel, K = 4000 MFMA GEMM + folded bias, LN(512)) per call at this "
                    "batch, best of %d, on SURVEY 8(d)'s bytes (volume + weight + 512-wide output); the chain is GEMM-bound: "
                    "2 x tokens x 4000 x 512 flops = %.2f TFLOP" % (reps, 2.0 * M * F_ * dim / 1e12),
            "mfma_tflops_of_the_chain": 2.0 * M * F_ * dim / (f * 1e-3) / 1e12}


def attribution_bench(model, vol, txt, dev, windows=256, occlusion_batch=32, ig_steps=50, ig_batch=10):
    """BASELINE configs[4] (reference src/utils/visualizations.py:335-424,851-910): batched occlusion sensitivity and
    integrated gradients over ONE 480x480x240 volume on this GPU.  Occlusion: the first `windows` of the 12 167 windows of the
    default (20,40,40)/(10,20,20) sweep, `occlusion_batch` per no-grad forward; IG: `ig_steps` interpolation points, forward +
    backward to the volume.  Rate = CT-ViT forwards (or forward+backwards) per second; MFMA fraction on the SURVEY 8(a)
    forward flops of the image tower (789.7 GF per volume)."""
    from utils.visualizations import Visualizations

    class Acc:
        is_main_process, process_index, num_processes, device = True, 0, 1, dev
    image = vol[:1]
    txt1 = {k: v[:1] for k, v in txt.items()}
    was = model.training
    quiet = Visualizations(model, Acc(), occlusion_batch=occlusion_batch, max_windows=windows)
    quiet.maybe_print = lambda *a, **k: None
    quiet.max_windows = occlusion_batch                     # warm-up: one batch
    quiet._compute_occlusion(image, txt1, None, (20, 40, 40), (10, 20, 20), 0.0)
    quiet.max_windows = windows
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    quiet._compute_occlusion(image, txt1, None, (20, 40, 40), (10, 20, 20), 0.0)
    torch.cuda.synchronize()
    occ_s = time.perf_counter() - t0
    quiet._integrated_gradients(image, txt1, steps=ig_batch, ig_batch=ig_batch)      # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    quiet._integrated_gradients(image, txt1, steps=ig_steps, ig_batch=ig_batch)
    torch.cuda.synchronize()
    ig_s = time.perf_counter() - t0
    model.train(was)
    model.zero_grad(set_to_none=False)
    fwd_gf = 789.7e9
    occ_rate = (windows + 1) / occ_s
    return {"workload": "BASELINE configs[4]: one 480x480x240 bf16 volume, 128-token report, production CT-CLIP",
            "occlusion": {"windows": windows, "batch": occlusion_batch, "seconds": occ_s, "value": occ_rate, "unit": "windows/s",
                          "full_sweep_12167_windows_s": 12167.0 / occ_rate,
                          "reference_notebook_rate": "about 10 forwards/s on an unnamed GPU (SURVEY section 6)",
                          "roofline": {"bound": "mfma", "achieved": occ_rate * fwd_gf / 1e12, "peak": PEAK_BF16_TFLOPS,
                                       "unit": "TFLOP/s", "frac": occ_rate * fwd_gf / 1e12 / PEAK_BF16_TFLOPS}},
            "integrated_gradients": {"steps": ig_steps, "batch": ig_batch, "seconds": ig_s, "value": ig_steps / ig_s,
                                     "unit": "points/s (forward + backward to the volume)"}}



def measured_peaks(hip, dev):
    """What this part sustains, next to the datasheet peaks: a register-only MFMA 32x32x16 bf16 loop (no memory traffic) and
    a float4 streaming copy of 2 GiB."""
    out = torch.zeros(1, device=dev)
    blocks, iters = 512, 20000
    hip.probe_mfma(out, blocks, 200)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); hip.probe_mfma(out, blocks, iters); e1.record()
    torch.cuda.synchronize()
    mfma = blocks * 8 * iters * 16 * (2.0 * 32 * 32 * 16) / (e0.elapsed_time(e1) * 1e-3) / 1e12
    nbytes = 1 << 31
    src = torch.empty(nbytes, dtype=torch.uint8, device=dev).random_(0, 255)
    dst = torch.empty_like(src)

    def best_of(fn, reps=3):
        fn()
        best = 1e30
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best
    t_copy = best_of(lambda: hip.probe_copy(src, dst, nbytes))
    t_read = best_of(lambda: hip.probe_stream(src, dst, nbytes, 1))
    t_write = best_of(lambda: hip.probe_stream(src, dst, nbytes, 2))
    del src, dst
    return {"mfma_bf16_tflops": mfma, "hbm_copy_gbps": 2.0 * nbytes / (t_copy * 1e-3) / 1e9,
            "hbm_read_gbps": nbytes / (t_read * 1e-3) / 1e9, "hbm_write_gbps": nbytes / (t_write * 1e-3) / 1e9,
            "how": "ctclip_probe_mfma (512 x 8 waves x 20000 x 16 MFMA 32x32x16, registers only); ctclip_probe_copy / _stream "
                   "(2 GiB, eight independent 16-byte non-temporal accesses in flight per lane; copy counts read + write bytes; "
                   "best of 3)"}


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
    # BASELINE.md section 2 asks for torch.set_num_threads(os.cpu_count()).  The cores this process may run on are its affinity
    # mask (256 on the driver's box, 16 on a shared one) -- but PyTorch's CPU kernels on these shapes get SLOWER beyond a few
    # dozen threads (one spatial layer forward + backward: 28 s on 256 threads of a 256-core host against ~3 s on 32), so
    # the pool size is calibrated: a 3-frame slice of one spatial layer is timed on 16, 32, 64, 128 and all available threads
    # and the fastest count is used and reported (`cores`), with the affinity next to it.
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    log = lambda m: print(f"[cpu_baseline] {m}", file=sys.stderr, flush=True)
    P = "visual_transformer."
    st_cal = {k: v.detach().float().cpu() for k, v in model.state_dict().items() if k.startswith(P + "enc_spatial_transformer.")}
    cal_x = torch.randn(3, 576, VIT["dim"])
    cal_bias = torch.randn(VIT["heads"], 576, 576) * 0.1

    def cal():
        xs = cal_x.clone().requires_grad_(True)
        O.transformer(xs, st_cal, P + "enc_spatial_transformer.", 1, VIT["heads"], (1, 3, 24, 24), cal_bias).square().mean().backward()

    timings = {}
    for nthr in sorted({t for t in (16, 32, 64, 128, avail) if t <= avail} | {min(avail, 8)}):
        torch.set_num_threads(nthr)
        cal()
        t0 = time.time()
        cal()
        timings[nthr] = time.time() - t0
    cores = min(timings, key=timings.get)
    torch.set_num_threads(cores)
    log(f"affinity {avail}, os.cpu_count {os.cpu_count()}; calibration (3-frame spatial layer fwd+bwd, s): "
        + ", ".join(f"{k}: {v:.2f}" for k, v in timings.items()) + f" -> {cores} threads")
    del st_cal
    st = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    vol, txt = synthetic_batch(1, depth, size, L, vocab, torch.device("cpu"), 0, dtype=torch.float32)

    def req(prefixes):
        out = dict(st)
        for k, v in st.items():
            if v.is_floating_point() and any(k.startswith(p) for p in prefixes) and not k.endswith(".beta"):
                out[k] = v.clone().requires_grad_(True)
        return out

    spent = [0.0]

    def timed(name, fn):
        t0 = time.time()
        fn()                                              # warm-up (allocator, thread pool, first-touch)
        warm = time.time() - t0
        ts = []
        # bounded: a stage whose warm-up already took long, or a baseline that has used its budget, is timed once
        for _ in range(reps if (warm < 6.0 and spent[0] < 60.0) else 1):
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
    out = {"value": 1.0 / total, "unit": "pairs/s", "cores": cores, "affinity_cores": avail,
           "thread_calibration_s": {str(k): round(v, 3) for k, v in timings.items()}, "kind": "port",
           "sample": (f"1 pair at 480x480x240 fp32, L={L}: each distinct stage of train_step run fwd+bwd with the oracle, 1 warm-up + "
                      f"up to {reps} timed repetitions, median ({spent[0]:.1f} s of timed CPU work), layers multiplied out to 4+4+12 -> "
                      f"{total:.1f} s per pair-step"),
           "parts_s": {k: round(v, 3) for k, v in parts.items()}}
    # the composition checked against ONE whole production-shape training step through the oracle (forward, InfoNCE, backward, clip,
    # Adam: O.train_steps, what BASELINE.md section 2 describes), two pairs so that the contrastive loss is not degenerate -- bounded:
    # skipped on a host where the stages say it would take more than ~40 s
    if 2.0 * total <= 40.0:
        try:
            vol2, txt2 = synthetic_batch(2, depth, size, L, vocab, torch.device("cpu"), 1, dtype=torch.float32)
            frozen = [k for k in st if k.endswith(".beta") or "vq._codebook." in k or not st[k].is_floating_point()]
            ocfg = dict(VIT, text_layers=TEXT["num_hidden_layers"], text_heads=TEXT["num_attention_heads"])
            t0 = time.time()
            losses, _, _ = O.train_steps(st, [(txt2, vol2)], ocfg, lr=1.25e-5, max_grad_norm=0.5, frozen=frozen)
            e2e = time.time() - t0
            log(f"one whole production step, 2 pairs, end to end: {e2e:.1f} s = {2.0 / e2e:.4f} pairs/s (composed stages: {2.0 * total:.1f} s)")
            out["end_to_end_step"] = {"pairs": 2, "seconds": e2e, "value": 2.0 / e2e, "unit": "pairs/s", "loss": losses[-1],
                                      "composed_seconds_for_the_same": 2.0 * total,
                                      "what": "one whole training step of the oracle at the production shape (no warm-up, no repetition)"}
        except Exception as exc:                              # a host without the memory for it keeps the composed figure
            log(f"end-to-end production step skipped: {exc!r}")
            out["end_to_end_step"] = {"error": repr(exc)}
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


def visible_gpus():
    """GPUs this process would see, WITHOUT the HIP runtime (the parent of self_launch must not initialise it before it starts
    its ranks): KFD topology nodes with a non-zero simd_count, cut by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES /
    CUDA_VISIBLE_DEVICES when one is set.  None when sysfs says nothing (then the ranks themselves report a missing device)."""
    import glob
    n = 0
    paths = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not paths:
        return None
    for p in paths:
        try:
            for line in open(p):
                k, _, v = line.partition(" ")
                if k == "simd_count" and int(v) > 0:
                    n += 1
        except (OSError, ValueError):
            return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            ids = [x for x in v.split(",") if x.strip() != ""]
            n = min(n, len(ids))
    return n


def self_launch(n):
    """`python bench.py --gpus N` from a plain shell (no RANK / WORLD_SIZE): start the N ranks the way the reference gets them
    from `accelerate launch` (src/utils/CTClipTrainer.py:62-69) -- one process per GPU through torch.distributed.run on
    127.0.0.1 -- relay rank 0's JSON line (the children inherit this stdout) and return their exit code.  This parent never
    touches HIP or torch.cuda: the device count for the error message is read from sysfs (visible_gpus), the ranks are fresh
    child processes (subprocess, never an exec of this one)."""
    import socket
    import subprocess
    backend = os.environ.get("CTCLIP_DIST_BACKEND", "nccl")
    have = visible_gpus()
    if backend == "nccl" and have is not None and have < n:
        print(f"bench.py: --gpus {n} but {have} device(s) visible (RCCL needs one device per rank; "
              "CTCLIP_DIST_BACKEND=gloo rehearses several ranks on one)", file=sys.stderr)
        return 2
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool: RCCL needs it
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__), *sys.argv[1:]]
    print(f"[bench] self-launch: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def rendezvous():
    """Ranks started by torch.distributed.run (or by self_launch): bring the process group up BEFORE the model is built, so a
    launch problem shows in seconds; CTClipTrainer's runtime then finds it initialised."""
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1 or dist.is_initialized():
        return
    from datetime import timedelta
    backend = os.environ.get("CTCLIP_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)) % max(1, torch.cuda.device_count()))
    dist.init_process_group(backend=backend, timeout=timedelta(seconds=36000))
    print(f"[bench] rank {dist.get_rank()} of {dist.get_world_size()}: process group up ({backend})", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("CTCLIP_BENCH_BATCH", 96)),
                    help="pairs per GPU (BASELINE configs[1]: sized to the 288 GB of HBM; 96 pairs peak at ~234 GiB; measured "
                         "pairs/s at 88 / 96 / 104: profiles/r04_batch_sweep.txt)")
    ap.add_argument("--text-len", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-attribution", action="store_true", help="skip the BASELINE configs[4] measurement (occlusion + IG)")
    ap.add_argument("--lean", action="store_true", help="profiling runs: only warm-up + timed steps + the two single-stream "
                    "steps (no event-timing A/B, no patch-embed chain, no attribution, no CPU baseline)")
    ap.add_argument("--local-negatives", action="store_true", help="BASELINE config 3: no embedding all-gather")
    ap.add_argument("--small", action="store_true", help="debug: 2+2-layer model on 64^3 volumes")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    # stdout carries exactly ONE line, the JSON: libraries that print to file descriptor 1 (RCCL's version banner at the first
    # collective, gloo's connection messages) are sent to stderr for the whole run and the line goes to the real stdout
    real_stdout = os.fdopen(os.dup(1), "w")
    sys.stdout.flush()
    os.dup2(2, 1)
    rendezvous()

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
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node and --gpus disagree")
    vol, txt = synthetic_batch(args.batch, depth, size, args.text_len, text_cfg["vocab_size"], dev, rank)
    batch = (vol, txt)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    loss = None
    for _ in range(args.warmup):
        loss = trainer.train_step(batch)
    # dominant kernel family: the bf16 MFMA GEMMs.  HIP-event pairs around every launch, on the stream it is launched on,
    # inside the timed region; algorithmic work of a launch = 2 M N K (unpadded) flops and operand + output bytes.
    gemm_fns = gemm_work_fns()

    def arm(fns):
        for name, fn in fns.items():
            hip.time_kernel(name, fn)

    main_sid = str(torch.cuda.current_stream().cuda_stream)

    def family(timed, only_main=False):
        # only_main: launches on the step's own stream.  With the text tower on a second stream (ops.fork_text_stream) its ~150
        # small GEMMs run INSIDE the image tower's kernels; an event pair around one of them spans whatever it waited behind, so
        # their durations say nothing about the kernel -- they are counted in `text_stream` below, not in the family's rate
        names = ["ctclip_" + n for n in gemm_fns]
        items = [it for n in names for it in timed[n]["items"] if not only_main or it[1].get("stream") == main_sid]
        ms = sum(m for m, _ in items)
        flops = sum(w["flops"] for _, w in items)
        # per-shape bound: a launch can run no faster than max(flops / MFMA peak, bytes / HBM peak)
        bound = sum(max(w["flops"] / (PEAK_BF16_TFLOPS * 1e12), w["bytes"] / (PEAK_HBM_GBPS * 1e9)) for _, w in items) * 1e3
        mfma_only = flops / (PEAK_BF16_TFLOPS * 1e12) * 1e3
        hbm_bound_ms = sum(m for m, w in items if w["bytes"] / (PEAK_HBM_GBPS * 1e9) > w["flops"] / (PEAK_BF16_TFLOPS * 1e12))
        return {"launches": len(items), "total_ms": ms, "flops": flops, "bytes": sum(w["bytes"] for _, w in items),
                "bound_ms": bound, "mfma_only_ms": mfma_only, "ms_in_hbm_bound_launches": hbm_bound_ms}

    arm(gemm_fns)
    sync()
    step_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        loss = trainer.train_step(batch)          # returns the python float the reference's train_step returns (.item())
        step_ms.append(1e3 * (time.perf_counter() - ts))    # .item() is a host sync, so this is the step's wall time
    sync()
    dt = time.perf_counter() - t0
    timed1 = hip.stop_timing()
    from ctclip_hip import ops as _ops0
    text_on = bool(_ops0._text_stream["on"]) and dev.type == "cuda"
    timing = family(timed1, only_main=text_on)
    timing_all = family(timed1)
    # The weight-gradient GEMMs run on a second stream next to the HBM-bound backward kernels, so inside the timed region
    # a GEMM launch shares the chip and its event-to-event duration is longer than the kernel alone.  Two extra, untimed
    # steps with that overlap switched off give the family's stand-alone rate, and time the other kernels north_star names
    # (tubelet patch-embed, spatial attention, VQ search) undisturbed.
    from ctclip_hip import ops as _ops
    text_was = _ops._text_stream["on"]
    _ops._text_stream["on"] = False
    arm(gemm_fns)
    arm(other_work_fns())
    extra = 2
    for _ in range(extra):
        trainer.train_step(batch)
    sync()
    timed2 = hip.stop_timing()
    alone = family(timed2)
    _ops._text_stream["on"] = text_was
    # what the HIP-event pairs around the GEMM launches cost the timed region: the same steps once more with no timing armed
    sync()
    t1 = time.perf_counter()
    for _ in range(0 if args.lean else args.steps):
        trainer.train_step(batch)
    sync()
    dt_plain = (time.perf_counter() - t1) if not args.lean else dt
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    peak_mem = torch.cuda.max_memory_allocated() / 2 ** 30

    if rank == 0:
        import statistics
        pairs = args.batch * world * args.steps
        fam_traffic, per_kernel_traffic = pmc_traffic(args)
        peaks = measured_peaks(hip, dev)
        # counter traffic and algorithmic bytes over the SAME launch set: every launch of the family's kernels in one step, on
        # both streams (the PMC passes cannot tell streams apart) -- per step, and their ratio
        traffic_vs_alg = {"algorithmic_bytes_per_step_all_launches": timing_all["bytes"] / args.steps,
                          "launches_per_step_all": timing_all["launches"] / args.steps}
        if fam_traffic.get("traffic_per_step"):
            traffic_vs_alg["traffic_over_algorithmic"] = fam_traffic["traffic_per_step"] / (timing_all["bytes"] / args.steps)
        gemm_tflops = timing["flops"] / (timing["total_ms"] * 1e-3) / 1e12 if timing["total_ms"] > 0 else 0.0
        alone_tflops = alone["flops"] / (alone["total_ms"] * 1e-3) / 1e12 if alone["total_ms"] > 0 else 0.0
        kernels = {}
        tags = {}
        for n in ("ctclip_patch_ln_fwd", "ctclip_attn_fwd", "ctclip_attn_bwd", "ctclip_attn_hm_fwd", "ctclip_attn_hm_bwd", "ctclip_vq_topk"):
            for ms, w in timed2.get(n, {}).get("items", []):
                d = tags.setdefault(w["tag"], {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0, "extra_written": 0.0})
                d["ms"] += ms; d["flops"] += w.get("flops", 0.0); d["bytes"] += w["bytes"]; d["launches"] += 1
                d["extra_written"] += w.get("operand_bytes_written", 0.0)
        pmc_names = {"tubelet_gather_ln_fwd": ("patch_ln_fwd_fast",),
                     "spatial_attention_fwd": ("hm_fwd_kernel<true, true",), "vq_search": ("vq_topk3_kernel",),
                     "spatial_attention_bwd": ("hm_bwd_dq_kernel", "hm_bwd_dkv_kernel")}
        for tag, d in tags.items():
            if d["ms"] <= 0:
                continue
            ent = {"launches_per_step": d["launches"] / extra, "ms_per_step": d["ms"] / extra}
            # which roof: flops / MFMA peak against algorithmic bytes / HBM peak.  The temporal attention (24 tokens per sequence:
            # 12 FLOP per byte of q, k, v, o) sits two decades under the 312 FLOP/B ridge: it is an HBM kernel, like the tubelet gather
            # (the spatial attention, 288 FLOP/B, stays on the MFMA roof it has been reported against since round 1)
            hbm_bound = tag.startswith("temporal") and d["bytes"] / (PEAK_HBM_GBPS * 1e9) > d["flops"] / (PEAK_BF16_TFLOPS * 1e12)
            if tag.startswith("tubelet") or hbm_bound:
                gbps = d["bytes"] / (d["ms"] * 1e-3) / 1e9
                ent.update(bound="hbm", achieved=gbps, peak=PEAK_HBM_GBPS, unit="GB/s", frac=gbps / PEAK_HBM_GBPS,
                           frac_of_measured_copy=gbps / peaks["hbm_copy_gbps"], algorithmic_bytes_per_launch=d["bytes"] / d["launches"],
                           operand_bytes_written_per_launch=d["extra_written"] / d["launches"],
                           achieved_incl_operand_write=(d["bytes"] + d["extra_written"]) / (d["ms"] * 1e-3) / 1e9)
            else:
                tf = d["flops"] / (d["ms"] * 1e-3) / 1e12
                ent.update(bound="mfma", achieved=tf, peak=PEAK_BF16_TFLOPS, unit="TFLOP/s", frac=tf / PEAK_BF16_TFLOPS,
                           frac_of_measured_mfma=tf / peaks["mfma_bf16_tflops"], algorithmic_bytes_per_launch=d["bytes"] / d["launches"])
            hits = [[v for k, v in per_kernel_traffic.items() if k.startswith(pref)] for pref in pmc_names.get(tag, ())]
            ent["traffic"] = sum(h[0] for h in hits) if hits and all(hits) else None      # a call = one launch of each kernel named
            kernels[tag] = ent
        # SURVEY 8(d)'s "attention-block GEMMs" (K4 + K5 + K6: the q, kv and out projections AND the score / PV products,
        # src/utils/attention.py:118-124,140-182), the number north_star's 40 % target is defined on: flops of all of them / the
        # summed HIP-event time of their launches, forward and backward, spatial and temporal transformer.  The projections are
        # told from the other GEMMs by shape, forward from backward and spatial from temporal by launch order inside a step (the
        # forward runs four spatial then four temporal layers, the backward the reverse).
        try:
            kernels["attention_block_gemms"] = attention_block_aggregate(timed2, tags, extra, vit_cfg, peaks)
        except Exception as exc:                                  # a changed launch pattern must not take the benchmark line down
            kernels["attention_block_gemms"] = {"error": repr(exc)}
        if world == 1 and not args.lean:
            kernels["patch_embed_fwd"] = patch_embed_chain(model, vol)
        # the whole step against HBM: counter traffic of every kernel of one step (same committed PMC passes) / the copy rate this
        # box sustains -- the step moves ~1.1 TB, so this, not the MFMA peak, is the bound the step as a whole runs against
        step_hbm = None
        if per_kernel_traffic.get("__step_total__") and world == 1:
            sb = per_kernel_traffic["__step_total__"]
            ms_copy = sb / (peaks["hbm_copy_gbps"] * 1e9) * 1e3
            step_hbm = {"bytes_per_step": sb, "ms_at_8_TBps": sb / (PEAK_HBM_GBPS * 1e9) * 1e3, "ms_at_measured_copy_rate": ms_copy,
                        "frac_of_measured_copy_bound": ms_copy / statistics.median(step_ms),
                        "what": "sum over all kernels of launches x (2 x FETCH_SIZE + WRITE_SIZE) per step; bound = that / measured float4 copy rate"}
        out = {
            "metric": "CT-volume-report pairs/sec (480x480x240 bf16)", "value": pairs / dt, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "ms_per_step_median": statistics.median(step_ms), "value_at_median_step": args.batch * world / (statistics.median(step_ms) * 1e-3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ("debug-small" if args.small else
                                    "BASELINE configs[1]: CT-ViT base (512d, 4+4 layers, 8x32 heads, cb 8192) + BERT-base-shape "
                                    "text encoder, 480x480x240 bf16 volumes, 128-token reports, text dropout 0.1, full train_step"),
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world, "text_len": args.text_len,
                       "negatives": "local" if args.local_negatives or world == 1 else "global (all-gather)",
                       "parallelism": f"dp{world}", "peak_hbm_gib": round(peak_mem, 1), "final_loss": float(loss)},
            "roofline": {"kernel": "ctclip_gemm_bf16 family (gemm3_kernel: 8 waves x 128 x 64, the k-major products incl. FF2 dgrad + GEGLU', the "
                                   "head-major and LayerNorm-backward epilogues; gemm5_kernel: one wave per SIMD, 4 x 128 x 128, FF1 + GEGLU and "
                                   "N >= 2048 and, evenly spaced requests, every plain K >= 1024 product; gemm4_kernel: the gemm3 tile with transposed operands, weight gradients; gemm2 / gemm_bf16 "
                                   "kernels for small grids)",
                         "bound": "mfma", "achieved": gemm_tflops,
                         "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": gemm_tflops / PEAK_BF16_TFLOPS,
                         **fam_traffic, **traffic_vs_alg, "launches_per_step": timing["launches"] / args.steps,
                         "gemm_ms_per_step": timing["total_ms"] / args.steps,
                         "algorithmic_flops_per_step": timing["flops"] / args.steps,
                         "algorithmic_bytes_per_launch": timing_all["bytes"] / max(1, timing_all["launches"]),
                         "algorithmic_bytes_per_launch_main_stream": timing["bytes"] / max(1, timing["launches"]),
                         "achieved_without_stream_overlap": alone_tflops,
                         "frac_without_stream_overlap": alone_tflops / PEAK_BF16_TFLOPS,
                         "gemm_ms_per_step_without_stream_overlap": alone["total_ms"] / extra,
                         "per_shape_bound": {
                             "what": "sum over launches of max(flops / 2.5 PFLOP/s, algorithmic bytes / 8 TB/s): several of the step's "
                                     "products (out-projection, FF2 with the f32 residual, K = 256/512 data gradients) sit below the "
                                     "312 FLOP/B ridge and are HBM-bound, so the MFMA-only bound overstates what the family can reach",
                             "bound_ms_per_step": alone["bound_ms"] / extra, "mfma_only_ms_per_step": alone["mfma_only_ms"] / extra,
                             "measured_ms_per_step": alone["total_ms"] / extra,
                             "frac_of_bound": alone["bound_ms"] / alone["total_ms"] if alone["total_ms"] else None,
                             "share_of_time_in_hbm_bound_launches": alone["ms_in_hbm_bound_launches"] / alone["total_ms"] if alone["total_ms"] else None},
                         "measured_peaks": peaks,
                         "step_hbm": step_hbm,
                         "frac_of_measured_mfma": gemm_tflops / peaks["mfma_bf16_tflops"],
                         "frac_of_measured_mfma_without_stream_overlap": alone_tflops / peaks["mfma_bf16_tflops"],
                         "kernels": kernels,
                         "text_stream": {"on": bool(text_was),
                                         "launches_per_step": (timing_all["launches"] - timing["launches"]) / args.steps,
                                         "gflops_per_step": (timing_all["flops"] - timing["flops"]) / args.steps / 1e9,
                                         "event_ms_per_step": (timing_all["total_ms"] - timing["total_ms"]) / args.steps,
                                         "what": "GEMM launches of the text tower, issued on a second stream next to the image tower "
                                                 "(ops.fork_text_stream; CTCLIP_TEXT_STREAM=0 keeps one stream): their event pairs span "
                                                 "the image-tower kernels they run inside, so they are left out of `achieved`; "
                                                 "`*_without_stream_overlap` and `kernels` come from two extra steps on one stream"},
                         "event_timing": {"ms_per_step_with_event_pairs": 1e3 * dt / args.steps,
                                          "ms_per_step_without": 1e3 * dt_plain / args.steps,
                                          "overhead_frac": (dt - dt_plain) / dt_plain,
                                          "what": "the timed region records a HIP-event pair around each of the step's ~294 GEMM "
                                                  "launches; the same steps re-run with nothing armed show what that costs"},
                         "note": "achieved: HIP-event durations inside the timed region of the launches on the step's own stream (the text tower's, "
                                 "on their own stream, are in text_stream); *_without_stream_overlap and `kernels`: every launch, two extra "
                                 "untimed steps on one stream; flops counted on the unpadded GEGLU width 1365"},
        }
        if not args.no_attribution and not args.lean and world == 1 and not args.small:
            del batch
            torch.cuda.empty_cache()
            out["attribution"] = attribution_bench(model, vol, txt, dev)
        if not args.no_cpu_baseline and not args.lean and world == 1 and not args.small:
            out["cpu_baseline"] = cpu_baseline(model, depth, size, args.text_len, text_cfg["vocab_size"])
        real_stdout.write(json.dumps(out) + "\n")
        real_stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
