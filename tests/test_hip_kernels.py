"""GPU unit tests of the C-ABI kernels against plain PyTorch fp32 math on the same (bf16-rounded) inputs.

Tolerances: kernels take bf16 operands and accumulate in f32; bf16 outputs carry 2^-9 relative rounding,
so bf16 results are compared at rtol 2e-2 / f32 results at 2e-3 against an fp32 reference computed from
the SAME bf16-rounded inputs.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from ctclip_hip.lib import hip as h
    assert torch.cuda.is_available()
    return h


DEV = "cuda"


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def bf(x):
    return x.to(torch.bfloat16).contiguous()


def relerr(a, b):
    a, b = a.float(), b.float()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def check(name, got, ref, tol):
    e = relerr(got, ref)
    print(f"  {name}: max-rel-to-peak err {e:.3e} (tol {tol:.1e})")
    assert math.isfinite(e) and e <= tol, f"{name}: {e} > {tol}"


# ---------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("akm,bkm", [(1, 1), (1, 0), (0, 0), (0, 1)])
@pytest.mark.parametrize("M,N,K", [(200, 136, 72), (512, 256, 512), (96, 2816, 64), (128, 128, 4032)])
def test_gemm_layouts(hip, akm, bkm, M, N, K):
    if not akm and M % 8:
        M = (M + 7) // 8 * 8
    A = bf(rnd(M, K, seed=1)) if akm else bf(rnd(K, M, seed=1))
    B = bf(rnd(N, K, seed=2)) if bkm else bf(rnd(K, N, seed=2))
    Am = A.float() if akm else A.float().t()
    Bm = B.float().t() if bkm else B.float()
    ref = Am @ Bm
    C = torch.empty(M, N, device=DEV, dtype=torch.float32)
    hip.gemm_bf16(A, B, C, None, None, M, N, K, A.stride(0), B.stride(0), N, 0, akm, bkm, 1, 1, 0, 1.0, 0)
    check(f"gemm f32 ({akm},{bkm}) {M}x{N}x{K}", C, ref, 2e-3)
    C16 = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    hip.gemm_bf16(A, B, C16, None, None, M, N, K, A.stride(0), B.stride(0), N, 0, akm, bkm, 0, 1, 0, 1.0, 0)
    check(f"gemm bf16 ({akm},{bkm})", C16, ref, 1e-2)


@pytest.mark.parametrize("akm,bkm", [(1, 1), (1, 0), (0, 0), (0, 1)])
@pytest.mark.parametrize("M,N,K,split", [(4096, 1536, 256, 1), (4000, 1496, 192, 1), (520, 392, 8192, 16), (3000, 640, 64, 1),
                                         (20000, 1408, 512, 1),   # 869 tiles > CUs: persistent walk, ragged last row tile
                                         (16384, 2048, 128, 1),   # persistent, two K-steps per tile, all tiles interior
                                         (9000, 1104, 64, 1)])    # persistent, one K-step per tile, ragged in both directions
def test_gemm_pipelined_variant(hip, akm, bkm, M, N, K, split):
    """Shapes large enough (and K % 64 == 0) to dispatch to the LDS-DMA pipelined kernel (csrc/gemm2.hip), including
    ragged M/N tiles, the one-k-tile case, and split-K accumulation.  CTCLIP_GEMM_V2_ALL is set by conftest for the gpu
    session so the gate is wide open here; the (1,1) long-K case goes there by default."""
    A = bf(rnd(M, K, seed=1)) if akm else bf(rnd(K, M, seed=1))
    B = bf(rnd(N, K, seed=2)) if bkm else bf(rnd(K, N, seed=2))
    ref = (A.float() if akm else A.float().t()) @ (B.float().t() if bkm else B.float())
    bias, res = rnd(N, seed=5), rnd(M, N, seed=6)
    if split > 1:
        C = res.clone()
        hip.gemm_bf16(A, B, C, bias, None, M, N, K, A.stride(0), B.stride(0), N, 0, akm, bkm, 1, split, 1, 1.0, 0)
        check(f"gemm2 split-k ({akm},{bkm})", C, ref + bias + res, 2e-3)
        return
    C = torch.empty(M, N, device=DEV)
    hip.gemm_bf16(A, B, C, bias, res, M, N, K, A.stride(0), B.stride(0), N, N, akm, bkm, 1, 1, 0, 1.0, 0)
    check(f"gemm2 f32+bias+res ({akm},{bkm}) {M}x{N}x{K}", C, ref + bias + res, 2e-3)
    C16 = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    hip.gemm_bf16(A, B, C16, None, None, M, N, K, A.stride(0), B.stride(0), N, 0, akm, bkm, 0, 1, 0, 1.0, 1)
    check(f"gemm2 bf16 gelu ({akm},{bkm})", C16, torch.nn.functional.gelu(ref), 1e-2)


@pytest.mark.parametrize("M,N,K,split", [(512, 768, 4096, 5),     # interior tiles, uneven last split
                                         (2816, 512, 2048, 3),    # FF1 weight-gradient shape (11 x 2 tiles)
                                         (520, 1408, 1024, 4),    # ragged in both directions (FF2: 5.5 tiles)
                                         (304, 200, 32 * 23, 2),  # splits of 12 and 11 K-steps
                                         (256, 264, 96, 2),       # fewer K-steps (3) than ring stages
                                         (72, 40, 32, 1)])        # one K-step, one partial tile
def test_gemm_weight_gradient_kernel(hip, M, N, K, split):
    """dW += alpha dy^T x with both operands row-major over the tokens (torch.nn.Linear backward): the 256x256x32
    transposed-operand kernel (csrc/gemm4.hip) with split-K through the workspace / atomics into a running sum."""
    A, B = bf(rnd(K, M, seed=1)), bf(rnd(K, N, seed=2))
    ref = A.float().t() @ B.float()
    run = rnd(M, N, seed=6)
    C = run.clone()
    hip.gemm_bf16(A, B, C, None, None, M, N, K, M, N, N, 0, 0, 0, 1, split, 1, 0.5, 0)
    check(f"wgrad {M}x{N}x{K} split {split}", C, 0.5 * ref + run, 2e-3)
    # strided operands (column slices of wider buffers, as the fused q/kv gradients are)
    Aw, Bw = bf(rnd(K, M + 64, seed=3)), bf(rnd(K, N + 128, seed=4))
    C2 = torch.zeros(M, N + 8, device=DEV)
    hip.gemm_bf16(Aw[:, 64:], Bw[:, 128:], C2, None, None, M, N, K, M + 64, N + 128, N + 8, 0, 0, 0, 1, split, 1, 1.0, 0)
    check("wgrad strided", C2[:, :N], Aw[:, 64:].float().t() @ Bw[:, 128:].float(), 2e-3)
    assert float(C2[:, N:].abs().max()) == 0.0


@pytest.mark.parametrize("M,inner,K", [(3000, 384, 96), (2100, 128, 64), (70, 64, 40)])
def test_linear_geglu_fused_and_blocked_backward(hip, M, inner, K):
    """ctclip_gemm_bf16_geglu (reference attention.py:38-50): interleaved [val 32 | gate 32] weight rows, H and
    G = gelu(gate) * value from one pass (fused epilogue for the first two shapes, product + blocked GEGLU for the last),
    and the blocked GEGLU backward."""
    x = bf(rnd(M, K, seed=70))
    w = bf(rnd(2 * inner, K, seed=71) * 0.2)                              # reference layout: value rows then gate rows
    wi = torch.stack((w[:inner].view(inner // 32, 32, K), w[inner:].view(inner // 32, 32, K)), 1).reshape(2 * inner, K).contiguous()
    href = x.float() @ w.float().t()
    val, gate = href[:, :inner].clone().requires_grad_(True), href[:, inner:].clone().requires_grad_(True)
    gref = torch.nn.functional.gelu(gate) * val
    H = torch.empty(M, 2 * inner, device=DEV, dtype=torch.bfloat16)
    G = torch.empty(M, inner, device=DEV, dtype=torch.bfloat16)
    hip.gemm_bf16_geglu(x, wi, H, G, M, inner, K, K, K, 2 * inner, inner)
    Hb = H.float().view(M, inner // 32, 2, 32)
    check("fused h value", Hb[:, :, 0].reshape(M, inner), href[:, :inner], 1e-2)
    check("fused h gate", Hb[:, :, 1].reshape(M, inner), href[:, inner:], 1e-2)
    check("fused g", G, gref, 1.5e-2)
    dg = bf(rnd(M, inner, seed=72))
    gref.backward(dg.float())
    dH = torch.empty_like(H)
    hip.geglu_bwd(dg, H, dH, M, inner, 32, inner, 2 * inner)
    dHb = dH.float().view(M, inner // 32, 2, 32)
    check("blocked geglu dval", dHb[:, :, 0].reshape(M, inner), val.grad, 2e-2)
    check("blocked geglu dgate", dHb[:, :, 1].reshape(M, inner), gate.grad, 2e-2)
    # dg = dY W2 + GEGLU backward in one pass, in place over H (ctclip_gemm_bf16_geglu_bwd)
    Kd = 96
    dY = bf(rnd(M, Kd, seed=73))
    w2 = bf(rnd(Kd, inner, seed=74) * 0.2)                                 # Linear(inner, Kd).weight
    w2T = w2.t().contiguous()                                              # [inner, Kd]: the k-major operand of dg = dY w2
    dg_ref = bf(dY.float() @ w2.float())                                   # what the unfused path rounds to bf16
    val.grad = None
    gate.grad = None
    (torch.nn.functional.gelu(gate) * val).backward(dg_ref.float())
    H2 = H.clone()
    scratch = torch.empty(M, inner, device=DEV, dtype=torch.bfloat16)
    hip.gemm_bf16_geglu_bwd(dY, w2T, H2, scratch, M, inner, Kd, Kd, Kd, 2 * inner, inner)
    H2b = H2.float().view(M, inner // 32, 2, 32)
    check("fused dgrad+geglu dval", H2b[:, :, 0].reshape(M, inner), val.grad, 2.5e-2)
    check("fused dgrad+geglu dgate", H2b[:, :, 1].reshape(M, inner), gate.grad, 2.5e-2)


@pytest.mark.parametrize("M,N,K", [(20000, 1408, 1024),    # 474 tiles > CUs: several tiles per workgroup (flattened ring), ragged last row tile
                                   (4096, 512, 1408),      # FF2 forward shape (K = 44 K-steps)
                                   (1000, 264, 2816),      # ragged in both directions, N % 256 != 0 (FF1 data-gradient K)
                                   (3072, 768, 1024)])     # all tiles interior
def test_gemm_one_wave_per_simd_tile(hip, M, N, K):
    """k-major x k-major products with K % 64 == 0 and K >= 1024 go to csrc/gemm5.hip under the suite's size gates (4 waves x 128 x 128, accumulators in
    AccVGPRs, flattened (tile, K-step) ring): every epilogue form of gemm_tile.h against f32 torch math on the same bf16
    operands -- f32 + bias + residual + alpha, bf16 + bias + erf-GELU, head-major bf16, FF1 + GEGLU, FF2 dgrad + GEGLU
    backward (reference attention.py:38-51,118-124)."""
    A, B = bf(rnd(M, K, seed=11) * 0.5), bf(rnd(N, K, seed=12) * 0.5)
    ref = A.float() @ B.float().t()
    bias, res = rnd(N, seed=5), rnd(M, N, seed=6)
    C = torch.empty(M, N, device=DEV)
    hip.gemm_bf16(A, B, C, bias, res, M, N, K, K, K, N, N, 1, 1, 1, 1, 0, 0.5, 0)
    check(f"gemm5 f32 + bias + resid {M}x{N}x{K}", C, 0.5 * ref + bias + res, 2e-3)
    C16 = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    hip.gemm_bf16(A, B, C16, bias, None, M, N, K, K, K, N, 0, 1, 1, 0, 1, 0, 1.0, 1)
    check("gemm5 bf16 + bias + gelu", C16, torch.nn.functional.gelu(ref + bias), 1e-2)
    wide = torch.zeros(M, N + 40, device=DEV, dtype=torch.bfloat16)                      # ldc > N: the columns beyond stay untouched
    hip.gemm_bf16(A, B, wide, None, None, M, N, K, K, K, N + 40, 0, 1, 1, 0, 1, 0, 1.0, 0)
    check("gemm5 bf16 ldc > N", wide[:, :N], ref, 1e-2)
    assert float(wide[:, N:].float().abs().max()) == 0.0
    if N % 64 == 0 and M % 8 == 0:
        # head-major: [part][sequence][head][token][32]
        n_tok = 8
        heads = N // 32 // 2 if (N // 32) % 2 == 0 else N // 32
        parts = N // (32 * heads)
        hm = torch.empty(parts, M // n_tok, heads, n_tok, 32, device=DEV, dtype=torch.bfloat16)
        hip.gemm_bf16_headmajor(A, B, hm, M, N, K, K, K, n_tok, heads)
        want = ref.view(M // n_tok, n_tok, parts, heads, 32).permute(2, 0, 3, 1, 4)
        check("gemm5 head-major", hm, want, 1e-2)
    if N % 128 == 0:
        inner = N // 2                                                                    # B rows as interleaved [val 32 | gate 32] blocks
        Hh = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        G = torch.empty(M, inner, device=DEV, dtype=torch.bfloat16)
        hip.gemm_bf16_geglu(A, B, Hh, G, M, inner, K, K, K, N, inner)
        hb = ref.view(M, inner // 32, 2, 32)
        check("gemm5 fused h", Hh, ref, 1e-2)
        check("gemm5 fused g", G, (torch.nn.functional.gelu(hb[:, :, 1]) * hb[:, :, 0]).reshape(M, inner), 1.5e-2)
        # dg = dY W2T^T with the GEGLU backward in place over h: here dY = A [M, K], W2T = B[:inner] ([inner, K])
        h0 = bf(rnd(M, 2 * inner, seed=13))
        h2 = h0.clone()
        hip.gemm_bf16_geglu_bwd(A, B[:inner], h2, None, M, inner, K, K, K, 2 * inner, inner)
        dg = bf(A.float() @ B[:inner].float().t()).float()
        hv = h0.float().view(M, inner // 32, 2, 32)
        val = hv[:, :, 0].reshape(M, inner).clone().requires_grad_(True)
        gate = hv[:, :, 1].reshape(M, inner).clone().requires_grad_(True)
        (torch.nn.functional.gelu(gate) * val).backward(dg)
        got = h2.float().view(M, inner // 32, 2, 32)
        check("gemm5 dgrad + geglu dval", got[:, :, 0].reshape(M, inner), val.grad, 2.5e-2)
        check("gemm5 dgrad + geglu dgate", got[:, :, 1].reshape(M, inner), gate.grad, 2.5e-2)


@pytest.mark.parametrize("M,N,K", [(70000, 256, 192),      # nk = 6: the ring fill nearly spans a tile; 274 tiles > CUs: the flattened ring crosses
                                   #                         tiles every six steps, ragged last row tile (C_ST / `post` wait path)
                                   (66000, 512, 320),      # nk = 10, 516 tiles
                                   (3000, 2816, 512),      # production FF1 + GEGLU: K = 512, N = 2 x 1408
                                   (2048, 3072, 768),      # BERT intermediate: K = 768, N = 3072
                                   (1000, 264, 960)])      # ragged both ways, N % 256 != 0, the longest "short" ring
def test_gemm_one_wave_per_simd_short_rings(hip, M, N, K):
    """csrc/gemm5.hip through its direct entry (ctclip_gemm5_bf16) on the K = 192 .. 960 rings the dispatcher's size gates keep
    away from it in the suite but a training run sends there (FF1 + GEGLU at K = 512, BERT's K = 768 / N = 3072; reference
    attention.py:38-51, BertIntermediate): every epilogue form against f32 torch math on the same bf16 operands, several
    tiles per workgroup, ragged edges."""
    A, B = bf(rnd(M, K, seed=21) * 0.5), bf(rnd(N, K, seed=22) * 0.5)
    ref = A.float() @ B.float().t()
    bias, res = rnd(N, seed=5), rnd(M, N, seed=6)
    C = torch.empty(M, N, device=DEV)
    hip.gemm5_bf16(A, B, C, bias, res, M, N, K, K, K, N, N, 1, 0.5, 0, None, 0)
    check(f"gemm5 direct f32 + bias + resid {M}x{N}x{K}", C, 0.5 * ref + bias + res, 2e-3)
    del C, res
    C16 = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    hip.gemm5_bf16(A, B, C16, bias, None, M, N, K, K, K, N, 0, 0, 1.0, 1, None, 0)
    check("gemm5 direct bf16 + bias + gelu", C16, torch.nn.functional.gelu(ref + bias), 1e-2)
    if N % 128 == 0:
        inner = N // 2
        Hh = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        G = torch.empty(M, inner, device=DEV, dtype=torch.bfloat16)
        hip.gemm5_bf16(A, B, Hh, None, None, M, N, K, K, K, N, 0, 0, 1.0, 2, G, inner)
        hb = ref.view(M, inner // 32, 2, 32)
        check("gemm5 direct fused h", Hh, ref, 1e-2)
        check("gemm5 direct fused g", G, (torch.nn.functional.gelu(hb[:, :, 1]) * hb[:, :, 0]).reshape(M, inner), 1.5e-2)
        h0 = bf(rnd(M, 2 * inner, seed=13))
        h2 = h0.clone()
        hip.gemm5_bf16(A, B[:inner], None, None, None, M, inner, K, K, K, 0, 0, 0, 1.0, 3, h2, 2 * inner)
        dg = bf(A.float() @ B[:inner].float().t()).float()
        hv = h0.float().view(M, inner // 32, 2, 32)
        val = hv[:, :, 0].reshape(M, inner).clone().requires_grad_(True)
        gate = hv[:, :, 1].reshape(M, inner).clone().requires_grad_(True)
        (torch.nn.functional.gelu(gate) * val).backward(dg)
        got = h2.float().view(M, inner // 32, 2, 32)
        check("gemm5 direct dgrad + geglu dval", got[:, :, 0].reshape(M, inner), val.grad, 2.5e-2)
        check("gemm5 direct dgrad + geglu dgate", got[:, :, 1].reshape(M, inner), gate.grad, 2.5e-2)
    # not eligible: K not a whole number of k64 pairs / shorter than the ring fill -> refused before anything is launched
    with pytest.raises(RuntimeError):
        hip.gemm5_bf16(A, B, C16, None, None, M, N, 160, K, K, N, 0, 0, 1.0, 0, None, 0)


def test_gemm_mfma_orientation_asymmetric(hip):
    """A = I with an asymmetric B catches a transposed C write (guide: 'always A=I-check')."""
    n = 128
    A = bf(torch.eye(n, device=DEV))
    Bw = bf(torch.arange(n * n, device=DEV, dtype=torch.float32).reshape(n, n) % 251)   # exact in bf16
    C = torch.empty(n, n, device=DEV, dtype=torch.float32)
    hip.gemm_bf16(A, Bw, C, None, None, n, n, n, n, n, n, 0, 1, 1, 1, 1, 0, 1.0, 0)    # C = I @ Bw^T
    assert torch.equal(C, Bw.float().t())
    hip.gemm_bf16(A, Bw, C, None, None, n, n, n, n, n, n, 0, 1, 0, 1, 1, 0, 1.0, 0)    # C = I @ Bw
    assert torch.equal(C, Bw.float())
    hip.gemm_bf16(Bw, A, C, None, None, n, n, n, n, n, n, 0, 0, 1, 1, 1, 0, 1.0, 0)    # C = Bw^T @ I
    assert torch.equal(C, Bw.float().t())


def test_gemm_epilogues_and_splitk(hip):
    M, N, K = 300, 264, 1024
    A, B = bf(rnd(M, K, seed=3)), bf(rnd(N, K, seed=4))
    bias, res = rnd(N, seed=5), rnd(M, N, seed=6)
    ref = A.float() @ B.float().t()
    C = torch.empty(M, N, device=DEV)
    hip.gemm_bf16(A, B, C, bias, res, M, N, K, K, K, N, N, 1, 1, 1, 1, 0, 0.5, 0)
    check("bias+resid+alpha", C, 0.5 * ref + bias + res, 2e-3)
    hip.gemm_bf16(A, B, C, bias, None, M, N, K, K, K, N, 0, 1, 1, 1, 1, 0, 1.0, 1)
    check("gelu", C, torch.nn.functional.gelu(ref + bias), 2e-3)
    C.copy_(res)
    hip.gemm_bf16(A, B, C, bias, None, M, N, K, K, K, N, 0, 1, 1, 1, 5, 1, 1.0, 0)
    check("split-k atomic accumulate", C, ref + bias + res, 2e-3)
    # strided output / operand views (ld > extent)
    big = torch.zeros(M, N + 24, device=DEV)
    hip.gemm_bf16(A, B, big, None, None, M, N, K, K, K, N + 24, 0, 1, 1, 1, 1, 0, 1.0, 0)
    check("ldc > N", big[:, :N], ref, 2e-3)
    assert float(big[:, N:].abs().max()) == 0.0


def test_gemm_argmax_partial_and_exact_select(hip):
    C_, T, K = 320, 200, 64
    Ef = torch.nn.functional.normalize(rnd(C_, K, seed=7), dim=-1).contiguous()
    Xraw = (rnd(T, K, seed=8) * 3).contiguous()
    # plant exact near-ties that bf16 scoring cannot resolve: token t is 1e-4 closer to code c2 than to c1
    for t, (c1, c2) in enumerate([(5, 70), (200, 201), (319, 3)]):
        Xraw[t] = (Ef[c1] + Ef[c2]) * 2 + 2e-4 * (Ef[c2] - Ef[c1])
    inv = 1.0 / Xraw.norm(dim=-1)
    E, X = bf(Ef), bf(Xraw * inv[:, None])
    ncand = 4 * ((C_ + 127) // 128)
    pv = torch.full((T, ncand), float("nan"), device=DEV)
    pi = torch.full((T, ncand), -1, device=DEV, dtype=torch.int32)
    hip.gemm_argmax_partial(E, X, pv, pi, C_, T, K, K, K)
    s16 = X.float() @ E.float().t()                          # what the MFMA pass scores
    valid = pi != 0x7fffffff
    best = torch.where(valid, pv, torch.full_like(pv, -1e30)).max(dim=1)
    got16 = s16.gather(1, pi.gather(1, best.indices[:, None]).long())[:, 0]
    assert float((s16.max(dim=1).values - got16).abs().max()) <= 1e-5, "slab top-2 must contain the bf16 arg-max"
    idx = torch.empty(T, dtype=torch.long, device=DEV)
    quant = torch.empty(T, K, device=DEV)
    hip.vq_select(pv, pi, ncand, Xraw, inv, Ef, idx, quant, T, K, 2.0 ** -7)
    exact = (Xraw * inv[:, None]) @ Ef.t()
    ref = exact.argmax(dim=1)
    gap = exact.max(dim=1).values - exact.gather(1, idx[:, None])[:, 0]
    assert float(gap.max()) <= 2e-7, "selected code must attain the exact f32 maximum"
    assert float((idx == ref).float().mean()) >= 0.995
    assert torch.equal(idx[:3].cpu(), torch.tensor([70, 201, 3]))
    assert torch.equal(quant, Ef[idx])
    # the sweep variant used by the VQ module: 16 candidates per token
    pv2 = torch.full((T, 16), float("nan"), device=DEV)
    pi2 = torch.full((T, 16), -1, device=DEV, dtype=torch.int32)
    hip.vq_topk(E, X, pv2, pi2, C_, T, K, K, K)
    idx2 = torch.empty(T, dtype=torch.long, device=DEV)
    hip.vq_select(pv2, pi2, 16, Xraw, inv, Ef, idx2, quant, T, K, 2.0 ** -7)
    gap2 = exact.max(dim=1).values - exact.gather(1, idx2[:, None])[:, 0]
    assert float(gap2.max()) <= 2e-7 and torch.equal(idx2[:3].cpu(), torch.tensor([70, 201, 3]))
    assert float((idx2 == ref).float().mean()) >= 0.995


@pytest.mark.parametrize("C_,T,K", [(512, 600, 64), (1024, 256, 512), (256, 40, 32)])
def test_vq_topk_whole_code_tiles(hip, C_, T, K):
    """ctclip_vq_topk when the codebook is whole 256-code tiles (the CT-ViT's 8192 x 512): the 256 x 256 LDS-DMA sweep
    in csrc/gemm3.hip.  Ragged token tiles, more and fewer K-steps than ring stages; the 16 candidates must contain the
    bf16 arg-max and the exact f32 re-ranking must return the true nearest code, planted near-ties included."""
    Ef = torch.nn.functional.normalize(rnd(C_, K, seed=7), dim=-1).contiguous()
    Xraw = (rnd(T, K, seed=8) * 3).contiguous()
    for t, (c1, c2) in enumerate([(5, 70), (200, 201), (C_ - 1, 3)]):
        Xraw[t] = (Ef[c1] + Ef[c2]) * 2 + 2e-4 * (Ef[c2] - Ef[c1])
    inv = 1.0 / Xraw.norm(dim=-1)
    E, X = bf(Ef), bf(Xraw * inv[:, None])
    pv = torch.full((T, 16), float("nan"), device=DEV)
    pi = torch.full((T, 16), -1, device=DEV, dtype=torch.int32)
    hip.vq_topk(E, X, pv, pi, C_, T, K, K, K)
    assert bool(((pi >= 0) & (pi < C_)).all()), "every candidate slot holds a code (256-code tiles fill all 16)"
    assert bool((pi.sort(dim=1).values.diff(dim=1) != 0).all()), "candidates of a token are distinct codes"
    s16 = X.float() @ E.float().t()
    check("candidate scores", pv, s16.gather(1, pi.long()), 1e-5)
    got16 = pv.max(dim=1).values
    assert float((s16.max(dim=1).values - got16).abs().max()) <= 1e-5, "candidates must contain the bf16 arg-max"
    idx = torch.empty(T, dtype=torch.long, device=DEV)
    quant = torch.empty(T, K, device=DEV)
    hip.vq_select(pv, pi, 16, Xraw, inv, Ef, idx, quant, T, K, 2.0 ** -7)
    exact = (Xraw * inv[:, None]) @ Ef.t()
    gap = exact.max(dim=1).values - exact.gather(1, idx[:, None])[:, 0]
    assert float(gap.max()) <= 2e-7 and torch.equal(idx[:3].cpu(), torch.tensor([70, 201, 3]))
    assert float((idx == exact.argmax(dim=1)).float().mean()) >= 0.995
    # a NaN token (a broken upstream weight) has no comparable score: it must come out as code 0, not as the empty-slot
    # marker used as a row number
    Xn = X.clone()
    Xn[T - 1] = float("nan")
    hip.vq_topk(E, Xn, pv, pi, C_, T, K, K, K)
    hip.vq_select(pv, pi, 16, Xraw, inv, Ef, idx, quant, T, K, 2.0 ** -7)
    assert int(idx[T - 1]) == 0 and bool(((idx >= 0) & (idx < C_)).all())


@pytest.mark.parametrize("C_,T,K,groups", [(1024, 600, 512, 2), (1024, 600, 512, 4), (2048, 2100, 64, 4), (2048, 2100, 64, 8),
                                           (8192, 300, 64, 32)])
def test_vq_topk_code_groups(hip, C_, T, K, groups):
    """ctclip_vq_topk_grouped: the codebook split into code groups, the workgroups an XCD holds at a time being (token tiles x
    code groups).  Token-tile counts that leave most of the last round of eight XCDs empty, ragged last tiles.  The candidate
    list [T][16 * groups] must contain everything the one-group list holds, every candidate must carry its own bf16 score, and
    the exact re-rank must give the same codes as with one group (the true f32 arg-max)."""
    Ef = torch.nn.functional.normalize(rnd(C_, K, seed=17), dim=-1).contiguous()
    Xraw = (rnd(T, K, seed=18) * 3).contiguous()
    for t, (c1, c2) in enumerate([(5, C_ // 2 + 70), (200, 201), (C_ - 1, 3)]):
        Xraw[t] = (Ef[c1] + Ef[c2]) * 2 + 2e-4 * (Ef[c2] - Ef[c1])
    inv = 1.0 / Xraw.norm(dim=-1)
    E, X = bf(Ef), bf(Xraw * inv[:, None])
    pv1 = torch.full((T, 16), float("nan"), device=DEV)
    pi1 = torch.full((T, 16), -1, device=DEV, dtype=torch.int32)
    hip.vq_topk_grouped(E, X, pv1, pi1, C_, T, K, K, K, 1)
    pv = torch.full((T, 16 * groups), float("nan"), device=DEV)
    pi = torch.full((T, 16 * groups), -1, device=DEV, dtype=torch.int32)
    hip.vq_topk_grouped(E, X, pv, pi, C_, T, K, K, K, groups)
    per = C_ // groups
    assert bool(((pi >= 0) & (pi < C_)).all()), "every slot holds a code"
    lo = (torch.arange(groups, device=DEV) * per).repeat_interleave(16)[None]
    assert bool(((pi >= lo) & (pi < lo + per)).all()), "a group's candidates are codes of that group"
    assert bool((pi.sort(dim=1).values.diff(dim=1) != 0).all()), "candidates of a token are distinct codes"
    s16 = X.float() @ E.float().t()
    check("candidate scores", pv, s16.gather(1, pi.long()), 1e-5)
    member = (pi1[:, :, None] == pi[:, None, :]).any(dim=2)
    assert bool(member.all()), "the grouped list contains the one-group list"
    idx1 = torch.empty(T, dtype=torch.long, device=DEV)
    idx = torch.empty(T, dtype=torch.long, device=DEV)
    quant = torch.empty(T, K, device=DEV)
    hip.vq_select(pv1, pi1, 16, Xraw, inv, Ef, idx1, quant, T, K, 2.0 ** -7)
    hip.vq_select(pv, pi, 16 * groups, Xraw, inv, Ef, idx, quant, T, K, 2.0 ** -7)
    exact = (Xraw * inv[:, None]) @ Ef.t()
    gap = exact.max(dim=1).values - exact.gather(1, idx[:, None])[:, 0]
    assert float(gap.max()) <= 2e-7 and torch.equal(idx[:3].cpu(), torch.tensor([C_ // 2 + 70, 201, 3]))
    gap1 = exact.max(dim=1).values - exact.gather(1, idx1[:, None])[:, 0]
    assert float((gap - gap1).max()) <= 0.0, "more candidates can only move a token to a code at least as near"
    assert torch.equal(quant, Ef[idx])
    # 3 / 5 / 33 groups do not divide the 32 workgroups of an XCD or the code tiles: refused, nothing launched
    for bad in (3, 5, 64):
        with pytest.raises(RuntimeError):
            hip.vq_topk_grouped(E, X, pv, pi, C_, T, K, K, K, bad)


@pytest.mark.parametrize("nseq,n,H,D,use_bias,use_mask", [
    (3, 128, 4, 64, False, True),      # BERT shape with a padding mask
    (2, 40, 2, 32, True, False),       # ragged rows, dense bias and its gradient
    (2, 6, 3, 64, False, True),        # n % 4 != 0: byte-wise flag reads
    (2, 64, 2, 32, False, False),      # would take the wave-per-sequence kernels (attention_ws.hip) without dropout
    (4, 24, 2, 32, False, False),      # would take the one-wave backward without dropout
])
def test_attention_probability_dropout(hip, nseq, n, H, D, use_bias, use_mask):
    """ctclip_attn_fwd_dropout / _bwd_dropout (transformers BertSelfAttention): softmax over all keys, then the kept
    probabilities scaled by 1/(1-p) go into P.V; the backward uses the same flags.  p = 0.25, flags from the caller."""
    pdrop = 0.25
    scale = 1.0 / math.sqrt(D)
    ld = H * D
    q, k, v, do = (bf(rnd(nseq * n, ld, seed=s)) for s in (40, 41, 42, 43))
    bias = rnd(H, n, n, seed=44) if use_bias else None
    mask = None
    if use_mask:
        lens = torch.randint(max(1, n // 2), n + 1, (nseq,), generator=torch.Generator().manual_seed(6))
        lens[0] = n
        mask = ((torch.arange(n)[None] >= lens[:, None]).float() * torch.finfo(torch.float32).min).to(DEV)
    keep = (torch.rand(nseq, H, n, n, generator=torch.Generator().manual_seed(9)) >= pdrop).to(torch.uint8).to(DEV)
    sp = lambda t: t.float().reshape(nseq, n, H, D).permute(0, 2, 1, 3).contiguous().requires_grad_(True)
    qr, kr, vr = sp(q), sp(k), sp(v)
    br = bias.clone().requires_grad_(True) if use_bias else None
    s = torch.einsum("shid,shjd->shij", qr, kr) * scale
    if br is not None:
        s = s + br[None]
    if mask is not None:
        s = s + mask[:, None, None, :]
    pref = s.softmax(-1) * keep.float() / (1.0 - pdrop)
    oref = torch.einsum("shij,shjd->shid", pref, vr)
    o = torch.empty(nseq * n, ld, device=DEV, dtype=torch.bfloat16)
    lse = torch.empty(nseq, H, n, device=DEV)
    hip.attn_fwd_dropout(q, k, v, o, lse, bias, mask, keep, 1.0 / (1.0 - pdrop), nseq, n, H, D, ld, ld, ld, ld, scale)
    un = lambda t: t.float().reshape(nseq, n, H, D).permute(0, 2, 1, 3)
    check("dropout attn out", un(o), oref, 2e-2)
    check("lse is the undropped row's", lse, torch.logsumexp(s.detach(), -1), 1e-3)
    oref.backward(un(do))
    dq, dk, dv = (torch.empty(nseq * n, ld, device=DEV, dtype=torch.bfloat16) for _ in range(3))
    delta = torch.empty(nseq, H, n, device=DEV)
    dbias = torch.zeros(H, n, n, device=DEV) if use_bias else None
    hip.attn_bwd_dropout(q, k, v, o, do, lse, delta, dq, dk, dv, bias, mask, keep, 1.0 / (1.0 - pdrop), dbias, None, None, 0,
                         0, 0, nseq, n, H, D, ld, ld, ld, ld, ld, ld, ld, ld, scale)
    check("dropout attn dq", un(dq), qr.grad, 3e-2)
    check("dropout attn dk", un(dk), kr.grad, 3e-2)
    check("dropout attn dv", un(dv), vr.grad, 3e-2)
    if use_bias:
        check("dropout attn dbias", dbias, br.grad, 3e-2)
    # all flags set and scale 1 is the plain kernel pair
    ones = torch.ones_like(keep)
    o2 = torch.empty_like(o)
    hip.attn_fwd_dropout(q, k, v, o2, lse, bias, mask, ones, 1.0, nseq, n, H, D, ld, ld, ld, ld, scale)
    o3 = torch.empty_like(o)
    hip.attn_fwd(q, k, v, o3, lse, bias, mask, nseq, n, H, D, ld, ld, ld, ld, scale)
    check("keep-all == no dropout", o2, o3.float(), 1e-2)


# ---------------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("B,A,C,dim", [(2, 3, 5, 512), (1, 24, 7, 64), (3, 4, 4, 56)])
def test_layernorm_with_token_reordering(hip, B, A, C, dim):
    """ctclip_layernorm_swap_fwd / _bwd: Transformer.norm_out (attention.py:311,336) writing its rows [B][A][C] straight
    into the [B][C][A] order of the rearrange that follows it in CTViT.encode (ctvit.py:96,101), and the backward reading
    dy in that order."""
    rows = B * A * C
    x = (rnd(rows, dim, seed=31) * 2 + 0.5).requires_grad_(True)
    gm = (1 + 0.3 * rnd(dim, seed=32)).requires_grad_(True)
    ref = torch.nn.functional.layer_norm(x, (dim,), gm, None, 1e-5).reshape(B, A, C, dim).permute(0, 2, 1, 3).contiguous()
    y = torch.empty(B, C, A, dim, device=DEV)
    mean, rstd = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    hip.layernorm_swap_fwd(x.detach(), gm.detach(), None, y, mean, rstd, rows, dim, 1e-5, A, C)
    check("swapped ln y", y, ref, 1e-5)
    dy = rnd(B, C, A, dim, seed=33)
    ref.backward(dy)
    dx = torch.empty(rows, dim, device=DEV)
    dx16 = torch.empty(rows, dim, device=DEV, dtype=torch.bfloat16)
    dg = torch.zeros(dim, device=DEV)
    hip.layernorm_swap_bwd(dy, x.detach(), gm.detach(), mean, rstd, dx, dx16, dg, None, rows, dim, A, C)
    check("swapped ln dx", dx, x.grad, 1e-4)
    check("swapped ln dx16", dx16, x.grad, 1e-2)
    check("swapped ln dgamma", dg, gm.grad, 1e-4)


@pytest.mark.parametrize("rows,dim,with_beta", [(37, 512, False), (130, 768, True), (9, 56, True), (5, 4000, True)])
def test_layernorm_fwd_bwd(hip, rows, dim, with_beta):
    x = (rnd(rows, dim, seed=9) * 2 + 0.5).requires_grad_(True)
    gm = (1 + 0.3 * rnd(dim, seed=10)).requires_grad_(True)
    bt = (0.2 * rnd(dim, seed=11)).requires_grad_(True) if with_beta else None
    ref = torch.nn.functional.layer_norm(x, (dim,), gm, bt, 1e-5)
    y16 = torch.empty(rows, dim, device=DEV, dtype=torch.bfloat16)
    y32 = torch.empty(rows, dim, device=DEV)
    mean, rstd = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    hip.layernorm_fwd(x.detach(), gm.detach(), None if bt is None else bt.detach(), y16, y32, mean, rstd, rows, dim, 1e-5)
    check("ln y32", y32, ref, 1e-5)
    check("ln y16", y16, ref, 1e-2)
    dy, dres = rnd(rows, dim, seed=12), rnd(rows, dim, seed=13)
    ref.backward(dy)
    dx = torch.empty(rows, dim, device=DEV)
    dx16 = torch.empty(rows, dim, device=DEV, dtype=torch.bfloat16)
    dg, db = torch.zeros(dim, device=DEV), torch.zeros(dim, device=DEV)
    hip.layernorm_bwd(dy, x.detach(), gm.detach(), mean, rstd, dres, dx, dx16, dg, db if with_beta else None, rows, dim)
    check("ln dx", dx, x.grad + dres, 2e-5)
    check("ln dx16", dx16, x.grad + dres, 1e-2)
    check("ln dgamma", dg, gm.grad, 2e-5)
    if with_beta:
        check("ln dbeta", db, bt.grad, 2e-5)
    # bf16 LN-path gradient + f32 residual term + bf16 second residual term: exact in the bf16-rounded inputs
    dy16, dres2 = bf(dy), bf(rnd(rows, dim, seed=14))
    x.grad = None
    gm.grad = None
    torch.nn.functional.layer_norm(x, (dim,), gm, bt, 1e-5).backward(dy16.float())
    dg2, db2 = torch.zeros(dim, device=DEV), torch.zeros(dim, device=DEV)
    hip.layernorm_bwd_bf16(dy16, x.detach(), gm.detach(), mean, rstd, dres, dres2, dx, dx16, dg2, db2 if with_beta else None,
                           rows, dim)
    check("ln16 dx", dx, x.grad + dres + dres2.float(), 2e-5)
    check("ln16 dx16", dx16, x.grad + dres + dres2.float(), 1e-2)
    check("ln16 dgamma", dg2, gm.grad, 2e-5)
    hip.layernorm_bwd_bf16(dy16, x.detach(), gm.detach(), mean, rstd, None, None, dx, None, dg2, None, rows, dim)
    check("ln16 dx (no residual terms)", dx, x.grad, 2e-5)


@pytest.mark.parametrize("D", [32, 64])
def test_headnorm_fwd_bwd(hip, D):
    rows, H, mult = 77, 3, 8.0
    x = bf(rnd(rows, H * D + 16, seed=14))[:, : H * D]                   # strided view
    sc = 1 + 0.3 * rnd(D, seed=15)
    xr = x.float().reshape(rows, H, D).requires_grad_(True)
    scr = sc.clone().requires_grad_(True)
    ref = torch.nn.functional.normalize(xr, dim=-1) * scr * mult
    y = torch.empty(rows, H * D, device=DEV, dtype=torch.bfloat16)
    inv = torch.empty(rows, H, device=DEV)
    hip.headnorm_fwd(x, sc, y, inv, rows, H, D, x.stride(0), H * D, mult, 0, 0)
    check("headnorm y", y.float().reshape(rows, H, D), ref, 1e-2)
    dy = bf(rnd(rows, H * D, seed=16))
    ref.backward(dy.float().reshape(rows, H, D))
    dx = torch.empty(rows, H * D, device=DEV, dtype=torch.bfloat16)
    ds = torch.zeros(D, device=DEV)
    hip.headnorm_bwd(dy, x, inv, sc, dx, ds, rows, H, D, H * D, x.stride(0), H * D, mult, 0, 0)
    check("headnorm dx", dx.float().reshape(rows, H, D), xr.grad, 1e-2)
    check("headnorm dscale", ds, scr.grad, 1e-3)


# ---------------------------------------------------------------------------------------------- attention
def attn_ref(q, k, v, bias, mask, scale):
    s = torch.einsum("shid,shjd->shij", q, k) * scale
    if bias is not None:
        s = s + bias[None]
    if mask is not None:
        s = s + mask[:, None, None, :]
    p = s.softmax(-1)
    return torch.einsum("shij,shjd->shid", p, v), p


@pytest.mark.parametrize("nseq,n,H,D,use_bias,use_mask", [
    (3, 576, 8, 32, True, False),      # CT-ViT spatial
    (40, 24, 8, 32, False, False),     # CT-ViT temporal
    (4, 128, 12, 64, False, True),     # BERT L=128
    (2, 128, 2, 32, True, False),      # 8 x 16 position grid: image rows wrap inside a 32-query tile
    (2, 40, 2, 32, True, False),       # ragged: n not a multiple of 32
    (2, 6, 4, 32, True, True),         # n % 4 != 0
    (1, 512, 2, 64, False, True),      # BERT L=512 (128 KiB of LDS)
    (7, 320, 3, 32, True, False),      # wave-per-sequence kernels (attention_ws.hip), 10 key tiles, odd head count
    (5, 256, 2, 32, False, False),     # wave-per-sequence kernels without a bias
    (4, 160, 2, 32, True, False),      # 5 tiles: the last query-block / key-block pair of a workgroup is half empty
    (2, 640, 1, 32, True, False),      # 20 tiles: the fused dQ + d(bias) pass only has room for 8 waves here
    (2, 704, 1, 32, True, False),      # 22 tiles: forward by the wave-per-sequence kernel, backward by the per-sequence kernels (LDS)
    (5, 24, 3, 32, False, True),       # one-wave fused backward: masked keys, 15 (sequence, head) items over 8-wave groups
    (3, 32, 2, 32, True, False),       # one-wave fused backward at its largest row count, bias as an input only
])
def test_attention_fwd_bwd(hip, nseq, n, H, D, use_bias, use_mask, monkeypatch):
    scale = 1.0 if D == 32 else 1.0 / math.sqrt(D)
    if n == 320:
        scale = 0.75                   # exercises the bias / scale folding of the persistent kernels
    if D == 32 and n % 32 == 0 and nseq > 2:
        monkeypatch.setenv("CTCLIP_ATTN_SP_CHUNK", "3")   # several sequences per workgroup + a ragged last chunk
    ld = H * D
    q, k, v = (rnd(nseq * n, ld, seed=s) for s in (20, 21, 22))
    if D == 32:   # what ctclip_headnorm_fwd feeds the kernel: unit rows per head, q carries the fixed scale 8
        unit = lambda t: torch.nn.functional.normalize(t.reshape(nseq * n, H, D), dim=-1).reshape(nseq * n, ld)
        q, k = unit(q) * 8.0, unit(k)
    q, k, v = bf(q), bf(k), bf(v)
    do = bf(rnd(nseq * n, ld, seed=23))
    bias = rnd(H, n, n, seed=24) if use_bias else None
    mask = None
    if use_mask:
        lens = torch.randint(max(1, n // 3), n + 1, (nseq,), generator=torch.Generator().manual_seed(5))
        lens[0] = n
        mask = ((torch.arange(n)[None] >= lens[:, None]).float() * torch.finfo(torch.float32).min).to(DEV)
    sp = lambda t: t.float().reshape(nseq, n, H, D).permute(0, 2, 1, 3).contiguous().requires_grad_(True)
    qr, kr, vr = sp(q), sp(k), sp(v)
    br = bias.clone().requires_grad_(True) if use_bias else None
    oref, pref = attn_ref(qr, kr, vr, br, mask, scale)
    o = torch.empty(nseq * n, ld, device=DEV, dtype=torch.bfloat16)
    lse = torch.empty(nseq, H, n, device=DEV)
    hip.attn_fwd(q, k, v, o, lse, bias, mask, nseq, n, H, D, ld, ld, ld, ld, scale)
    og = o.float().reshape(nseq, n, H, D).permute(0, 2, 1, 3)
    check("attn out", og, oref, 2e-2)
    probs = torch.empty(nseq, H, n, n, device=DEV)
    hip.attn_probs(q, k, lse, bias, mask, probs, nseq, n, H, D, ld, ld, scale)
    check("attn probs", probs, pref, 2e-2)

    oref.backward(do.float().reshape(nseq, n, H, D).permute(0, 2, 1, 3))
    dq, dk, dv = (torch.empty(nseq * n, ld, device=DEV, dtype=torch.bfloat16) for _ in range(3))
    delta = torch.empty(nseq, H, n, device=DEV)
    dbias = torch.zeros(H, n, n, device=DEV) if use_bias else None
    hip.attn_bwd(q, k, v, o, do, lse, delta, dq, dk, dv, bias, mask, dbias, None, None, 0, 0, 0, nseq, n, H, D,
                 ld, ld, ld, ld, ld, ld, ld, ld, scale)
    un = lambda t: t.float().reshape(nseq, n, H, D).permute(0, 2, 1, 3)
    check("attn dq", un(dq), qr.grad, 3e-2)
    check("attn dk", un(dk), kr.grad, 3e-2)
    check("attn dv", un(dv), vr.grad, 3e-2)
    if use_bias:
        check("attn dbias dense", dbias, br.grad, 3e-2)
        # bias as an input only (no bias gradient asked for): n <= 32 takes the one-wave fused backward
        for t in (dq, dk, dv):
            t.zero_()
        hip.attn_bwd(q, k, v, o, do, lse, delta, dq, dk, dv, bias, mask, None, None, None, 0, 0, 0, nseq, n, H, D,
                     ld, ld, ld, ld, ld, ld, ld, ld, scale)
        check("attn dq (no dbias)", un(dq), qr.grad, 3e-2)
        check("attn dk (no dbias)", un(dk), kr.grad, 3e-2)
        check("attn dv (no dbias)", un(dv), vr.grad, 3e-2)
        # table mode: bias[h,i,j] = table[h, relidx[i,j]]; d(table) = scatter-add of d(bias)
        R = 37
        relidx = torch.randint(0, R, (n, n), generator=torch.Generator().manual_seed(3)).to(DEV)
        dt = torch.zeros(H, R, device=DEV)
        hip.attn_bwd(q, k, v, o, do, lse, delta, dq, dk, dv, bias, mask, None, relidx.to(torch.uint16), dt, R, 0, 0,
                     nseq, n, H, D, ld, ld, ld, ld, ld, ld, ld, ld, scale)
        ref_t = torch.zeros(H, R, device=DEV).index_add_(1, relidx.reshape(-1), br.grad.reshape(H, -1))
        check("attn dbias table", dt, ref_t, 3e-2)
        # 2-D relative-position mode (index computed on chip): n = gh*gw
        gh = next(d for d in (24, 8, 5, 4, 3, 2, 1) if n % d == 0)
        gw = n // gh
        ii = torch.arange(n, device=DEV)
        rel2 = ((ii[:, None] // gw - ii[None] // gw + gh - 1) * (2 * gw - 1) + (ii[:, None] % gw - ii[None] % gw + gw - 1))
        R2 = (2 * gh - 1) * (2 * gw - 1)
        dt2 = torch.zeros(H, R2, device=DEV)
        hip.attn_bwd(q, k, v, o, do, lse, delta, dq, dk, dv, bias, mask, None, None, dt2, R2, gh, gw,
                     nseq, n, H, D, ld, ld, ld, ld, ld, ld, ld, ld, scale)
        ref2 = torch.zeros(H, R2, device=DEV).index_add_(1, rel2.reshape(-1), br.grad.reshape(H, -1))
        check("attn dbias 2-D grid table", dt2, ref2, 3e-2)


def test_spatial_attention_bias_gradient_under_lds_locks_is_reproducible(hip):
    """The CT-ViT spatial shape with enough sequences that every wave of the fused dQ + d(bias) pass competes for the
    per-tile LDS locks many times: repeated runs agree to f32 summation order, the relative-position table equals the
    dense gradient scattered by index, dQ is bit-identical run to run."""
    nseq, n, H, D, gh, gw = 192, 576, 8, 32, 24, 24
    ld = H * D
    unit = lambda t: torch.nn.functional.normalize(t.reshape(nseq * n, H, D), dim=-1).reshape(nseq * n, ld)
    q, k = bf(unit(rnd(nseq * n, ld, seed=80)) * 8.0), bf(unit(rnd(nseq * n, ld, seed=81)))
    v, do = bf(rnd(nseq * n, ld, seed=82)), bf(rnd(nseq * n, ld, seed=83))
    bias = rnd(H, n, n, seed=84)
    o = torch.empty_like(q)
    lse = torch.empty(nseq, H, n, device=DEV)
    hip.attn_fwd(q, k, v, o, lse, bias, None, nseq, n, H, D, ld, ld, ld, ld, 1.0)
    dq, dk, dv = (torch.empty_like(q) for _ in range(3))
    delta = torch.empty_like(lse)
    R = (2 * gh - 1) * (2 * gw - 1)
    runs = []
    for _ in range(4):
        dt = torch.zeros(H, R, device=DEV)
        hip.attn_bwd(q, k, v, o, do, lse, delta, dq, dk, dv, bias, None, None, None, dt, R, gh, gw, nseq, n, H, D,
                     ld, ld, ld, ld, ld, ld, ld, ld, 1.0)
        torch.cuda.synchronize()
        runs.append((dt, dq.clone()))
    for dt, dq_i in runs[1:]:
        assert float((dt - runs[0][0]).abs().max() / runs[0][0].abs().max()) < 1e-5
        assert torch.equal(dq_i, runs[0][1])
    dense = torch.zeros(H, n, n, device=DEV)
    hip.attn_bwd(q, k, v, o, do, lse, delta, dq, dk, dv, bias, None, dense, None, None, 0, 0, 0, nseq, n, H, D,
                 ld, ld, ld, ld, ld, ld, ld, ld, 1.0)
    ii = torch.arange(n, device=DEV)
    rel = (ii[:, None] // gw - ii[None] // gw + gh - 1) * (2 * gw - 1) + (ii[:, None] % gw - ii[None] % gw + gw - 1)
    ref = torch.zeros(H, R, device=DEV).index_add_(1, rel.reshape(-1), dense.reshape(H, -1))
    check("table vs scattered dense d(bias)", runs[0][0], ref, 1e-5)


# ---------------------------------------------------------------------------------------------- head-major spatial attention
def to_hm(t, nseq, n, H, D):
    """row-major [nseq * n, H * D] -> head-major [nseq, H, n, D]"""
    return t.reshape(nseq, n, H, D).permute(0, 2, 1, 3).contiguous()


def test_headnorm_and_gemm_write_head_major(hip):
    """The producers of the head-major operands: ctclip_headnorm_fwd / _bwd with x_hm_n / y_hm_n, and
    ctclip_gemm_bf16_headmajor (the kv projection's two parts; a ragged last row tile), against the row-major forms."""
    nseq, n, H, D, K = 3, 96, 4, 32, 64
    rows = nseq * n
    x = bf(rnd(rows, H * D, seed=90))
    sc = 1 + 0.3 * rnd(D, seed=91)
    y_rm, inv_rm = torch.empty_like(x), torch.empty(rows, H, device=DEV)
    hip.headnorm_fwd(x, sc, y_rm, inv_rm, rows, H, D, H * D, H * D, 8.0, 0, 0)
    y_hm, inv = torch.empty(nseq, H, n, D, device=DEV, dtype=torch.bfloat16), torch.empty(rows, H, device=DEV)
    hip.headnorm_fwd(x, sc, y_hm, inv, rows, H, D, H * D, 0, 8.0, 0, n)             # row-major in, head-major out
    assert torch.equal(y_hm, to_hm(y_rm, nseq, n, H, D)) and torch.equal(inv, inv_rm)
    x_hm = to_hm(x, nseq, n, H, D)
    y2 = torch.empty_like(y_hm)
    hip.headnorm_fwd(x_hm, sc, y2, inv, rows, H, D, 0, 0, 8.0, n, n)                 # head-major in and out
    assert torch.equal(y2, y_hm)
    dy = bf(rnd(rows, H * D, seed=92))
    dx_rm, ds_rm = torch.empty_like(x), torch.zeros(D, device=DEV)
    hip.headnorm_bwd(dy, x, inv_rm, sc, dx_rm, ds_rm, rows, H, D, H * D, H * D, H * D, 8.0, 0, 0)
    dx2, ds2 = torch.empty_like(x), torch.zeros(D, device=DEV)
    hip.headnorm_bwd(dy, x_hm, inv_rm, sc, dx2, ds2, rows, H, D, H * D, 0, H * D, 8.0, n, 0)   # x head-major
    assert torch.equal(dx2, dx_rm) and torch.equal(ds2, ds_rm)
    # GEMM: M = 288 rows (one full 256-row tile + a ragged one), N = 2 parts x 4 heads x 32
    A, B = bf(rnd(rows, K, seed=93)), bf(rnd(2 * H * D, K, seed=94))
    C_rm = torch.empty(rows, 2 * H * D, device=DEV, dtype=torch.bfloat16)
    hip.gemm_bf16(A, B, C_rm, None, None, rows, 2 * H * D, K, K, K, 2 * H * D, 0, 1, 1, 0, 1, 0, 1.0, 0)
    C_hm = torch.full((2, nseq, H, n, D), 7.0, device=DEV, dtype=torch.bfloat16)
    hip.gemm_bf16_headmajor(A, B, C_hm, rows, 2 * H * D, K, K, K, n, H)
    want = torch.stack((to_hm(C_rm[:, : H * D], nseq, n, H, D), to_hm(C_rm[:, H * D:], nseq, n, H, D)))
    check("head-major GEMM output", C_hm, want.float(), 1e-2)
    assert torch.equal(C_hm, want)                                            # same kernel, same arithmetic: only the addresses differ


@pytest.mark.parametrize("nseq,n,H,K", [(3, 96, 4, 64),       # 288 rows: one full 256-row tile + a ragged one; N = 256 (k | v: 2 x 4 heads)
                                        (28, 576, 8, 512),     # the CT-ViT spatial block: 16 128 rows, dim 512, 8 heads
                                        (700, 24, 8, 512),     # the temporal block's rows (row-major output is what it uses)
                                        (5, 40, 2, 96)])       # one slab per part, three K-steps
def test_projection_with_head_norm_in_the_gemm_epilogue(hip, nseq, n, H, K):
    """ctclip_gemm_bf16_headnorm: the q projection and the k | v projection with the per-head cosine normalisation of
    attention.py:146-153 in the GEMM's register epilogue (gemm_tile.h EPI 5) -- head-major and row-major outputs, 1 / norm, the v
    half untouched -- against f32 torch on the same bf16 operands; then ctclip_headnorm_bwd from the NORMALISED rows (x_normed = 1:
    the raw projection is never kept) against autograd through F.normalize * scale."""
    D = 32
    M, inner = nseq * n, H * D
    A = bf(rnd(M, K, seed=301))
    Wq, Wkv = bf(rnd(inner, K, seed=302) * 0.2), bf(rnd(2 * inner, K, seed=303) * 0.2)
    qs, ks = 1 + 0.2 * rnd(D, seed=304), 1 + 0.2 * rnd(D, seed=305)
    mult = 8.0 * 1.4426950408889634
    q_raw, kv_raw = A.float() @ Wq.float().t(), A.float() @ Wkv.float().t()
    unit = lambda t: torch.nn.functional.normalize(t.reshape(M, H, D), dim=-1)
    q_ref = (unit(q_raw) * qs * mult).reshape(M, inner)
    k_ref = (unit(kv_raw[:, :inner]) * ks).reshape(M, inner)
    qinv_ref = 1.0 / q_raw.reshape(M, H, D).norm(dim=-1).clamp_min(1e-12)
    kinv_ref = 1.0 / kv_raw[:, :inner].reshape(M, H, D).norm(dim=-1).clamp_min(1e-12)
    # head-major
    qh = torch.full((nseq, H, n, D), 7.0, device=DEV, dtype=torch.bfloat16)
    kvh = torch.full((2, nseq, H, n, D), 7.0, device=DEV, dtype=torch.bfloat16)
    qinv, kinv = torch.empty(M, H, device=DEV), torch.empty(M, H, device=DEV)
    hip.gemm_bf16_headnorm(A, Wq, qh, qinv, qs, M, inner, K, K, K, 0, n, H, inner, mult)
    hip.gemm_bf16_headnorm(A, Wkv, kvh, kinv, ks, M, 2 * inner, K, K, K, 0, n, H, inner, 1.0)
    check("q normalised, head-major", qh, to_hm(q_ref, nseq, n, H, D), 1e-2)
    check("k normalised, head-major", kvh[0], to_hm(k_ref, nseq, n, H, D), 1e-2)
    check("v untouched, head-major", kvh[1], to_hm(kv_raw[:, inner:], nseq, n, H, D), 1e-2)
    check("1 / |q|", qinv, qinv_ref, 2e-3)
    check("1 / |k|", kinv, kinv_ref, 2e-3)
    # row-major (ldc > N for q: the columns beyond stay untouched)
    q_rm = torch.zeros(M, inner + 64, device=DEV, dtype=torch.bfloat16)
    kv_rm = torch.empty(M, 2 * inner, device=DEV, dtype=torch.bfloat16)
    qinv2, kinv2 = torch.empty(M, H, device=DEV), torch.empty(M, H, device=DEV)
    hip.gemm_bf16_headnorm(A, Wq, q_rm, qinv2, qs, M, inner, K, K, K, inner + 64, 0, H, inner, mult)
    hip.gemm_bf16_headnorm(A, Wkv, kv_rm, kinv2, ks, M, 2 * inner, K, K, K, 2 * inner, 0, H, inner, 1.0)
    assert torch.equal(to_hm(q_rm[:, :inner], nseq, n, H, D), qh) and torch.equal(qinv2, qinv)     # same arithmetic, other addresses
    assert float(q_rm[:, inner:].float().abs().max()) == 0.0
    assert torch.equal(to_hm(kv_rm[:, :inner], nseq, n, H, D), kvh[0]) and torch.equal(to_hm(kv_rm[:, inner:], nseq, n, H, D), kvh[1])
    assert torch.equal(kinv2, kinv)
    # backward from the normalised rows: against autograd through normalize * scale * mult at the raw projection
    dy = bf(rnd(M, inner, seed=306))
    qr = q_raw.reshape(M, H, D).clone().requires_grad_(True)
    qsr = qs.clone().requires_grad_(True)
    (torch.nn.functional.normalize(qr, dim=-1) * qsr * mult).backward(dy.float().reshape(M, H, D))
    for x_in, ldx, hmn in ((q_rm, inner + 64, 0), (qh, 0, n)):
        dx, ds = torch.empty(M, inner, device=DEV, dtype=torch.bfloat16), torch.zeros(D, device=DEV)
        hip.headnorm_bwd(dy, x_in, qinv, qs, dx, ds, M, H, D, inner, ldx, inner, mult, hmn, 1)
        check(f"dq from the normalised rows (hm_n {hmn})", dx, qr.grad.reshape(M, inner), 1.2e-2)
        check("d(q_scale) from the normalised rows", ds, qsr.grad, 5e-3)
    # a zero row (|x| = 0): 1 / norm saturates at 1e12, the output row is 0, nothing is NaN
    A0 = A.clone()
    A0[1] = 0
    hip.gemm_bf16_headnorm(A0, Wq, qh, qinv, qs, M, inner, K, K, K, 0, n, H, inner, mult)
    assert float(qh.float().abs().reshape(nseq, H, n, D)[0, :, 1].max()) == 0.0 and bool(torch.isfinite(qinv).all())
    assert float(qinv[1].min()) > 9e11


@pytest.mark.parametrize("nseq,n,H,use_bias,chunk", [
    (5, 64, 2, True, 0),        # two tiles, one query-block group
    (12, 576, 8, True, 0),      # the CT-ViT spatial shape: 18 tiles, nine groups of two query blocks
    (7, 160, 3, True, 3),       # 5 tiles: a half-empty last group; three sequences per workgroup + a ragged last chunk
    (4, 96, 4, False, 0),       # no bias
    (3, 640, 1, True, 0),       # 20 tiles: the d(bias) pass has room for 8 waves only
    (37, 96, 2, True, 5),       # ODD tile count (3) with the operand stream running across sequences (the next sequence's first
                                # tiles land in the slots by parity), five sequences per workgroup handed out from the LDS counter
    (19, 32, 2, True, 0),       # ONE tile per sequence: every request of a sequence is already the next sequence's
    (41, 64, 1, True, 7),       # two tiles, chunks that are no multiple of the wave count, one head
])
def test_head_major_attention_fwd_bwd(hip, nseq, n, H, use_bias, chunk, monkeypatch):
    """ctclip_attn_hm_fwd / _bwd (csrc/attention_hm.hip) against f32 torch on the same bf16-rounded operands: log2-domain
    logits (q carries scale * log2 e, natural logit = ln2 * q.k + bias), no-maximum softmax under ctclip_attn_shift's bound, bias
    through the score MFMA's C operand (forward) / the f16 identity MFMA (backward), sequences handed out to the waves from an
    LDS counter with the K / V / q stream running across them, out / lse / dq / dk / dv and d(bias) in its dense, index-table
    and 2-D grid forms.
    Also: the online-softmax kernel of the launch pair (no shift, and a shift flagged unsafe) gives the same result."""
    D, LOG2E, LN2 = 32, 1.4426950408889634, 0.6931471805599453
    if chunk:
        monkeypatch.setenv("CTCLIP_ATTN_SP_CHUNK", str(chunk))
    ld = H * D
    qs, ks = 1 + 0.2 * rnd(D, seed=101), 1 + 0.2 * rnd(D, seed=102)
    unit = lambda t: torch.nn.functional.normalize(t.reshape(nseq * n, H, D), dim=-1)
    qmult = 8.0 * LOG2E
    q = bf((unit(rnd(nseq * n, ld, seed=103)) * qs * qmult).reshape(nseq * n, ld))
    k = bf((unit(rnd(nseq * n, ld, seed=104)) * ks).reshape(nseq * n, ld))
    v, do = bf(rnd(nseq * n, ld, seed=105)), bf(rnd(nseq * n, ld, seed=106))
    bias = 1.5 * rnd(H, n, n, seed=107) if use_bias else None
    sp = lambda t: t.float().reshape(nseq, n, H, D).permute(0, 2, 1, 3).contiguous().requires_grad_(True)
    qr, kr, vr = sp(q), sp(k), sp(v)
    br = bias.clone().requires_grad_(True) if use_bias else None
    oref, _ = attn_ref(qr, kr, vr, br, None, LN2)
    lse_ref = torch.logsumexp(torch.einsum("shid,shjd->shij", qr, kr).detach() * LN2 + (bias[None] if use_bias else 0.0), dim=-1)
    q_hm, k_hm, v_hm, do_hm = (to_hm(t, nseq, n, H, D) for t in (q, k, v, do))
    shift = torch.empty(H + 1, device=DEV)
    if use_bias:
        hip.attn_shift(qs, ks, D, qmult, bias, n * n, n * n, 1, H, shift)
    else:
        hip.attn_shift(qs, ks, D, qmult, None, 0, 0, 0, H, shift)
    assert float(shift[H]) == 0.0                                          # the bound fits f32 with room to spare
    smax = (torch.einsum("shid,shjd->shij", qr, kr).detach() + (bias[None] * LOG2E if use_bias else 0.0)).abs().amax(dim=(0, 2, 3))
    assert bool((shift[:H] >= smax).all()), "ctclip_attn_shift is not a bound of the |log2-logits|"
    o = torch.empty(nseq * n, ld, device=DEV, dtype=torch.bfloat16)
    lse = torch.empty(nseq, H, n, device=DEV)
    hip.attn_hm_fwd(q_hm, k_hm, v_hm, o, lse, bias, shift, nseq, n, H, ld)
    un = lambda t: t.float().reshape(nseq, n, H, D).permute(0, 2, 1, 3)
    check("hm attn out (bounded logits, no maximum)", un(o), oref, 2e-2)
    check("hm attn lse", lse, lse_ref, 2e-3)
    for label, sh_arg in (("no shift given", None), ("shift flagged unsafe", torch.cat((shift[:H], torch.ones(1, device=DEV))))):
        o2, lse2 = torch.full_like(o, 3.0), torch.empty_like(lse)
        hip.attn_hm_fwd(q_hm, k_hm, v_hm, o2, lse2, bias, sh_arg, nseq, n, H, ld)
        check(f"hm attn out (online softmax: {label})", un(o2), oref, 2e-2)
        check(f"hm attn lse (online softmax: {label})", lse2, lse_ref, 2e-3)
        check("static vs online out", o2, o.float(), 1e-2)

    oref.backward(do.float().reshape(nseq, n, H, D).permute(0, 2, 1, 3))
    dq, dk, dv = (torch.empty(nseq * n, ld, device=DEV, dtype=torch.bfloat16) for _ in range(3))
    delta = torch.empty(nseq, H, n, device=DEV)
    hip.attn_hm_bwd(q_hm, k_hm, v_hm, o, do_hm, lse, delta, dq, dk, dv, bias, None, None, None, 0, 0, 0, nseq, n, H,
                    ld, ld, ld, ld)
    check("hm attn dq", un(dq), qr.grad, 3e-2)
    check("hm attn dk", un(dk), kr.grad, 3e-2)
    check("hm attn dv", un(dv), vr.grad, 3e-2)
    check("hm attn delta", delta, (un(do) * un(o)).sum(-1), 2e-2)
    if not use_bias:
        return
    dq0, dk0, dv0 = dq.clone(), dk.clone(), dv.clone()
    dense = torch.zeros(H, n, n, device=DEV)
    hip.attn_hm_bwd(q_hm, k_hm, v_hm, o, do_hm, lse, delta, dq, dk, dv, bias, dense, None, None, 0, 0, 0, nseq, n, H,
                    ld, ld, ld, ld)
    check("hm attn dbias dense", dense, br.grad, 3e-2)
    check("hm attn dq (d(bias) pass)", un(dq), qr.grad, 3e-2)
    assert torch.equal(dk, dk0) and torch.equal(dv, dv0)
    R = 37
    relidx = torch.randint(0, R, (n, n), generator=torch.Generator().manual_seed(3)).to(DEV)
    dt = torch.zeros(H, R, device=DEV)
    hip.attn_hm_bwd(q_hm, k_hm, v_hm, o, do_hm, lse, delta, dq, dk, dv, bias, None, relidx.to(torch.uint16), dt, R, 0, 0,
                    nseq, n, H, ld, ld, ld, ld)
    check("hm attn dbias table", dt, torch.zeros(H, R, device=DEV).index_add_(1, relidx.reshape(-1), br.grad.reshape(H, -1)), 3e-2)
    gh = next(d for d in (24, 8, 5, 4, 3, 2, 1) if n % d == 0)
    gw = n // gh
    ii = torch.arange(n, device=DEV)
    rel2 = ((ii[:, None] // gw - ii[None] // gw + gh - 1) * (2 * gw - 1) + (ii[:, None] % gw - ii[None] % gw + gw - 1))
    R2 = (2 * gh - 1) * (2 * gw - 1)
    dt2 = torch.zeros(H, R2, device=DEV)
    hip.attn_hm_bwd(q_hm, k_hm, v_hm, o, do_hm, lse, delta, dq, dk, dv, bias, None, None, dt2, R2, gh, gw, nseq, n, H,
                    ld, ld, ld, ld)
    check("hm attn dbias 2-D grid table", dt2, torch.zeros(H, R2, device=DEV).index_add_(1, rel2.reshape(-1), br.grad.reshape(H, -1)), 3e-2)


def test_attention_shift_flags_wide_bounds(hip):
    """ctclip_attn_shift: learned scales far beyond initialisation (2 * qk bound + bias range > 100 binades) set the flag
    that sends the launch pair to the online-softmax kernel."""
    D, H = 32, 2
    shift = torch.empty(H + 1, device=DEV)
    ones = torch.ones(D, device=DEV)
    table = rnd(50, H, seed=110)                                            # [R, heads]: element stride = heads
    hip.attn_shift(ones, ones, D, 8.0 * 1.4426950408889634, table, 50, 1, H, H, shift)
    assert float(shift[H]) == 0.0
    want = 8.0 * 1.4426950408889634 * 1.02 + 0.25 + table.abs().amax(0) * 1.4426950408889634
    check("shift from a [R, heads] table", shift[:H], want, 1e-6)
    hip.attn_shift(ones * 3.0, ones * 2.0, D, 8.0 * 1.4426950408889634, table, 50, 1, H, H, shift)   # qk bound ~71 binades
    assert float(shift[H]) == 1.0


# ---------------------------------------------------------------------------------------------- dropout (text encoder)
def test_dropout_kernels_share_one_counter_based_stream(hip):
    """ctclip_dropout_keep / _add / _bwd evaluate the same keep(seed, offset + i): forward, backward and the materialised
    flags agree element for element, different offsets and seeds give different flags, the keep rate is 1 - p."""
    n, p, seed, off = 1 << 20, 0.1, 987654321, 7 << 40
    keep = torch.empty(n, dtype=torch.uint8, device=DEV)
    hip.dropout_keep(keep, n, p, seed, off)
    rate = float(keep.float().mean())
    assert abs(rate - (1 - p)) < 5 * math.sqrt(p * (1 - p) / n), rate
    x, br, g = rnd(n, seed=70), rnd(n, seed=71), rnd(n, seed=72)
    out = torch.empty_like(x)
    hip.dropout_add(x, br, out, n, p, seed, off)
    assert torch.allclose(out, x + keep.float() * (br * (1.0 / (1.0 - p))), rtol=1e-6, atol=1e-6)
    assert torch.equal(out[keep == 0], x[keep == 0])
    d32 = torch.empty_like(g)
    d16 = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    hip.dropout_bwd(g, d32, d16, n, p, seed, off)
    assert torch.allclose(d32, keep.float() * (g * (1.0 / (1.0 - p))), rtol=1e-6, atol=0) and bool((d32[keep == 0] == 0).all())
    assert torch.equal(d16, d32.to(torch.bfloat16))
    for s2, o2 in ((seed + 1, off), (seed, off + n)):
        other = torch.empty_like(keep)
        hip.dropout_keep(other, n, p, s2, o2)
        agree = float((other == keep).float().mean())
        assert abs(agree - (p * p + (1 - p) * (1 - p))) < 0.01, agree     # independent streams agree by chance only
    # a sub-range of the counter space is the same stream: chunked evaluation is consistent
    part = torch.empty(1000, dtype=torch.uint8, device=DEV)
    hip.dropout_keep(part, 1000, p, seed, off + 12345)
    assert torch.equal(part, keep[12345:13345])
    zero = torch.empty(4096, dtype=torch.uint8, device=DEV)
    hip.dropout_keep(zero, 4096, 0.0, seed, off)
    assert bool(zero.all())


# ---------------------------------------------------------------------------------------------- volume ingest
@pytest.mark.parametrize("H,W,D,xy,z,target,i16", [
    (40, 36, 22, 0.9, 2.4, (48, 40, 30), False),     # up-sampling in every axis; crop in H/W, pad in D
    (30, 34, 50, 0.6, 1.0, (32, 20, 40), True),      # down-sampling; pad in H, crop in W and D; int16 on the wire
    (16, 16, 16, 0.75, 1.5, (16, 16, 16), False),    # identity spacing and shape
])
def test_volume_ingest_vs_reference_pipeline(hip, H, W, D, xy, z, target, i16):
    """utils.preprocess.process_volume / ctclip_ingest_volume against the oracle restatement of the reference's tensor
    pipeline (src/utils/preprocess.py:123-152: HU rescale, permute, trilinear resample, clamp/1000, crop-or-pad with -1)."""
    from oracle import ctclip_oracle as O
    from utils.preprocess import process_volume
    gen = torch.Generator().manual_seed(60)
    raw = torch.randint(-1200, 2500, (H, W, D), generator=gen).float()
    slope, intercept = 1.0, -1024.0
    if i16:
        raw = raw.to(torch.int16)
    ref = O.preprocess_volume(raw.float(), slope, intercept, xy, z, target_shape=target).to(DEV)
    out32 = process_volume(raw, slope, intercept, xy, z, target_shape=target, out_dtype=torch.float32, device=DEV)
    assert tuple(out32.shape) == tuple(ref.shape) == (1, target[2], target[0], target[1])
    check("ingest f32", out32, ref, 2e-6)
    out16 = process_volume(raw, slope, intercept, xy, z, target_shape=target, device=DEV)
    assert out16.dtype == torch.bfloat16
    check("ingest bf16", out16, ref, 4e-3)
    pad_ref = (ref == -1).float().mean()
    assert abs(float((out32 == -1).float().mean()) - float(pad_ref)) < 1e-3      # same padded region


# ---------------------------------------------------------------------------------------------- patch embed
@pytest.mark.parametrize("geom", [
    (2, 1, 8, 12, 16, 4, 4),
    (2, 1, 4, 4, 8, 2, 2),       # tiny: 8 features = ONE 16-byte chunk per A row, and (bf16) one 16-byte vector per LDS row --
    (2, 1, 4, 4, 4, 2, 2),       # ... (f32, Wx = 4): both multiply-high divisors are 1, which has no 32-bit magic
    (1, 1, 20, 40, 80, 10, 20),  # the production tubelet (p = 20, pt = 10) on a small volume: divisors 20 / 200 / 30 / 500
])
@pytest.mark.parametrize("in16", [False, True])
def test_patch_ln_forward_and_volume_gradient(hip, in16, geom):
    """reference src/utils/ctvit.py:44-49 (Rearrange + LayerNorm over the tubelet) and its gradient w.r.t. the VOLUME
    (ctclip_patch_ln_bwd_dx; used by integrated gradients only), against einops-free torch on the same inputs."""
    B, C, Dz, Hy, Wx, pt, p = geom
    F_ = C * pt * p * p
    vol = rnd(B, C, Dz, Hy, Wx, seed=50)
    if in16:
        vol = bf(vol)
    gm, bt = (1 + 0.3 * rnd(F_, seed=51)), 0.2 * rnd(F_, seed=52)
    vr = vol.float().clone().requires_grad_(True)
    t, h, w = Dz // pt, Hy // p, Wx // p
    rows = vr.view(B, C, t, pt, h, p, w, p).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(B * t * h * w, F_)
    ref = torch.nn.functional.layer_norm(rows, (F_,), gm, bt, 1e-5)
    M, ldA = B * t * h * w, F_
    A = torch.empty(M, ldA, device=DEV, dtype=torch.bfloat16)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    hip.patch_ln_fwd(vol, int(in16), gm, bt, A, mean, rstd, B, C, Dz, Hy, Wx, pt, p, ldA, 1e-5)
    check("patch ln A", A, ref, 1e-2)
    # gamma == NULL: the plain normalised rows (what the folded projection consumes), bit-equal to gamma = 1, beta = 0
    A0, A1 = torch.empty_like(A), torch.empty_like(A)
    hip.patch_ln_fwd(vol, int(in16), None, None, A0, mean, rstd, B, C, Dz, Hy, Wx, pt, p, ldA, 1e-5)
    hip.patch_ln_fwd(vol, int(in16), torch.ones(F_, device=DEV), torch.zeros(F_, device=DEV), A1, mean, rstd, B, C, Dz, Hy, Wx, pt, p,
                     ldA, 1e-5)
    assert torch.equal(A0, A1)
    check("patch ln A (no affine)", A0, torch.nn.functional.layer_norm(rows.detach(), (F_,), None, None, 1e-5), 1e-2)
    dA = bf(rnd(M, ldA, seed=53))
    ref.backward(dA.float())
    dvol = torch.empty(B, C, Dz, Hy, Wx, device=DEV)
    hip.patch_ln_bwd_dx(vol, int(in16), dA, ldA, gm, mean, rstd, dvol, B, C, Dz, Hy, Wx, pt, p)
    check("patch ln d(volume)", dvol, vr.grad, 2e-5)


def tubelet_rows(vol, pt, p):
    """[B,C,Dz,Hy,Wx] -> [tokens, C pt p p] in the reference's feature order (ctvit.py:45: 'b c (t pt) (h p1) (w p2) -> b t h w (c pt p1 p2)')"""
    B, C, Dz, Hy, Wx = vol.shape
    t, h, w = Dz // pt, Hy // p, Wx // p
    return vol.reshape(B, C, t, pt, h, p, w, p).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(B * t * h * w, C * pt * p * p)


@pytest.mark.parametrize("geom", [(1, 1, 20, 40, 480, 10, 20),     # production tubelet (F = 4000, 24 tokens per row), 96 tokens: one ragged tile
                                  (2, 1, 40, 80, 480, 10, 20),     # 768 tokens = 6 whole tiles, two volumes
                                  (3, 2, 16, 32, 64, 8, 8),        # two channels, F = 1024, 192 tokens (1.5 tiles)
                                  (1, 1, 32, 48, 48, 16, 16)])     # F = 4096: the largest piece table, 18 tokens
def test_patch_embed_fused_forward(hip, geom):
    """ctclip_patch_embed_fused (csrc/patch_gemm.hip): Rearrange + LayerNorm(F) + Linear(F, 512) of reference ctvit.py:44-50 with
    the MFMA operand built from the raw voxels inside the GEMM -- against f32 torch (LayerNorm over the tubelet rows, then the
    folded bf16 weight), the row statistics against torch's, and CONSTANT tubelets (air = -1 padding: the reference's xhat is
    exactly 0 there) must come out as the folded bias, bit for bit."""
    B, C, Dz, Hy, Wx, pt, p = geom
    N, F_ = 512, C * pt * p * p
    vol = bf((rnd(B, C, Dz, Hy, Wx, seed=400) * 0.5 + 0.1).clamp(-1, 1))
    vol[0, :, :pt, :p, : 3 * p] = -1.0                               # three constant tubelets
    vol[-1, :, -pt:, -p:, -p:] = 0.375                               # and the very last token
    W, b = rnd(N, F_, seed=401) * (F_ ** -0.5), rnd(N, seed=402) * 0.05
    gm, bt = 1 + 0.2 * rnd(F_, seed=403), 0.1 * rnd(F_, seed=404)
    Wg, bfold = torch.empty(N, F_, device=DEV, dtype=torch.bfloat16), torch.empty(N, device=DEV)
    hip.patch_affine_fold(W, b, gm, bt, Wg, bfold, N, F_, F_)
    wsum = Wg.float().sum(1).contiguous()
    rows = tubelet_rows(vol.float(), pt, p)
    M = rows.shape[0]
    xhat = torch.nn.functional.layer_norm(rows, (F_,), None, None, 1e-5)
    ref = xhat @ Wg.float().t() + bfold
    Z = torch.full((M, N), 7.0, device=DEV)
    tstat = torch.empty(M, 4, device=DEV)
    hip.patch_embed_fused(vol, Wg, F_, wsum, bfold, Z, N, tstat, B, C, Dz, Hy, Wx, pt, p, N, 1e-5)
    cbase, mup, rstd, mean = tstat.unbind(1)
    assert torch.equal(mean, cbase + mup)
    check(f"fused tubelet embedding {geom}", Z, ref, 1e-2)
    check("tubelet mean", mean, rows.mean(1), 1e-5)
    check("tubelet rstd", rstd, (rows.var(1, unbiased=False) + 1e-5).rsqrt(), 1e-4)
    const = rows.std(1) == 0
    assert int(const.sum()) >= 4
    assert torch.equal(Z[const], bfold[None].expand(int(const.sum()), N))           # xhat = 0 exactly: z is the folded bias
    assert torch.equal(cbase[const], rows[const][:, 0]) and torch.equal(mean[const], rows[const][:, 0])
    # against the unfused chain on the same operands (gather + LayerNorm kernel -> bf16 operand -> GEMM): same rounding budget
    A = torch.empty(M, F_, device=DEV, dtype=torch.bfloat16)
    m2, r2 = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    hip.patch_ln_fwd(vol, 1, None, None, A, m2, r2, B, C, Dz, Hy, Wx, pt, p, F_, 1e-5)
    Z2 = torch.empty(M, N, device=DEV)
    hip.gemm_bf16(A, Wg, Z2, bfold, None, M, N, F_, F_, F_, N, 0, 1, 1, 1, 1, 0, 1.0, 0)
    e_fused, e_chain = relerr(Z, ref), relerr(Z2, ref)
    print(f"  error vs f32: fused {e_fused:.3e}, unfused chain {e_chain:.3e}")
    assert e_fused <= 1.5 * e_chain + 1e-4
    # geometry the kernel does not take: refused before anything is launched
    with pytest.raises(RuntimeError):
        hip.patch_embed_fused(vol, Wg, F_, wsum, bfold, Z, N, tstat, B, C, Dz, Hy, Wx, pt, p, 256, 1e-5)


@pytest.mark.parametrize("geom", [(4, 1, 20, 40, 480, 10, 20),     # production tubelet: F + 2 = 4002 columns = 16 feature windows (the last one ragged, with
                                  #                                  the two virtual columns), 384 tokens = 12 K-steps
                                  (8, 1, 40, 80, 480, 10, 20),     # 3072 tokens: the tokens are split over several workgroups per tile
                                  (6, 2, 16, 32, 64, 8, 8),        # two channels, F = 1024 (the virtual columns open a window of their own)
                                  (32, 1, 32, 48, 48, 16, 16),     # F = 4096, 576 tokens
                                  (2, 1, 240, 480, 480, 10, 20)])  # two production volumes: 27 648 tokens, 864 K-steps in 8 splits
def test_patch_wgrad_fused(hip, geom):
    """ctclip_patch_wgrad_fused + ctclip_patch_affine_bwd(ncorr = 2): the tubelet projection's weight-gradient product with the
    normalised operand rebuilt from the volume (reference ctvit.py:49-50 backward) -- G = dz^T xhat against f32 torch, then d(W),
    d(gamma), d(beta) against autograd of the unfolded expression; constant tubelets contribute exactly nothing; two runs agree
    bit for bit (partial tiles summed in split order)."""
    B, C, Dz, Hy, Wx, pt, p = geom
    N, F_ = 512, C * pt * p * p
    vol = bf((rnd(B, C, Dz, Hy, Wx, seed=410) * 0.5 + 0.1).clamp(-1, 1))
    vol[0, :, :pt, :p, : 2 * p] = -1.0
    W, b = rnd(N, F_, seed=411) * (F_ ** -0.5), rnd(N, seed=412) * 0.05
    gm, bt = 1 + 0.2 * rnd(F_, seed=413), 0.1 * rnd(F_, seed=414)
    Wg, bfold = torch.empty(N, F_, device=DEV, dtype=torch.bfloat16), torch.empty(N, device=DEV)
    hip.patch_affine_fold(W, b, gm, bt, Wg, bfold, N, F_, F_)
    rows = tubelet_rows(vol.float(), pt, p)
    M = rows.shape[0]
    Z, tstat = torch.empty(M, N, device=DEV), torch.empty(M, 4, device=DEV)
    hip.patch_embed_fused(vol, Wg, F_, Wg.float().sum(1).contiguous(), bfold, Z, N, tstat, B, C, Dz, Hy, Wx, pt, p, N, 1e-5)
    dz = bf(rnd(M, N, seed=415))
    xhat = torch.nn.functional.layer_norm(rows, (F_,), None, None, 1e-5)
    G_ref = dz.float().t() @ xhat
    Gx = torch.zeros(N, F_ + 2, device=DEV)
    hip.patch_wgrad_fused(vol, dz, N, tstat, Gx, F_ + 2, B, C, Dz, Hy, Wx, pt, p, N)
    G = Gx[:, :F_] - (Gx[:, F_] + Gx[:, F_ + 1])[:, None]
    check(f"G = dz^T xhat from the volume {geom}", G, G_ref, 1e-2)
    Gx2 = torch.zeros_like(Gx)
    hip.patch_wgrad_fused(vol, dz, N, tstat, Gx2, F_ + 2, B, C, Dz, Hy, Wx, pt, p, N)
    assert torch.equal(Gx, Gx2)
    # constant tubelets: zero rows of the operand -- dropping them from dz changes nothing
    const = rows.std(1) == 0
    assert int(const.sum()) >= 2
    dz0 = dz.clone()
    dz0[const] = 0
    Gx3 = torch.zeros_like(Gx)
    hip.patch_wgrad_fused(vol, dz0, N, tstat, Gx3, F_ + 2, B, C, Dz, Hy, Wx, pt, p, N)
    assert torch.equal(Gx3, Gx)
    # the parameter gradients out of it, against autograd of z = (xhat gamma + beta) W^T + b
    Wr, gr, btr = (t.clone().requires_grad_(True) for t in (W, gm, bt))
    ((xhat * gr + btr) @ Wr.t()).backward(dz.float())
    db = dz.float().sum(0)
    dW, dg, dbt = torch.zeros(N, F_, device=DEV), torch.zeros(F_, device=DEV), torch.zeros(F_, device=DEV)
    hip.patch_affine_bwd(Gx, db, W, gm, bt, dW, dg, dbt, N, F_, F_ + 2, 2)
    check("d(W)", dW, Wr.grad, 1e-2)
    check("d(gamma)", dg, gr.grad, 1.5e-2)
    check("d(beta)", dbt, btr.grad, 1e-4)


@pytest.mark.parametrize("N,F_,M", [(24, 40, 96), (64, 4000, 200), (16, 8, 50)])
def test_patch_affine_fold_and_backward(hip, N, F_, M):
    """ctclip_patch_affine_fold / _bwd: LayerNorm(F)'s gamma / beta folded into the tubelet projection (reference
    src/utils/ctvit.py:49-50).  Forward identity and the three parameter gradients against torch autograd of the UNFOLDED
    expression z = (xhat * gamma + beta) W^T + b on the same xhat, dz (all f32: the algebra is what is checked)."""
    xhat = rnd(M, F_, seed=60)
    W, b = rnd(N, F_, seed=61) * 0.1, rnd(N, seed=62)
    gm, bt = (1 + 0.3 * rnd(F_, seed=63)), 0.2 * rnd(F_, seed=64)
    Wr, br, gr, btr = (t.clone().requires_grad_(True) for t in (W, b, gm, bt))
    z = (xhat * gr + btr) @ Wr.t() + br
    dz = rnd(M, N, seed=65)
    z.backward(dz)
    ldw = (F_ + 7) // 8 * 8
    Wg = torch.full((N, ldw), 7.0, device=DEV, dtype=torch.bfloat16)
    bfold = torch.empty(N, device=DEV)
    hip.patch_affine_fold(W, b, gm, bt, Wg, bfold, N, F_, ldw)
    check("folded weight", Wg[:, :F_], W * gm, 4e-3)
    assert float(Wg[:, F_:].float().abs().max()) == 0.0 if ldw > F_ else True
    check("folded bias", bfold, b + W @ bt, 1e-5)
    check("folded forward", xhat @ (W * gm).t() + (b + W @ bt), z.detach(), 1e-5)
    G, db = dz.t() @ xhat, dz.sum(0)
    dW, dg, dbt = torch.ones(N, F_, device=DEV), torch.ones(F_, device=DEV), torch.ones(F_, device=DEV)   # accumulate on top
    hip.patch_affine_bwd(G.contiguous(), db, W, gm, bt, dW, dg, dbt, N, F_, F_, 0)
    check("d(W)", dW - 1, Wr.grad, 1e-5)
    check("d(gamma)", dg - 1, gr.grad, 1e-5)
    check("d(beta)", dbt - 1, btr.grad, 1e-5)


# ---------------------------------------------------------------------------------------------- PEG
@pytest.mark.parametrize("B,T,H,W,d", [
    (2, 5, 24, 24, 32),      # the CT-ViT plane: 4 strips of 6 per row, 384 threads
    (3, 1, 4, 7, 16),        # single time step, ragged strip (7 = 6 + 1); half of the block's one wave is idle
    (2, 4, 4, 4, 512),       # the 64^3 debug volume's grid: 32 of 64 threads idle in every workgroup
    (3, 24, 24, 24, 64),     # the CT-ViT grid: 24 planes of 24 x 24, several batch items per workgroup chunk
    (1, 2, 3, 5, 48),        # T = 2: the backward drain stores both remaining planes
    (2, 4, 9, 13, 16),       # odd plane
    (1, 3, 30, 30, 16),      # plane too large for one workgroup -> generic sweep kernel
    (2, 3, 4, 5, 8),         # d/4 not a multiple of the channel slice -> generic sweep kernel
])
@pytest.mark.parametrize("residual", [0, 1])
def test_peg_kernels(hip, B, T, H, W, d, residual):
    """reference src/utils/attention.py:55-83 (+ the residual of :325): causal depthwise 3x3x3 convolution"""
    x = rnd(B, T, H, W, d, seed=40)
    wt = rnd(d, 1, 3, 3, 3, seed=41) * 0.3
    bias = rnd(d, seed=42)
    xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    xp = torch.nn.functional.pad(xr.permute(0, 4, 1, 2, 3), (1, 1, 1, 1, 2, 0))
    ref = torch.nn.functional.conv3d(xp, wr, br, groups=d).permute(0, 2, 3, 4, 1)
    if residual:
        ref = ref + xr
    w27 = wt.reshape(d, 27).t().contiguous()
    y = torch.empty_like(x)
    y16 = torch.empty(x.shape, device=DEV, dtype=torch.bfloat16)
    hip.peg_fwd(x, w27, bias, y, y16, B, T, H, W, d, residual)
    check("peg y", y, ref, 1e-5)
    check("peg y16", y16, ref, 1e-2)
    dy = rnd(B, T, H, W, d, seed=43)
    ref.backward(dy)
    dx = torch.empty_like(x)
    dx16 = torch.empty(x.shape, device=DEV, dtype=torch.bfloat16)
    hip.peg_bwd_data(dy, w27, dx, dx16, B, T, H, W, d, residual)
    check("peg dx", dx, xr.grad, 1e-5)
    check("peg dx16", dx16, xr.grad, 1e-2)
    dw27 = torch.zeros(27, d, device=DEV)
    db = torch.zeros(d, device=DEV)
    # LDS is not cleared between kernels: the weight-gradient kernels round their blocks up to whole waves, and an idle thread's
    # zero x times whatever a stray LDS word holds must not reach the sums (round 5: two processes on one device left NaN patterns
    # there and the step's loss went NaN for grids whose rows x strips x 8 is no multiple of 64) -- poison the LDS first
    sink = torch.zeros(4, device=DEV, dtype=torch.int32)
    hip.probe_lds_fill(0x7FC00000, sink)
    hip.peg_bwd_weight(dy, x, dw27, db, B, T, H, W, d)
    check("peg dw", dw27.t().reshape(d, 1, 3, 3, 3), wr.grad, 1e-4)
    check("peg db", db, br.grad, 1e-4)
    # both gradients in ONE pass over dy (ctclip_peg_bwd_fused): the same results, for the grids its plane tiling takes
    from ctclip_hip import ops
    dx2, dx2_16 = torch.full_like(x, 7.0), torch.empty(x.shape, device=DEV, dtype=torch.bfloat16)
    dw2, db2 = torch.zeros(27, d, device=DEV), torch.zeros(d, device=DEV)
    if ops.peg_fused_ok(H, W, d):
        hip.probe_lds_fill(0x7FC00000, sink)
        hip.peg_bwd_fused(dy, x, w27, dx2, dx2_16, dw2, db2, B, T, H, W, d, residual)
        check("fused peg dx", dx2, xr.grad, 1e-5)
        check("fused peg dx16", dx2_16, xr.grad, 1e-2)
        check("fused peg dw", dw2.t().reshape(d, 1, 3, 3, 3), wr.grad, 1e-4)
        check("fused peg db", db2, br.grad, 1e-4)
        dw3, db3 = torch.zeros_like(dw2), torch.zeros_like(db2)
        hip.peg_bwd_fused(dy, x, w27, dx2, None, dw3, db3, B, T, H, W, d, residual)
        assert torch.equal(dw3, dw2) and torch.equal(db3, db2)                 # two-stage sums: bit-reproducible
    else:
        with pytest.raises(RuntimeError):
            hip.peg_bwd_fused(dy, x, w27, dx2, dx2_16, dw2, db2, B, T, H, W, d, residual)


# ---------------------------------------------------------------------------------------------- elementwise
def test_elementwise(hip):
    rows, I = 50, 152
    h = bf(rnd(rows, 2 * I, seed=30))
    val, gate = h.float()[:, :I].clone().requires_grad_(True), h.float()[:, I:].clone().requires_grad_(True)
    ref = torch.nn.functional.gelu(gate) * val
    g = torch.empty(rows, I, device=DEV, dtype=torch.bfloat16)
    hip.geglu_fwd(h, g, rows, I, I, 2 * I, I)
    check("geglu", g, ref, 1e-2)
    dg = bf(rnd(rows, I, seed=31))
    ref.backward(dg.float())
    dh = torch.empty_like(h)
    hip.geglu_bwd(dg, h, dh, rows, I, I, I, 2 * I)
    check("geglu dval", dh[:, :I], val.grad, 1e-2)
    check("geglu dgate", dh[:, I:], gate.grad, 1e-2)

    x = h.float().clone().requires_grad_(True)
    r = torch.nn.functional.gelu(x)
    m = torch.empty_like(h)
    hip.gelu_fwd(h, m, h.numel())
    check("gelu", m, r, 1e-2)
    r.backward(torch.ones_like(r))
    dm = bf(torch.ones_like(r))
    hip.gelu_bwd(dm, h, dh, h.numel())
    check("gelu bwd", dh, x.grad, 1e-2)

    t = rnd(3, 4, 5, 8, seed=32)
    out = torch.empty(3, 5, 4, 8, device=DEV)
    hip.swap_middle_f32(t, out, 3, 4, 5, 8)
    assert torch.equal(out, t.permute(0, 2, 1, 3).contiguous())
    y32 = torch.empty(3, 40, device=DEV)
    y16 = torch.empty(3, 40, device=DEV, dtype=torch.bfloat16)
    hip.mean_mid_fwd(t, y16, y32, 3, 4, 40)
    check("mean", y32, t.reshape(3, 4, 40).mean(1), 1e-6)
    dx = torch.empty(3, 4, 40, device=DEV)
    hip.mean_mid_bwd(y32, dx, 3, 4, 40)
    check("mean bwd", dx, (y32 / 4)[:, None].expand(3, 4, 40), 1e-6)
    a, b = rnd(64, seed=33), rnd(64, seed=34)
    s32, s16 = torch.empty(64, device=DEV), torch.empty(64, device=DEV, dtype=torch.bfloat16)
    hip.add_f32(a, b, s32, s16, 64)
    assert torch.equal(s32, a + b)
    c16 = torch.empty(64, device=DEV, dtype=torch.bfloat16)
    hip.cast_f32_bf16(a, c16, 64)
    assert torch.equal(c16, a.to(torch.bfloat16))


# ---------------------------------------------------------------------------------------------- reproducibility
def test_two_stage_reductions_are_bitwise_reproducible(hip):
    """include/ctclip_hip.h, "reproducibility": every entry point with a `partials` argument gives the same bits on every
    run (the reference's attribution code asks for deterministic algorithms, src/utils/visualizations.py:29-39).  Each is
    run several times on the same inputs -- with unrelated work in between to perturb workgroup scheduling -- and compared
    with torch.equal; values are checked against f32 torch."""
    rows, dim = 5000, 512
    x, dy = rnd(rows, dim, seed=70), rnd(rows, dim, seed=71)
    gm = 1 + 0.2 * rnd(dim, seed=72)
    mean, rstd = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    y = torch.empty(rows, dim, device=DEV)
    hip.layernorm_fwd(x, gm, None, None, y, mean, rstd, rows, dim, 1e-5)
    noise = rnd(1 << 20, seed=73)
    out_ref = {}

    def run_all():
        out = {}
        dg, db, dx = torch.zeros(dim, device=DEV), torch.zeros(dim, device=DEV), torch.empty(rows, dim, device=DEV)
        hip.layernorm_bwd(dy, x, gm, mean, rstd, None, dx, None, dg, db, rows, dim)
        out["ln dgamma"], out["ln dbeta"], out["ln dx"] = dg, db, dx
        noise.mul_(1.0)                                                   # something else on the stream
        H, D = 8, 32
        q, dq = bf(rnd(rows, H * D, seed=74)), bf(rnd(rows, H * D, seed=75))
        sc = 1 + 0.1 * rnd(D, seed=76)
        qn, inv = torch.empty_like(q), torch.empty(rows, H, device=DEV)
        hip.headnorm_fwd(q, sc, qn, inv, rows, H, D, H * D, H * D, 8.0, 0, 0)
        dxh, ds = torch.empty_like(q), torch.zeros(D, device=DEV)
        hip.headnorm_bwd(dq, q, inv, sc, dxh, ds, rows, H, D, H * D, H * D, H * D, 8.0, 0, 0)
        out["headnorm dscale"] = ds
        cs = torch.zeros(dim, device=DEV)
        hip.colsum_accum(dy, 0, rows, dim, dim, cs)
        out["colsum f32"] = cs
        cs16 = torch.zeros(dim, device=DEV)
        hip.colsum_accum(bf(dy), 1, rows, dim, dim, cs16)
        out["colsum bf16"] = cs16
        ss, dt = torch.zeros((), device=DEV), torch.zeros((), device=DEV)
        hip.sumsq_accum(x.reshape(-1), x.numel(), ss)
        hip.dot_accum(x.reshape(-1), dy.reshape(-1), dt, x.numel())
        out["sumsq"], out["dot"] = ss, dt
        # split-K weight gradient (m-major x n-major, K = tokens) with the workspace: partial tiles + ordered sum
        tok, nf, kf = 8192, 256, 320
        gy, gx = bf(rnd(tok, nf, seed=79)), bf(rnd(tok, kf, seed=80))
        dwt = torch.zeros(nf, kf, device=DEV)
        hip.gemm_bf16(gy, gx, dwt, None, None, nf, kf, tok, nf, kf, kf, 0, 0, 0, 1, 16, 1, 1.0, 0)
        out["split-K wgrad"] = dwt
        if "ref wgrad" not in out_ref:
            out_ref["ref wgrad"] = gy.float().t() @ gx.float()
            atom = torch.zeros(nf, kf, device=DEV)                         # the same product without a workspace: f32 atomics
            hip.gemm_bf16(gy, gx, atom, None, None, nf, kf, tok, nf, kf, kf, 0, 0, 0, 1, 16, 1, 1.0, 0, None, 0)
            out_ref["atomics wgrad"] = atom
        B, T, Hh, Ww, d = 3, 4, 24, 24, 32
        px, pdy = rnd(B, T, Hh, Ww, d, seed=77), rnd(B, T, Hh, Ww, d, seed=78)
        dw27, dbias = torch.zeros(27, d, device=DEV), torch.zeros(d, device=DEV)
        hip.peg_bwd_weight(pdy, px, dw27, dbias, B, T, Hh, Ww, d)
        out["peg dw"], out["peg dbias"] = dw27, dbias
        return out

    first = run_all()
    xh = (x - mean[:, None]) * rstd[:, None]
    check("ln dgamma value", first["ln dgamma"], (dy * xh).sum(0), 1e-5)
    check("colsum value", first["colsum f32"], dy.sum(0), 1e-5)
    check("sumsq value", first["sumsq"], (x.double() ** 2).sum().float(), 1e-5)
    check("dot value", first["dot"], (x.double() * dy.double()).sum().float(), 1e-4)
    check("split-K wgrad value (workspace)", first["split-K wgrad"], out_ref["ref wgrad"], 2e-3)
    check("split-K wgrad value (atomics)", out_ref["atomics wgrad"], out_ref["ref wgrad"], 2e-3)
    for rep in range(3):
        again = run_all()
        for k, v in first.items():
            assert torch.equal(v, again[k]), f"{k} differs between two runs of the same kernel on the same inputs"


def test_gelu_tail_saturates(hip):
    """The transcendental-free Phi(x) of the GELU epilogues (csrc/common.h) must reach 0 / 1 outside its fitted range: gelu(x)
    for x in [-8, -3.7] is 0 up to 4e-4 (the true values are below that), not a leak proportional to |x|; and on the whole
    real line the activation stays within 4e-4 of erf-GELU.  Checked through the elementwise kernels (bf16 in / out)."""
    x = torch.cat([torch.linspace(-8.0, -3.7, 2048), torch.linspace(-3.7, 3.7, 4096), torch.linspace(3.7, 8.0, 2048)]).to(DEV)
    h = bf(x)
    m = torch.empty_like(h)
    hip.gelu_fwd(h, m, h.numel())
    ref = torch.nn.functional.gelu(h.float())
    err = (m.float() - ref).abs()
    tail = h.float() < -3.7
    print(f"  gelu: max |err| {float(err.max()):.2e}; on x < -3.7: max |gelu| {float(m.float()[tail].abs().max()):.2e}")
    assert float(m.float()[tail].abs().max()) <= 4e-4
    assert float((err - 4e-3 * ref.abs()).max()) <= 4e-4                 # bf16 output rounding + the approximation
    big = h.float() > 3.7
    assert torch.equal(m[big], h[big])                                   # Phi = 1 exactly: gelu(x) = x


def test_layernorm_backward_from_normalised_rows(hip):
    """ctclip_layernorm_bwd_xhat: the LayerNorm backward when gamma has been folded into the following projection -- from the saved
    bf16 normalised rows alone (neither x nor the mean), against autograd through F.layer_norm without affine part on the same
    rows; plus the two residual-path terms.  ctclip_layernorm_fwd with gamma = NULL writes exactly those rows."""
    rows, dim = 777, 512
    x = rnd(rows, dim, seed=140) * 3 + 0.5
    xh16 = torch.empty(rows, dim, device=DEV, dtype=torch.bfloat16)
    mean, rstd = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    hip.layernorm_fwd(x, None, None, xh16, None, mean, rstd, rows, dim, 1e-5)
    xr = x.clone().requires_grad_(True)
    ref_y = torch.nn.functional.layer_norm(xr, (dim,), eps=1e-5)
    check("plain normalised rows", xh16, ref_y, 1e-2)
    dy = bf(rnd(rows, dim, seed=141))
    ref_y.backward(dy.float())
    dres, dres2 = rnd(rows, dim, seed=142), bf(rnd(rows, dim, seed=143))
    dx, dx16 = torch.empty(rows, dim, device=DEV), torch.empty(rows, dim, device=DEV, dtype=torch.bfloat16)
    hip.layernorm_bwd_xhat(dy, xh16, rstd, dres, dres2, dx, dx16, rows, dim)
    want = xr.grad + dres + dres2.float()
    check("dx from xhat", dx, want, 5e-3)                  # xhat enters its own backward rounded to bf16
    check("dx bf16 copy", dx16, want, 1e-2)
    hip.layernorm_bwd_xhat(dy, xh16, rstd, None, None, dx, None, rows, dim)
    check("dx, no residual terms", dx, xr.grad, 5e-3)


def test_attention_input_gradient_with_layernorm_backward_in_the_gemm(hip):
    """ctclip_headnorm_bwd_ln + ctclip_gemm_bf16_lnbwd: the input gradient of an attention block as ONE product
    [rstd dq | dk | dv] [Wqg ; Wkv] whose epilogue applies the LayerNorm backward, against f32 torch math of the chain it replaces
    (head-norm backward -> q / kv data gradients -> LayerNorm backward + residual terms; attention.py:140-153 backward).  Ragged
    row count, with and without the residual gradient and the bf16 copy."""
    M, dim, H, D = 1000, 512, 8, 32
    inner = H * D
    mult = 8.0 * 1.4426950408889634
    x = rnd(M, dim, seed=150) * 2 + 0.3
    gamma = 1 + 0.2 * rnd(dim, seed=151)
    wq, wkv = rnd(inner, dim, seed=152) * 0.05, rnd(2 * inner, dim, seed=153) * 0.05
    qs = 1 + 0.1 * rnd(D, seed=154)
    # forward quantities the way the block saves them
    xh16 = torch.empty(M, dim, device=DEV, dtype=torch.bfloat16)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    hip.layernorm_fwd(x, None, None, xh16, None, mean, rstd, M, dim, 1e-5)
    wqg = torch.empty(inner, dim, device=DEV, dtype=torch.bfloat16)
    hip.patch_affine_fold(wq, None, gamma, None, wqg, None, inner, dim, dim)
    q16 = bf(xh16.float() @ wqg.float().t())                                  # raw q as the projection GEMM rounds it
    qinv = 1.0 / q16.float().view(M, H, D).norm(dim=-1).clamp_min(1e-12)      # [M, H]
    dqh, dkv = bf(rnd(M, inner, seed=155)), bf(rnd(M, 2 * inner, seed=156))
    dres = rnd(M, dim, seed=157)
    # reference: f32 autograd of l2norm-and-scale, then the two data gradients and the LayerNorm backward
    qr = q16.float().view(M, H, D).clone().requires_grad_(True)
    (torch.nn.functional.normalize(qr, dim=-1) * qs * mult).backward(dqh.float().view(M, H, D))
    dq_ref = qr.grad.reshape(M, inner)
    dq16 = bf(dq_ref)
    dn = dq16.float() @ wqg.float()                                           # gradient w.r.t. xhat
    xh = xh16.float()
    ln = rstd[:, None] * (dn - dn.mean(1, keepdim=True) - xh * (dn * xh).mean(1, keepdim=True))
    want = ln + dkv.float() @ bf(wkv).float() + dres
    # fused path
    dcat = torch.empty(M, 3 * inner, device=DEV, dtype=torch.bfloat16)
    dcat[:, inner:] = dkv
    dq = torch.empty(M, inner, device=DEV, dtype=torch.bfloat16)
    gqs = torch.zeros(D, device=DEV)
    c1, c2 = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    wbar = wqg.float().sum(1).contiguous()
    hip.headnorm_bwd_ln(dqh, q16, qinv.contiguous(), qs, dq, gqs, M, H, D, inner, inner, inner, mult, rstd, wbar, dim, dcat,
                        3 * inner, c1, c2, 0, 0)
    check("dq (head-norm backward)", dq, dq_ref, 1e-2)
    # the same from the NORMALISED rows (x_normed = 1: what the block keeps when the q projection normalises in its epilogue),
    # row-major and head-major: dq, the scaled copy and both row constants must agree with the raw-row form
    nseq_, n_ = 10, 100
    qn16 = bf(torch.nn.functional.normalize(q16.float().view(M, H, D), dim=-1) * qs * mult).reshape(M, inner)
    for x_in, ldx, hmn in ((qn16, inner, 0), (to_hm(qn16, nseq_, n_, H, D), 0, n_)):
        dq_n, gqs_n = torch.empty_like(dq), torch.zeros(D, device=DEV)
        dcat_n, c1n, c2n = torch.empty_like(dcat), torch.empty(M, device=DEV), torch.empty(M, device=DEV)
        hip.headnorm_bwd_ln(dqh, x_in, qinv.contiguous(), qs, dq_n, gqs_n, M, H, D, inner, ldx, inner, mult, rstd, wbar, dim, dcat_n,
                            3 * inner, c1n, c2n, hmn, 1)
        check(f"dq from normalised rows (hm_n {hmn})", dq_n, dq_ref, 1e-2)
        check("d(q_scale) from normalised rows", gqs_n, gqs, 5e-3)
        check("rstd-scaled dq from normalised rows", dcat_n[:, :inner], rstd[:, None] * dq_n.float(), 1e-2)
        check("c1 from normalised rows", c1n, c1, 2e-2)
        assert float((c2n - c2).abs().max()) <= 2e-2 * float(c1.abs().max())       # c2 = rounding noise around 0 (dq is orthogonal to q)
    check("rstd-scaled dq", dcat[:, :inner], rstd[:, None] * dq.float(), 1e-2)
    wcat = torch.cat((wqg.t(), bf(wkv).t()), 1).contiguous()                  # [dim, 3 inner]
    dx, dx16 = torch.empty(M, dim, device=DEV), torch.empty(M, dim, device=DEV, dtype=torch.bfloat16)
    hip.gemm_bf16_lnbwd(dcat, wcat, dx, dx16, M, dim, 3 * inner, 3 * inner, 3 * inner, xh16, c1, c2, dres)
    check("dx = LN'(dq Wqg) + dkv Wkv + dres", dx, want, 6e-3)
    check("bf16 copy of dx", dx16, want, 1.2e-2)
    hip.gemm_bf16_lnbwd(dcat, wcat, dx, None, M, dim, 3 * inner, 3 * inner, 3 * inner, xh16, c1, c2, None)
    check("dx without the residual gradient", dx, want - dres, 6e-3)


def test_feed_forward_weight_gradient_out_of_the_blocked_order(hip):
    """ctclip_geglu_wgrad_unblock: rows of the FF1 weight-gradient product come in [value 32 | gate 32 | ...] blocks of the padded
    width; the nn.Linear(dim, 2 inner) gradient (value rows, then gate rows: attention.py:47) accumulates them -- exactly."""
    I, Ip, dim, blk = 85, 128, 24, 32
    gp = rnd(2 * Ip, dim, seed=170)
    dw0 = rnd(2 * I, dim, seed=171)
    want = dw0.clone()
    v = gp.view(Ip // blk, 2, blk, dim)
    want[:I] += v[:, 0].reshape(Ip, dim)[:I]
    want[I:] += v[:, 1].reshape(Ip, dim)[:I]
    dw = dw0.clone()
    hip.geglu_wgrad_unblock(gp, dw, I, blk, dim)
    assert torch.equal(dw, want)


def test_bert_embedding_backward_without_atomics(hip):
    """ctclip_bert_embed_bwd (transformers BertEmbeddings backward): d(word) / d(position) / d(token type) against
    torch's index_add on the same inputs -- ids with many repeats (and one id used by every row of a sequence), on top of
    non-zero gradients already in the buffers -- and bit-identical from run to run."""
    B, L, H, V, NT = 6, 40, 96, 50, 2
    rows = B * L
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, V, (rows,), generator=g)
    ids[L:2 * L] = 7                                                    # one id, a whole sequence long
    tt = torch.randint(0, NT, (rows,), generator=g)
    ids, tt = ids.to(DEV), tt.to(DEV)
    dy = rnd(rows, H, seed=120)
    base = {k: rnd(*shape, seed=s_) for k, shape, s_ in (("w", (V, H), 121), ("p", (64, H), 122), ("t", (NT, H), 123))}
    ref_w = base["w"].clone().index_add_(0, ids, dy)
    ref_p = base["p"].clone()
    ref_p[:L] += dy.reshape(B, L, H).sum(0)
    ref_t = base["t"].clone().index_add_(0, tt, dy)
    outs = []
    for _ in range(3):
        dw, dp, dt = base["w"].clone(), base["p"].clone(), base["t"].clone()
        hip.bert_embed_bwd(ids, tt, dy, dw, dp, dt, rows, L, H, NT, V)
        outs.append((dw, dp, dt))
    check("d(word)", outs[0][0], ref_w, 1e-5)
    check("d(position)", outs[0][1], ref_p, 1e-5)
    check("d(token type)", outs[0][2], ref_t, 1e-5)
    for o in outs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(o, outs[0]))
    dw, dp, dt = base["w"].clone(), base["p"].clone(), base["t"].clone()
    hip.bert_embed_bwd(ids, None, dy, dw, dp, dt, rows, L, H, NT, V)     # no token types given: all rows are type 0
    check("d(token type), tt = None", dt[0], base["t"][0] + dy.sum(0), 1e-5)
    assert torch.equal(dt[1], base["t"][1])


def test_bert_embedding_backward_at_the_reference_text_length(hip):
    """The reference tokenises to 512 tokens (CTClipTrainer max_text_length): 88 reports x 512 = 45 056 rows, more than any LDS
    list of rows could hold -- the word gradient walks the rows in windows.  Padding makes one id ([PAD]) tens of thousands of rows
    long (many windows), the rest is random over a BERT-sized vocabulary; against index_add, twice bit-identical."""
    B, L, H, V = 88, 512, 64, 30522
    rows = B * L
    g = torch.Generator().manual_seed(9)
    ids = torch.randint(1, V, (rows,), generator=g)
    lens = torch.randint(32, L + 1, (B,), generator=g)
    ids.view(B, L)[torch.arange(L)[None] >= lens[:, None]] = 0           # [PAD]
    ids = ids.to(DEV)
    dy = rnd(rows, H, seed=130)
    ref = torch.zeros(V, H, device=DEV, dtype=torch.float64).index_add_(0, ids, dy.double()).float()
    outs = []
    for _ in range(2):
        dw, dp, dt = torch.zeros(V, H, device=DEV), torch.zeros(L, H, device=DEV), torch.zeros(2, H, device=DEV)
        hip.bert_embed_bwd(ids, None, dy, dw, dp, dt, rows, L, H, 2, V)
        outs.append((dw, dp, dt))
    check("d(word), 45 056 rows", outs[0][0], ref, 2e-5)
    check("d(position)", outs[0][1], dy.view(B, L, H).double().sum(0).float(), 2e-5)
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
    with pytest.raises(RuntimeError):                                     # a word table larger than the scratch: refused before any launch
        hip.bert_embed_bwd(ids, None, dy, dw, dp, dt, rows, L, H, 2, 1 << 22)
    assert all(torch.equal(a, b) for a, b in zip((dw, dp, dt), outs[1]))


# ---------------------------------------------------------------------------------------------- ordered (reproducible) sums
@pytest.mark.parametrize("nseq,n,H,hm", [(37, 64, 2, True), (21, 40, 3, False), (12, 576, 8, True)])
def test_ordered_bias_gradient_is_reproducible_and_equals_the_fast_one(hip, nseq, n, H, hm):
    """ctclip_attn_dbias_ordered (csrc/attention_det.hip): the sum over the sequences of dS, every tile by one owner in
    sequence order -- equal to the dense d(bias) of the training kernels up to f32 summation order, bit-identical from run
    to run, on head-major and row-major operands, with a ragged last tile (n = 40); ctclip_attn_dbias_table gathers it into
    the 2-D relative-position table exactly like an index_add."""
    D, LOG2E, LN2 = 32, 1.4426950408889634, 0.6931471805599453
    ld = H * D
    unit = lambda t: torch.nn.functional.normalize(t.reshape(nseq * n, H, D), dim=-1).reshape(nseq * n, ld)
    q, k = bf(unit(rnd(nseq * n, ld, seed=130)) * 8.0 * LOG2E), bf(unit(rnd(nseq * n, ld, seed=131)))
    v, do = bf(rnd(nseq * n, ld, seed=132)), bf(rnd(nseq * n, ld, seed=133))
    bias = rnd(H, n, n, seed=134)
    o = torch.empty(nseq * n, ld, device=DEV, dtype=torch.bfloat16)
    lse, delta = torch.empty(nseq, H, n, device=DEV), torch.empty(nseq, H, n, device=DEV)
    dq, dk, dv = (torch.empty_like(o) for _ in range(3))
    fast = torch.zeros(H, n, n, device=DEV)
    if hm:
        qx, kx, vx, dox = (to_hm(t, nseq, n, H, D) for t in (q, k, v, do))
        hip.attn_hm_fwd(qx, kx, vx, o, lse, bias, None, nseq, n, H, ld)
        hip.attn_hm_bwd(qx, kx, vx, o, dox, lse, delta, dq, dk, dv, bias, fast, None, None, 0, 0, 0, nseq, n, H, ld, ld, ld, ld)
    else:
        qx, kx, vx, dox = q, k, v, do
        hip.attn_fwd(q, k, v, o, lse, bias, None, nseq, n, H, D, ld, ld, ld, ld, LN2)
        hip.attn_bwd(q, k, v, o, do, lse, delta, dq, dk, dv, bias, None, fast, None, None, 0, 0, 0, nseq, n, H, D,
                     ld, ld, ld, ld, ld, ld, ld, ld, LN2)
    runs = []
    for _ in range(3):
        dense = torch.zeros(H, n, n, device=DEV)
        hip.attn_dbias_ordered(qx, kx, vx, dox, lse, delta, bias, dense, nseq, n, H, int(hm), ld, ld, ld, ld, LN2)
        runs.append(dense)
    # the head-major training kernels round the bias to fp16 on its way into the score MFMA (2.4e-4 of each entry); the ordered
    # form and the row-major kernels add it in f32
    check("ordered vs fast d(bias)", runs[0], fast, 3e-3 if hm else 2e-5)
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])
    gh = next(d for d in (24, 8, 5, 4, 3, 2, 1) if n % d == 0)
    gw = n // gh
    ii = torch.arange(n, device=DEV)
    rel = (ii[:, None] // gw - ii[None] // gw + gh - 1) * (2 * gw - 1) + (ii[:, None] % gw - ii[None] % gw + gw - 1)
    R = (2 * gh - 1) * (2 * gw - 1)
    table = torch.ones(H, R, device=DEV)
    hip.attn_dbias_table(runs[0], table, H, gh, gw)
    check("table from dense", table - 1, torch.zeros(H, R, device=DEV).index_add_(1, rel.reshape(-1), runs[0].reshape(H, -1)), 1e-5)


@pytest.mark.parametrize("ntok,ncodes,dim,skew", [(5000, 64, 96, False), (3000, 16, 512, True), (700, 300, 32, False)])
def test_sorted_codebook_statistics(hip, ntok, ncodes, dim, skew):
    """ops.vq_ema_accum (ctclip_vq_ema_accum_sorted): per-code counts and sums of the normalised tokens from the stable sort
    by code -- codes inside one 256-row chunk, codes across several chunks (a code owning 2000 tokens), empty codes --
    against index_add, and bit-identical from run to run and to the token-order sum of each code."""
    from ctclip_hip import ops
    g = torch.Generator().manual_seed(9)
    idx = torch.randint(0, ncodes, (ntok,), generator=g)
    if skew:
        idx[100:2100] = 3
    idx = idx.to(DEV)
    x = rnd(ntok, dim, seed=140)
    inv = 1.0 / x.norm(dim=-1)
    outs = [ops.vq_ema_accum(x, inv, idx, ncodes, dim) for _ in range(3)]
    bins, esum, flat = outs[0]
    ref = torch.zeros(ncodes, dim, device=DEV).index_add_(0, idx, x * inv[:, None])
    check("embed_sum", esum, ref, 1e-5)
    assert torch.equal(bins, torch.bincount(idx, minlength=ncodes).float())
    assert flat.data_ptr() == bins.data_ptr()
    for b2, e2, _ in outs[1:]:
        assert torch.equal(e2, esum) and torch.equal(b2, bins)
    c = int(idx[0])                                                       # one code's sum, token by token in token order:
    seq = torch.zeros(dim, device=DEV)                                    # within one 256-row chunk the kernel adds exactly so
    xn = x * inv[:, None]
    if int((idx == c).sum()) <= 8:
        for r_ in xn[idx == c]:
            seq = seq + r_
        check("token-order sum of one code", esum[c], seq, 1e-6)
