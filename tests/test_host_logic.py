"""CPU tests (-m 'not gpu'): the C-ABI library loads and exports what include/ctclip_hip.h declares, host-side
mirrors of the reference interface (constructors, state-dict keys, optimiser factory, argument validation), and
that nothing silently falls back to CPU compute."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT, load_golden, sub


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from ctclip_hip.lib import hip, library_path, parse_header
    protos = parse_header()
    assert len(protos) >= 35
    dll = ctypes.CDLL(library_path())
    for name in protos:
        assert hasattr(dll, name), name
    hdr = open(os.path.join(ROOT, "include", "ctclip_hip.h")).read()
    assert set(re.findall(r"\bint\s+(ctclip_\w+)\s*\(", hdr)) == set(protos)
    # every entry point takes the stream last and returns int
    assert all(args[-1][1] == "stream" for args in protos.values())


def test_product_path_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "ct-clip-ut_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f


def test_cpu_tensors_raise_instead_of_falling_back():
    from utils.attention import Attention, FeedForward, PEG, Transformer, LayerNorm
    x = torch.randn(2, 4, 32)
    for m, kw in ((LayerNorm(32), {}), (FeedForward(32), {}), (Attention(32, dim_head=8, heads=4), {}),
                  (PEG(32, causal=True), {"shape": (1, 2, 2, 2)}),
                  (Transformer(32, depth=1, dim_head=8, heads=4, peg=True, peg_causal=True), {"video_shape": (1, 2, 2, 2)})):
        with pytest.raises(RuntimeError, match="no CPU fallback|HIP"):
            m(x, **kw)


def test_missing_library_is_loud(monkeypatch):
    from ctclip_hip import lib
    monkeypatch.setattr(lib, "library_path", lambda: "/nonexistent/libctclip_hip.so")
    h = lib._Hip()
    with pytest.raises(lib.HipLibraryMissing):
        h.symbols()


def test_reference_state_dicts_load_strict():
    from transformers import BertConfig, BertModel
    from models.ctclip import CTCLIP
    from utils.ctvit import CTViT
    g = load_golden("ctclip")
    vit = CTViT(dim=32, codebook_size=64, image_size=16, patch_size=4, temporal_patch_size=2, spatial_depth=1,
                temporal_depth=1, dim_head=8, heads=4)
    text = BertModel(BertConfig(hidden_size=32, num_hidden_layers=2, num_attention_heads=4, intermediate_size=64,
                                vocab_size=97, max_position_embeddings=40))
    clip = CTCLIP(text_encoder=text, image_encoder=vit, dim_text=32, dim_image=512, dim_latent=16)
    sd = sub(g, "sd.")
    clip.load_state_dict(sd, strict=True)
    assert set(clip.state_dict().keys()) == set(sd.keys())
    for k, v in clip.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), k
    # production shapes of SURVEY 3.4
    big = CTViT(dim=512, codebook_size=8192, image_size=480, patch_size=20, temporal_patch_size=10, spatial_depth=1,
                temporal_depth=1, dim_head=32, heads=8)
    s = big.state_dict()
    assert tuple(s["enc_spatial_transformer.layers.0.0.dsconv.weight"].shape) == (512, 1, 3, 3, 3)
    assert tuple(s["enc_spatial_transformer.layers.0.1.null_kv"].shape) == (8, 0, 32)
    assert tuple(s["enc_spatial_transformer.layers.0.3.1.weight"].shape) == (2730, 512)
    assert tuple(s["enc_spatial_transformer.layers.0.3.4.weight"].shape) == (512, 1365)
    assert tuple(s["vq._codebook.embed"].shape) == (1, 8192, 512)
    assert tuple(s["to_patch_emb.2.weight"].shape) == (512, 4000)


def test_ctclip_load_errors_match_reference(tmp_path):
    from models.ctclip import CTCLIP
    from utils.ctvit import CTViT
    clip = CTCLIP(text_encoder=torch.nn.Identity(), image_encoder=torch.nn.Identity(), dim_text=8, dim_image=8, dim_latent=8)
    with pytest.raises(FileNotFoundError):
        clip.load(tmp_path / "nope.pt")
    bad = tmp_path / "bad.pt"
    bad.write_bytes(b"not a checkpoint")
    with pytest.raises(RuntimeError):
        clip.load(bad)
    ok = tmp_path / "ok.pt"
    torch.save({"temperature": torch.tensor(0.5)}, ok)
    clip.load(ok, strict=False)
    assert float(clip.temperature) == 0.5


def test_get_optimizer_groups_like_reference():
    from utils.optimizer import get_optimizer, separate_params_by_weight_decay
    ps = [torch.nn.Parameter(torch.zeros(3, 3)), torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.zeros(2, 2, 2))]
    o = get_optimizer(ps, lr=1e-3, wd=0.0)
    assert len(o.param_groups) == 1 and o.param_groups[0]["betas"] == (0.9, 0.99)
    assert not o.param_groups[0]["decoupled_weight_decay"]
    o = get_optimizer(ps, lr=1e-3, wd=0.01)
    assert [len(g["params"]) for g in o.param_groups] == [2, 1]
    assert [g["weight_decay"] for g in o.param_groups] == [0.01, 0.0]
    assert all(g["decoupled_weight_decay"] for g in o.param_groups)
    wd, nwd = separate_params_by_weight_decay(ps)
    assert len(wd) == 2 and len(nwd) == 1
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        o.step()
    # the group keys are torch.optim.Adam's, so an exported state dict loads into the reference's optimiser
    ref = torch.optim.AdamW([{"params": wd}, {"params": nwd, "weight_decay": 0}], lr=1e-3, weight_decay=0.01)
    assert set(ref.param_groups[0]) == set(o.param_groups[0])


def test_optimizer_state_dict_interchanges_with_torch_adam():
    """reference CTClipTrainer.py:136-154 saves `optim.state_dict()` of torch.optim.Adam over model.parameters().
    HipAdam reads and writes that format: per-parameter step / exp_avg / exp_avg_sq keyed by the parameter's index in
    the FULL parameter list (statically unused parameters keep their index and carry no state, exactly like a torch
    Adam parameter whose grad stays None).  Arena plumbing only -- no HIP kernel runs here."""
    import copy
    from ctclip_hip.optim import HipAdam, mark_unused
    torch.manual_seed(0)
    shapes = [(4, 3), (5,), (2, 2, 2), (7,)]
    ref_ps = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    ref = torch.optim.Adam(ref_ps, lr=1e-2, betas=(0.9, 0.99), eps=1e-8)
    for _ in range(2):
        for i, p in enumerate(ref_ps):
            p.grad = None if i == 1 else torch.randn(p.shape)             # parameter 1 never gets a gradient
        ref.step()
    sd = ref.state_dict()
    assert set(sd["state"]) == {0, 2, 3}
    mine_ps = [torch.nn.Parameter(p.detach().clone()) for p in ref_ps]
    mark_unused(mine_ps[1])
    mine = HipAdam(mine_ps, lr=5e-1, betas=(0.5, 0.5), eps=1e-3)
    frozen = copy.deepcopy(sd)
    mine.load_state_dict(sd)
    assert set(sd) == set(frozen) and set(sd["state"]) == set(frozen["state"])      # the caller's dict is untouched
    g = mine.param_groups[0]
    assert (g["lr"], g["betas"], g["eps"]) == (1e-2, (0.9, 0.99), 1e-8) and mine._step == 2
    a = mine._arenas[0]
    assert [id(p) for p in a["params"]] == [id(mine_ps[i]) for i in (0, 2, 3)]        # no slot for the unused one
    for p, o in zip(a["params"], a["offs"]):
        i = [id(q) for q in mine_ps].index(id(p))
        torch.testing.assert_close(a["m"][o:o + p.numel()].view(p.shape), sd["state"][i]["exp_avg"])
        torch.testing.assert_close(a["v"][o:o + p.numel()].view(p.shape), sd["state"][i]["exp_avg_sq"])
    # and back: what HipAdam writes loads into a fresh torch Adam, which then continues from step 2
    out = mine.state_dict()
    assert set(out["state"]) == {0, 2, 3} and out["param_groups"][0]["params"] == [0, 1, 2, 3]
    ref2_ps = [torch.nn.Parameter(p.detach().clone()) for p in ref_ps]
    ref2 = torch.optim.Adam(ref2_ps, lr=1.0)
    ref2.load_state_dict(out)
    assert ref2.param_groups[0]["lr"] == 1e-2
    grads = [torch.randn(s) for s in shapes]
    for opt, ps in ((ref, ref_ps), (ref2, ref2_ps)):
        for p, gr in zip(ps, grads):
            p.grad = gr.clone()
        opt.step()
    for a_, b_ in zip(ref_ps, ref2_ps):
        torch.testing.assert_close(a_.detach(), b_.detach())
    # mismatching group sizes are rejected like torch does
    with pytest.raises(ValueError):
        HipAdam(mine_ps[:2], lr=1e-3).load_state_dict(sd)
    # a gradient assigned as a NEW tensor (the torch.optim contract) is adopted into the arena before the step
    mine.zero_grad()
    new = torch.randn(4, 3)
    mine_ps[0].grad = new.clone()
    mine._bind_grads(adopt=True)
    assert mine_ps[0].grad.data_ptr() == a["g"].data_ptr() and torch.equal(mine_ps[0].grad, new)


def test_unsupported_branches_fail_loudly():
    from utils.attention import Attention, Transformer
    from utils.ctvit import CTViT
    with pytest.raises(NotImplementedError):
        Attention(32, causal=True)
    with pytest.raises(NotImplementedError):
        Transformer(32, depth=1, has_cross_attn=True)
    v = CTViT(dim=32, codebook_size=16, image_size=8, patch_size=4, temporal_patch_size=2, spatial_depth=1,
              temporal_depth=1, dim_head=8, heads=4, model_type="ctgenerate")
    with pytest.raises(NotImplementedError):
        v(torch.zeros(1, 1, 2, 8, 8))


def test_position_table_indexing_matches_dense_grid():
    """relidx/unique-row construction of ContinuousPositionBias vs the oracle's dense (h*w)^2 grid (host logic)."""
    from utils.attention import ContinuousPositionBias
    from oracle import ctclip_oracle as O
    cpb = ContinuousPositionBias(dim=8, heads=2)
    for h, w in ((3, 4), (5, 2), (24, 24)):
        rows, relidx = cpb._tables((h, w), torch.device("cpu"))
        dense = O.cpb_relpos(h, w)
        assert rows.shape[0] == (2 * h - 1) * (2 * w - 1)
        assert torch.equal(rows[relidx.to(torch.int64)], dense)


def test_occlusion_window_slices_partition_the_scan():
    """reference src/utils/visualizations.py:352-362: ranks take contiguous, equal slices of the window list (extra
    windows dropped); the reduced rank maps equal the single-process maps (oracle restatement, tiny model)."""
    from transformers import BertConfig, BertModel
    from oracle import ctclip_oracle as O
    torch.manual_seed(0)
    cfg = dict(dim=32, codebook_size=64, image_size=32, patch_size=16, temporal_patch_size=16, spatial_depth=1,
               temporal_depth=1, dim_head=8, heads=2, text_layers=1, text_heads=2)
    from models.ctclip import CTCLIP
    from utils.ctvit import CTViT
    vit = CTViT(**{k: v for k, v in cfg.items() if not k.startswith("text_")})
    bert = BertModel(BertConfig(hidden_size=32, num_hidden_layers=1, num_attention_heads=2, intermediate_size=64,
                                vocab_size=50, max_position_embeddings=16, hidden_dropout_prob=0.0,
                                attention_probs_dropout_prob=0.0))
    clip = CTCLIP(text_encoder=bert, image_encoder=vit, dim_text=32, dim_image=2 * 2 * 32, dim_latent=16)
    st = {k: v.clone() for k, v in clip.state_dict().items()}
    gen = torch.Generator().manual_seed(3)
    image = (torch.randn(1, 1, 32, 32, 32, generator=gen) * 0.5).clamp(-1, 1)
    ids = torch.randint(0, 50, (1, 8), generator=gen)
    txt = {"input_ids": ids, "token_type_ids": torch.zeros_like(ids), "attention_mask": torch.ones_like(ids)}
    patch, stride = (16, 16, 16), (16, 16, 8)                       # 2 x 2 x 3 = 12 windows
    h1, c1, final = O.occlusion_heatmap(txt, image, st, cfg, patch, stride)
    parts = [O.occlusion_heatmap(txt, image, st, cfg, patch, stride, rank=r, world_size=2) for r in range(2)]
    assert all(p[2] is None for p in parts) and final.shape == (32, 32, 32)
    torch.testing.assert_close(parts[0][0] + parts[1][0], h1)
    torch.testing.assert_close(parts[0][1] + parts[1][1], c1)
    assert float(c1.max()) == 2.0 and float(c1.min()) == 1.0       # stride 8 along w: the middle band is covered twice
    assert 0.0 <= float(final.min()) and float(final.max()) <= 1.0
    # 5 ranks: 12 // 5 = 2 windows each, the last two windows are dropped (reference :356)
    p5 = [O.occlusion_heatmap(txt, image, st, cfg, patch, stride, rank=r, world_size=5) for r in range(5)]
    assert float(sum(p[1] for p in p5).sum()) == 10 * 16 ** 3


def test_visualizations_scope():
    from utils.visualizations import Visualizations

    class Acc:
        is_main_process, process_index, num_processes, device = True, 0, 1, torch.device("cpu")

    vis = Visualizations(torch.nn.Linear(2, 2), Acc())
    with pytest.raises(NotImplementedError):
        vis.visualize(visualizations=["grad_cam"])


def test_shadow_plan_holds_its_sets_weakly():
    """ops.PLAN (one copy launch refills every module's kernel-layout weight shadows after an optimiser step) must not keep dead
    models alive: a ShadowSet pins its module's parameters and shadow tensors, so the plan holds the sets weakly -- dropping the
    module's ShadowCache removes its sets from the plan and invalidates the cached all-sets table."""
    import gc
    from ctclip_hip import ops

    class Mod:
        def __init__(self):
            self.w = torch.nn.Parameter(torch.randn(8, 16))
            self.cache = ops.ShadowCache()

        def shadows(self, refresh):
            def make():
                S = ops.ShadowSet(self.w.device)
                S.out["w16"] = S.zeros(8, 16)
                S.add(self.w, S.out["w16"])
                return S
            real = ops.PLAN.refresh
            ops.PLAN.refresh = refresh                                 # no GPU here: the copy launch is stood in
            try:
                return self.cache.get_set("w", (self.w,), make)
            finally:
                ops.PLAN.refresh = real

    launched = []
    before = len(ops.PLAN.sets)
    a, b = Mod(), Mod()
    a.shadows(lambda s_, e: launched.append(s_))
    b.shadows(lambda s_, e: launched.append(s_))
    assert len(ops.PLAN.sets) == before + 2 and len(launched) == 2
    ops.PLAN._all = {"stale": None}
    set_b = b.cache._store["w"][1]
    del a, launched
    gc.collect()
    assert len(ops.PLAN.sets) == before + 1 and ops.PLAN.sets[-1] is set_b       # a's set left with its module
    assert ops.PLAN._all is None                                                  # ... and the all-sets table is rebuilt
    ops.PLAN.unregister(set_b)
    assert len(ops.PLAN.sets) == before


def test_no_read_of_registers_an_asm_load_has_in_flight(tmp_path):
    """csrc/patch_gemm.hip keeps registers loaded by inline-asm global loads in flight across K-steps (the volume pieces of the
    fused tubelet embedding and of its weight gradient).  The compiler believes an asm output is valid at once, so a register copy
    it inserts between the load and the tied `s_waitcnt` would read garbage -- silently.  Compile the file for gfx950 and scan both
    kernels' ISA (tools/check_inflight.py): no instruction may read such a register before the wait that covers the load."""
    import shutil
    import subprocess
    import sys
    if shutil.which("hipcc") is None:
        pytest.skip("no hipcc")
    # the same holds for the inline-asm transposed LDS reads (ds_read_b64_tr_b16) of the weight-gradient kernels: on no path to the
    # tied s_waitcnt lgkmcnt(0) may their destination be read or re-used (round 5: a one-wave-per-SIMD variant faulted on a dead one)
    for name, kernels in (("patch_gemm", ("patch_gemm_fwd_kernel", "patch_wgrad_kernel")), ("gemm4", ("gemm4_kernel",))):
        src = os.path.join(ROOT, "ct-clip-ut_amd", "csrc", name + ".hip")
        out = str(tmp_path / (name + ".s"))
        r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-munsafe-fp-atomics", "-std=c++17", "-Wno-unused-value",
                            "--cuda-device-only", "-S", "-o", out, src], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        for kernel in kernels:
            c = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_inflight.py"), out, kernel], capture_output=True, text=True)
            assert c.returncode == 0, c.stdout[-2000:]
