import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "ct-clip-ut_amd"))
import torch
from ctclip_hip.lib import hip
torch.manual_seed(0)
B = 64; nseq, n, H, D, gh, gw = 24 * B, 576, 8, 32, 24, 24
ld = H * D; dev = "cuda"
q, k, v, do = (torch.nn.functional.normalize(torch.randn(nseq * n, H, D, device=dev), dim=-1).reshape(nseq * n, ld).to(torch.bfloat16) for _ in range(4))
q = (q.float() * 8).to(torch.bfloat16)
bias = torch.randn(H, n, n, device=dev)
o = torch.empty_like(q); lse = torch.empty(nseq, H, n, device=dev)
hip.attn_fwd(q, k, v, o, lse, bias, None, nseq, n, H, D, ld, ld, ld, ld, 1.0)
dq, dk, dv = (torch.empty_like(q) for _ in range(3)); delta = torch.empty_like(lse)
R = (2 * gh - 1) * (2 * gw - 1)
outs = []
for it in range(8):
    dt = torch.zeros(H, R, device=dev)
    hip.attn_bwd(q, k, v, o, do, lse, delta, dq, dk, dv, bias, None, None, None, dt, R, gh, gw, nseq, n, H, D, ld, ld, ld, ld, ld, ld, ld, ld, 1.0)
    torch.cuda.synchronize()
    outs.append((dt.clone(), dq.clone()))
ref_t, ref_q = outs[0]
worst = max(float((t - ref_t).abs().max() / ref_t.abs().max()) for t, _ in outs[1:])
same_dq = all(torch.equal(x, ref_q) for _, x in outs[1:])
# dense path on a subset of sequences as an independent check of the table scatter
ns = 48
dd = torch.zeros(H, n, n, device=dev)
hip.attn_bwd(q[:ns*n], k[:ns*n], v[:ns*n], o[:ns*n], do[:ns*n], lse[:ns], delta[:ns].clone(), dq[:ns*n].clone(), dk[:ns*n].clone(), dv[:ns*n].clone(), bias, None, dd, None, None, 0, 0, 0, ns, n, H, D, ld, ld, ld, ld, ld, ld, ld, ld, 1.0)
dt2 = torch.zeros(H, R, device=dev)
hip.attn_bwd(q[:ns*n], k[:ns*n], v[:ns*n], o[:ns*n], do[:ns*n], lse[:ns], delta[:ns].clone(), dq[:ns*n].clone(), dk[:ns*n].clone(), dv[:ns*n].clone(), bias, None, None, None, dt2, R, gh, gw, ns, n, H, D, ld, ld, ld, ld, ld, ld, ld, ld, 1.0)
ii = torch.arange(n, device=dev)
rel = ((ii[:, None] // gw - ii[None] // gw + gh - 1) * (2 * gw - 1) + (ii[:, None] % gw - ii[None] % gw + gw - 1))
ref2 = torch.zeros(H, R, device=dev).index_add_(1, rel.reshape(-1), dd.reshape(H, -1))
print(f"run-to-run d(table) deviation over 8 runs (relative to peak): {worst:.2e}; dq bit-identical: {same_dq}")
print(f"table path vs dense path scattered (48 sequences): {float((dt2 - ref2).abs().max() / ref2.abs().max()):.2e}")
