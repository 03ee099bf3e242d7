"""Optimiser factory with the reference's signature (src/utils/optimizer.py:14-54).

wd == 0 -> Adam; otherwise AdamW with tensors of ndim >= 2 decayed and ndim < 2 not (when group_wd_params).
Returns a torch.optim.Optimizer subclass whose step is a fused HIP kernel over flat arenas."""
from ctclip_hip.optim import HipAdam


def separate_params_by_weight_decay(params):
    params = list(params)
    return [p for p in params if p.ndim >= 2], [p for p in params if p.ndim < 2]


def get_optimizer(params, lr=1e-4, wd=1e-4, betas=(0.9, 0.99), eps=1e-8, filter_requires_grad=False,
                  group_wd_params=True, **kwargs):
    params = list(params)
    if filter_requires_grad:
        params = [p for p in params if p.requires_grad]
    if wd == 0:
        return HipAdam(params, lr=lr, betas=betas, eps=eps, weight_decay=0.0, decoupled=False)
    if group_wd_params:
        wd_params, no_wd_params = separate_params_by_weight_decay(params)
        params = [{"params": wd_params}, {"params": no_wd_params, "weight_decay": 0.0}]
    return HipAdam(params, lr=lr, betas=betas, eps=eps, weight_decay=wd, decoupled=True)
