"""AdamW + gradient clipping of the training step as two HIP launches (csrc/optim.hip).

The reference trains with `torch.optim.AdamW(lr=1e-3, weight_decay=0.1)` behind Lightning's `gradient_clip_val: 5.0` (config yaml:117-120,
train.py:135-146).  Over RTFS-Net's 403 parameter tensors that pair is ~15 multi-tensor launches and a host-bound 2 ms at the end of every step;
`FusedAdamW.step(max_norm=5.0)` does the same arithmetic - total gradient norm, clip coefficient, in-place gradient scaling, decoupled weight decay,
moment updates, bias-corrected update, in torch.optim.AdamW's own operation order - with the clip coefficient formed on the device.

`FusedAdamW` is a `torch.optim.Optimizer`: param groups, `zero_grad`, `state_dict` / `load_state_dict` behave as usual and the per-parameter state
(`step`, `exp_avg`, `exp_avg_sq`) has torch.optim.AdamW's keys, so optimizer checkpoints interchange with it.  One group's hyper-parameters apply per
launch pair; parameters must be contiguous fp32 CUDA tensors (RTFS-Net's are).  `amsgrad`, `maximize`, sparse gradients: not supported (raise).
"""
from __future__ import annotations

import math

import torch

from . import lib

_CHUNK = 1024  # kOptChunk in csrc/optim.hip


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        if lr < 0.0 or eps < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or weight_decay < 0.0:
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._plans = {}

    # ---- per-group launch plan: static tables (parameters, state, sizes, chunk map) + a pinned row for the gradient pointers of the step ----
    def _plan(self, gi, ps):
        # the tables hold raw device addresses: a plan is valid for exactly these parameter objects AT these addresses (p.data = ..., module.to() and
        # re-sharding keep the Parameter object and move its storage) with these moment tensors
        key = tuple((id(p), p.data_ptr(), p.numel()) for p in ps)
        plan = self._plans.get(gi)
        if plan is not None and plan["key"] == key and all(self.state[p]["exp_avg"].data_ptr() == a and self.state[p]["exp_avg_sq"].data_ptr() == b
                                                           for p, a, b in zip(ps, plan["m_ptrs"], plan["v_ptrs"])):
            return plan
        dev = ps[0].device
        for p in ps:
            if p.dtype != torch.float32 or not p.is_cuda or p.device != dev or not p.is_contiguous():
                raise ValueError("FusedAdamW needs contiguous fp32 parameters on one HIP device")
            st = self.state[p]
            if len(st) == 0:
                st["step"] = torch.zeros((), dtype=torch.float32)  # host scalar per parameter, as torch.optim.AdamW's non-capturable path keeps it (a
                # tensor shared by all parameters would be incremented once per parameter by torch.optim.AdamW after a state_dict interchange)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        sizes = [p.numel() for p in ps]
        chunks = [(i, s) for i, n in enumerate(sizes) for s in range(0, n, _CHUNK)]
        host = torch.zeros(5, len(ps), dtype=torch.int64)
        host[0] = torch.tensor([p.data_ptr() for p in ps], dtype=torch.int64)
        host[2] = torch.tensor([self.state[p]["exp_avg"].data_ptr() for p in ps], dtype=torch.int64)
        host[3] = torch.tensor([self.state[p]["exp_avg_sq"].data_ptr() for p in ps], dtype=torch.int64)
        host[4] = torch.tensor(sizes, dtype=torch.int64)
        plan = {"key": key, "dev": host.to(dev), "n_chunks": len(chunks), "n": len(ps),
                "chunks": torch.tensor(chunks, dtype=torch.int32).reshape(-1, 2).to(dev), "sqnorm": torch.zeros(1, dtype=torch.float64, device=dev),
                "m_ptrs": [self.state[p]["exp_avg"].data_ptr() for p in ps], "v_ptrs": [self.state[p]["exp_avg_sq"].data_ptr() for p in ps]}
        self._plans[gi] = plan
        return plan

    @torch.no_grad()
    def step(self, closure=None, max_norm=None):
        """One AdamW step.  `max_norm`: clip the total 2-norm of ALL gradients handed to this optimizer to it first (torch.nn.utils.clip_grad_norm_'s
        arithmetic, the gradients are left scaled in place); None: no clipping.  With several param groups the norm is taken over all of them."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        work = []
        for gi, group in enumerate(self.param_groups):
            if group.get("amsgrad") or group.get("maximize"):
                raise NotImplementedError("FusedAdamW: amsgrad / maximize are not implemented")
            with_grad = [p for p in group["params"] if p.grad is not None]
            if not with_grad:
                continue
            # one launch applies ONE pair of bias corrections: parameters whose step counts differ (a parameter that first received a gradient later, one
            # skipped under DDP's find_unused_parameters, interchanged optimizer states) go in separate launches, as torch.optim.AdamW corrects per parameter
            by_step = {}
            for p in with_grad:
                st = self.state[p]
                by_step.setdefault(float(st["step"]) if len(st) else 0.0, []).append(p)
            for si, (_, ps) in enumerate(sorted(by_step.items())):
                work.append(self._stage(group, (gi, si, len(by_step)), ps))
        if not work:
            return loss
        clip = max_norm is not None
        if clip:
            if len(work) == 1:
                group, ps, plan = work[0]
                d = plan["dev"]
                lib.call("rtfs_grad_sqnorm", d[0], d[1], d[2], d[3], d[4], plan["chunks"], plan["n_chunks"], plan["sqnorm"])
                sq = plan["sqnorm"]
            else:  # several launches: one norm over all of them (the kernel zeroes its output, so the partial sums are added here on the device)
                sq = None
                for group, ps, plan in work:
                    d = plan["dev"]
                    lib.call("rtfs_grad_sqnorm", d[0], d[1], d[2], d[3], d[4], plan["chunks"], plan["n_chunks"], plan["sqnorm"])
                    sq = plan["sqnorm"].clone() if sq is None else sq + plan["sqnorm"]
        for group, ps, plan in work:
            torch._foreach_add_([self.state[p]["step"] for p in ps], 1)
            k = float(self.state[ps[0]]["step"])
            b1, b2 = group["betas"]
            d = plan["dev"]
            lib.call("rtfs_adamw_clip_step", d[0], d[1], d[2], d[3], d[4], plan["chunks"], plan["n_chunks"], sq if clip else plan["sqnorm"],
                     float(max_norm) if clip else 0.0, float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                     1.0 - b1 ** k, math.sqrt(1.0 - b2 ** k))
            # the kernel wrote the parameters, moments and gradients through raw pointers: tell autograd (and everything keyed on version counters -
            # the kernel-layout weight copies of rtfs_net_amd.models are rebuilt when a parameter's counter moves) that these tensors changed
            torch.autograd.graph.increment_version(ps)
            torch.autograd.graph.increment_version([p.grad for p in ps])
            torch.autograd.graph.increment_version([self.state[p]["exp_avg"] for p in ps] + [self.state[p]["exp_avg_sq"] for p in ps])
        return loss

    def _stage(self, group, slot, ps):
        """plan + this step's gradient-pointer row of one launch (the parameters of `group` that share a step count)"""
        plan = self._plan(slot, ps)
        # the gradient tensors are new objects every step: their pointers go up in a FRESH pinned row (the caching host allocator keeps a block until
        # the copy that reads it has run - the host may be several steps ahead of the device, a reused staging row would be overwritten under it)
        row = torch.empty(plan["n"], dtype=torch.int64, pin_memory=True)
        g_np = row.numpy()
        for i, p in enumerate(ps):
            g = p.grad
            if g.is_sparse or g.dtype != torch.float32 or not g.is_contiguous() or g.device != p.device:
                raise ValueError("FusedAdamW needs dense contiguous fp32 gradients on the parameters' device")
            g_np[i] = g.data_ptr()
        plan["dev"][1].copy_(row, non_blocking=True)
        return group, ps, plan
