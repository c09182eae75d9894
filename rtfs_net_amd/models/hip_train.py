"""Training step of the HIP separation path: forward with saved activations + hand-written backward chain.

The reference trains through torch autograd over its forward modules (train.py:148, src/system/core.py:94-117) and
sru's CUDA backward kernel.  Here the audio branch is ONE `torch.autograd.Function` (`AVNetHipFunction`) whose forward
and backward are sequences of include/rtfs_hip.h entry points; PyTorch autograd only sees its inputs (waveform,
the CAF video-side tables `att`/`rsz`, the audio-branch parameters) and output, and differentiates the small
video-side glue (VP block, CAF video projections) itself.

Activations are saved, not recomputed (B=32, R=6 needs ~40 GB of the 288 GB).  Parameter gradients are accumulated in
kernel layout and mapped back to the reference parameter shapes at the end (`_grads_to_reference`).
"""
from __future__ import annotations

import os

import torch

try:
    from .. import lib
except ImportError:  # a relocated copy of this sub-package (train.py:95 copies src/models into the experiment directory and test.py:33-36
    from rtfs_net_amd import lib  # imports it as <exp>.models): the binding is taken from the installed package
from .hip_path import C, F2, F_BINS, H, PACKED_WEIGHT_MODES, PreparedWeights, _f32, pack_bf16


def _t(x):
    return x.t().contiguous()


class TrainWeights(PreparedWeights):
    """PreparedWeights + the transposed / re-ordered copies the input-gradient GEMMs need."""

    def __init__(self, model, prec=0):
        super().__init__(model, pack_vp=False, training=model.training)  # rebuilt after every optimizer step: the VP kernel (eval only) is not needed here
        self.prec = prec
        w = self.w
        w["bn_wT"], w["mask_wT"], w["dec_wT"] = _t(w["bn_w"]), _t(w["mask_w"]), _t(w["dec_w"])
        sd = {k: v.detach() for k, v in model.state_dict().items()}
        p0 = "refinement_module.audio_net.blocks."
        shared = model.refinement_module.audio_net.shared
        for i, b in enumerate(self.blocks):
            p = p0 if shared else f"{p0}{i}."
            b["rwT"], b["pwT"] = _t(b["rw"]), _t(b["pw"])
            for j in (0, 1):
                d = b[f"dp{j}"]
                d["fold_w"] = _f32(d["w0"].view(256, 8, 64).flip(1).permute(2, 1, 0).reshape(64, 2048))
                d["ctbi_w"] = _f32(sd[f"{p}globalatt.{j}.linear.weight"].float().permute(0, 2, 1).reshape(64, 512))
                for lw in d["layers"][1:]:
                    lw["wT"] = _t(lw["w"])
            a = b["attn"]
            a["wT"], a["owT"] = _t(a["w"]), _t(a["ow"])
        # raw CAF depth-wise weights and BatchNorm parameters (training mode uses batch statistics)
        caf = "refinement_module.crossmodal_fusion.fusion_module." + ("" if model.refinement_module.crossmodal_fusion.fusion_shared else "0.") + "audio_lstm."
        self.caf_prefix = caf
        for tag in ("key", "value"):
            q = f"{caf}{tag}_embed.full_layer."
            w[f"caf_{tag}_dw"] = _f32(sd[q + "2.weight"].reshape(C))
            w[f"caf_{tag}_g"], w[f"caf_{tag}_be"] = _f32(sd[q + "3.weight"]), _f32(sd[q + "3.bias"])
        if prec in PACKED_WEIGHT_MODES:  # bf16 / split-bf16 step: every weight that is ONLY an MFMA operand is replaced by its host-packed form (hip_path.pack_bf16);
            # the SRU layer 1-3 weights `w` stay fp32 (packed inside rtfs_sru_layer_fwd_bf16 after the gate scaling)
            for k in ("bn_w", "mask_w", "dec_w", "bn_wT", "mask_wT", "dec_wT"):
                w[k] = pack_bf16(w[k])
            for b in self.blocks:
                for k in ("pw", "rw", "pwT", "rwT"):
                    b[k] = pack_bf16(b[k])
                for j in (0, 1):
                    d = b[f"dp{j}"]
                    for k in ("w0", "ct_w", "fold_w", "ctbi_w"):
                        d[k] = pack_bf16(d[k])
                    for lw in d["layers"][1:]:
                        lw["wT"] = pack_bf16(lw["wT"])
                a = b["attn"]
                for k in ("w", "ow", "wT", "owT"):
                    a[k] = pack_bf16(a[k])


class _IndexModel:
    """stand-in for the model while a gather plan is built: everything is the model's, except that state_dict() holds, for every floating-point
    tensor, the POSITIONS of its elements in one flat buffer (as float32 numbers, exact below 2^24)"""

    def __init__(self, model, sd_idx):
        object.__setattr__(self, "_m", model)
        object.__setattr__(self, "_sd", sd_idx)

    def state_dict(self):
        return self._sd

    def __getattr__(self, k):
        return getattr(self._m, k)


class GatherTrainWeights:
    """TrainWeights whose re-layout is ONE gather.  Every kernel-layout copy TrainWeights makes in the fp32 modes is a permutation / slice / concatenation of
    parameter elements (plus zero padding), but as ~150 separate torch launches issued right after the step's one device-to-host transfer, where the host cannot
    be ahead of the device: ~3 ms per training step, most of it idle device (round 5, tools/train_gaps.py).  Here the construction code of TrainWeights runs
    ONCE on index-valued stand-ins (_IndexModel): what comes out is, for every element of every derived tensor, the position of its source element in a flat
    buffer of all floating-point state tensors.  Per step: one multi-tensor copy of the state into the flat buffer, one index_select into a persistent buffer
    that the (persistent) nested dicts are views of, and the scalar slots (PReLU slopes, SRU scale_x: kernel arguments passed by value) from one transfer."""

    def __init__(self, model, prec):
        dev = next(model.parameters()).device
        sd = model.state_dict()
        names = [k for k, v in sd.items() if v.is_floating_point()]
        self.srcs = [sd[k].detach() for k in names]
        sizes = [t.numel() for t in self.srcs]
        total = 1 + sum(sizes)  # element 0 stays zero: the source of every padding element
        if total >= 1 << 24:
            raise ValueError("GatherTrainWeights: more than 2^24 state elements (float32 index arithmetic)")
        self.flat = torch.zeros(total, device=dev)
        sd_idx, self.flat_views, off = dict(sd), [], 1
        for k, t, n in zip(names, self.srcs, sizes):
            self.flat_views.append(self.flat[off:off + n].view(t.shape))
            sd_idx[k] = (torch.arange(n, device=dev, dtype=torch.float32) + off).view(t.shape)
            off += n
        tw = TrainWeights(_IndexModel(model, sd_idx), 0)  # (the fp32 structure; the bf16 / split-bf16 modes pack a contiguous region of it below)
        self.caf_prefix, self.prec = tw.caf_prefix, prec
        self.w, self.blocks, self._scal = tw.w, tw.blocks, tw._scal
        for tag in ("key", "value"):  # BatchNorm folded with RUNNING statistics (eval under autograd): arithmetic, not a gather - see refresh()
            self.w.pop(f"caf_{tag}_s", None), self.w.pop(f"caf_{tag}_b", None)
        leaves, self.scalars = [], []

        def walk(c):
            items = c.items() if isinstance(c, dict) else enumerate(c)
            for k, v in list(items):
                if isinstance(v, tuple):
                    v = c[k] = list(v)
                if isinstance(v, torch.Tensor):
                    leaves.append((c, k, v))
                elif isinstance(v, float):
                    self.scalars.append((c, k, int(round(v))))
                elif isinstance(v, (dict, list)):
                    walk(v)

        walk(self.w), walk(self.blocks), walk(self._scal)
        # bf16 / split-bf16 step: the weights that are ONLY MFMA operands are handed over host-packed (hip_path.pack_bf16: every 4 consecutive k -> 4 hi + 4 lo
        # bfloat16).  Their fp32 layouts are gathered into ONE contiguous region first, so the packing of all of them is one pass over that region.
        packed_ids = set()
        if prec in PACKED_WEIGHT_MODES:
            def mark(c, keys):
                for k in keys:
                    packed_ids.add((id(c), k))
            mark(self.w, ("bn_w", "mask_w", "dec_w", "bn_wT", "mask_wT", "dec_wT"))
            for b in self.blocks:
                mark(b, ("pw", "rw", "pwT", "rwT"))
                for j in (0, 1):
                    mark(b[f"dp{j}"], ("w0", "ct_w", "fold_w", "ctbi_w"))
                    for lw in b[f"dp{j}"]["layers"][1:]:
                        mark(lw, ("wT",))
                mark(b["attn"], ("w", "ow", "wT", "owT"))
            leaves.sort(key=lambda l: (id(l[0]), l[1]) not in packed_ids)  # (stable: packed leaves first)
        self.idx = torch.cat([t.reshape(-1) for _, _, t in leaves]).round().long()
        assert int(self.idx.min()) >= 0 and int(self.idx.max()) < total
        self.derived = torch.empty(self.idx.numel(), device=dev)
        self.n_packed = sum(t.numel() for c, k, t in leaves if (id(c), k) in packed_ids)
        assert self.n_packed % 4 == 0
        self.packed = torch.empty(self.n_packed // 4, 8, dtype=torch.bfloat16, device=dev) if self.n_packed else None
        o = 0
        for c, k, t in leaves:
            if (id(c), k) in packed_ids:
                assert t.ndim == 2 and t.shape[1] % 4 == 0 and o % 4 == 0
                c[k] = self.packed[o // 4:(o + t.numel()) // 4].view(t.shape[0], t.shape[1] // 4, 8)
            else:
                c[k] = self.derived[o:o + t.numel()].view(t.shape)
            o += t.numel()
        self.sidx = torch.tensor([i for _, _, i in self.scalars], device=dev, dtype=torch.long)
        self.ptrs = self._storage(model)
        # eval under autograd folds the CAF embeddings' BatchNorm with its running statistics: the six state tensors per embedding, looked up once (a
        # re-allocated one fails same_storage() and the plan is rebuilt)
        self._caf_bn = {tag: [sd[f"{self.caf_prefix}{tag}_embed.full_layer.{k}"] for k in ("2.weight", "3.weight", "3.bias", "3.running_mean", "3.running_var")]
                        for tag in ("key", "value")}
        self.version, self._pending = None, None
        self.refresh(model)

    @staticmethod
    def _storage(model):
        """addresses of every floating-point parameter and buffer, by the module-tree walk of PreparedWeights.fingerprint (model.state_dict() builds 365
        dotted names and an OrderedDict per call: most of the host time the fingerprint walk saved)"""
        out, stack = [], [model]
        while stack:
            mod = stack.pop()
            out.extend(p.data_ptr() for p in mod._parameters.values() if p is not None and p.is_floating_point())
            out.extend(b.data_ptr() for b in mod._buffers.values() if b is not None and b.is_floating_point())
            stack.extend(m for m in mod._modules.values() if m is not None)
        return tuple(out)

    def same_storage(self, model):
        """the plan reads the state through the tensors it was built from: a model moved / re-allocated since needs a new plan"""
        return self._storage(model) == self.ptrs

    @torch.no_grad()
    def refresh(self, model, fp=None, defer_scalars=False):
        """defer_scalars: the scalar slots keep their OLD values until finish_scalars() - the caller launches what needs no scalar first (STFT, encoder,
        bottleneck: 1.3 ms of device work) and waits for the transfer behind those launches instead of in front of them"""
        torch._foreach_copy_(self.flat_views, self.srcs)
        torch.index_select(self.flat, 0, self.idx, out=self.derived)
        if self.packed is not None:  # pack_bf16 over the whole packed region at once
            r = self.derived[:self.n_packed]
            hi = r.bfloat16()
            lo = (r - hi.float()).bfloat16()
            torch.cat([hi.view(-1, 4), lo.view(-1, 4)], -1, out=self.packed)
        host = torch.empty(self.sidx.numel(), dtype=torch.float32, pin_memory=True)
        host.copy_(self.flat[self.sidx], non_blocking=True)  # the step's one device-to-host transfer
        self._pending = (host, torch.cuda.current_stream(self.flat.device).record_event())
        if not defer_scalars:
            self.finish_scalars()
        if not model.training:  # eval under autograd: the CAF embeddings' BatchNorm folded with its running statistics (PreparedWeights.__init__)
            for tag in ("key", "value"):
                dw, gamma, beta, mean, var = self._caf_bn[tag]
                scale = gamma.float() / torch.sqrt(var.float() + 1e-5)
                self.w[f"caf_{tag}_s"] = _f32(dw.reshape(C).float() * scale)
                self.w[f"caf_{tag}_b"] = _f32(beta.float() - mean.float() * scale)
        self.version = fp if fp is not None else PreparedWeights.fingerprint(model, training=model.training)

    def finish_scalars(self):
        if self._pending is not None:
            host, ev = self._pending
            ev.synchronize()
            for (c, k, _), v in zip(self.scalars, host.tolist()):
                c[k] = v
            self._pending = None


class Ctx:
    """bag of saved tensors"""


def _zeros(n, dev, dtype=torch.float32):
    return torch.zeros(n, device=dev, dtype=dtype)


class _ZeroPool:
    """Zero-initialised scratch of one training step, carved from ONE pre-zeroed slab per dtype: the ~260 parameter-gradient accumulators
    and gLN adjoint reduction slots of a step otherwise cost one fill launch each (1578 FillFunctor launches per 6 steps in
    profiles/r02_c_train_kernel_stats.txt).  Slices are 128-byte aligned; a slab that runs out is replaced by a fresh one."""

    def __init__(self, dev, n_f32=2_200_000, n_f64=65_536):
        self.dev, self.size = dev, {torch.float32: n_f32, torch.float64: n_f64}
        self.slab, self.used = {}, {}

    def take(self, n, dtype=torch.float32):
        per = 128 // (4 if dtype == torch.float32 else 8)
        n_al = (n + per - 1) // per * per
        if dtype not in self.slab or self.used[dtype] + n_al > self.slab[dtype].numel():
            self.slab[dtype] = torch.zeros(max(self.size[dtype], n_al), dtype=dtype, device=self.dev)
            self.used[dtype] = 0
        o = self.used[dtype]
        self.used[dtype] = o + n_al
        return self.slab[dtype][o:o + n]


def _acc(gr, key, n, dev):
    """gradient accumulator `key` of the step, zero on FIRST use (a slice of the step's pre-zeroed pool, see _ZeroPool)"""
    t = gr.get(key)
    if t is None:
        pool = gr.get("_pool")
        if pool is None:
            pool = gr["_pool"] = _ZeroPool(dev)
        t = gr[key] = pool.take(n)
    return t


# entry points whose contraction runs on MFMA and that have a *_bf16 sibling (include/rtfs_hip.h); everything else is fp32 in every mode
_MFMA_ENTRY_POINTS = frozenset((
    "rtfs_bottleneck_fwd", "rtfs_proj_fwd", "rtfs_dp_unfold_gemm_fwd", "rtfs_sru_layer_fwd", "rtfs_dp_convt_fwd", "rtfs_dp_convt_fwd_to", "rtfs_attn_qkv_fwd", "rtfs_attn_core_fwd",
    "rtfs_attn_out_fwd", "rtfs_attn_out_fwd_to", "rtfs_resid_fwd", "rtfs_resid_proj_fwd", "rtfs_mask_fwd", "rtfs_gemm_rows", "rtfs_wgrad", "rtfs_proj_gateway_bwd", "rtfs_decoder_mask_bwd", "rtfs_fold_gemm_bwd",
    "rtfs_convt_bwd_input"))


class _Stage:
    """one backward stage of HipTrainer: begin_stage on entry of the outermost `with`, end_stage (flush + side-stream join) on its exit"""

    def __init__(self, trainer, dev):
        self.t, self.dev = trainer, dev

    def __enter__(self):
        depth = self.t.__dict__.setdefault("_stage_depth", {})
        depth[self.dev] = depth.get(self.dev, 0) + 1
        if depth[self.dev] == 1:
            self.t.begin_stage(self.dev)
        return self

    def __exit__(self, *exc):
        depth = self.t._stage_depth
        depth[self.dev] -= 1
        if depth[self.dev] == 0:
            self.t.end_stage(self.dev)
        return False


class HipTrainer:
    def __init__(self, model):
        self.model = model
        self._prep = None
        self.prec = 0  # AVNet.set_compute_dtype: 0 fp32, 1 bf16, 3 split-bf16 products in every MFMA kernel of the step
        self.attn_terms = 0  # "bf16-attn": the attention core's forward alone on the bf16 pipe (prec == 0)

    def weights(self, defer_scalars=False):
        """defer_scalars (forward_a only): the caller calls `.finish_scalars()` on the result before it reads a scalar slot (PReLU slopes, scale_x, `_scal`)"""
        fp = PreparedWeights.fingerprint(self.model, training=self.model.training)  # eval under autograd: running statistics are inputs too
        if self._prep is not None and self._prep.version == fp and self._prep.prec == self.prec:
            if isinstance(self._prep, GatherTrainWeights) and not defer_scalars:
                self._prep.finish_scalars()
            return self._prep
        if not self.model._hip.fuse.get("wgather", True):
            self._prep = TrainWeights(self.model, self.prec)
        elif isinstance(self._prep, GatherTrainWeights) and self._prep.prec == self.prec and self._prep.same_storage(self.model):
            self._prep.refresh(self.model, fp, defer_scalars)  # same tensors, new values (an optimizer step): one copy + one gather
        else:
            self._prep = GatherTrainWeights(self.model, self.prec)
        return self._prep

    def _call(self, name, *args):
        """lib.call, routed to the *_bf16 sibling (extra `terms` argument) for the MFMA entry points when a bf16 mode is selected"""
        if self.prec and name in _MFMA_ENTRY_POINTS:
            lib.call(name + "_bf16", *args, self.prec)
        elif self.attn_terms and name == "rtfs_attn_core_fwd":
            lib.call(name + "_bf16", *args, self.attn_terms)
        else:
            lib.call(name, *args)

    def invalidate(self):
        self._prep = None

    # ---- weight-gradient side stream ----------------------------------------------------------------------------------------------
    # The weight / tap gradient launches (rtfs_wgrad: MFMA-bound; rtfs_dwconv_bwd_weight) produce nothing the adjoint chain reads: their
    # results are needed when the stage ends.  They are issued on a second stream - scratch lane 1 of csrc/spread.hip - so that they run
    # UNDERNEATH the bandwidth-bound elementwise / scan adjoints of the chain instead of in line with them.  Ordering: the side stream waits for
    # the main stream at every launch (its operands were just produced there); operands are marked with record_stream (the caching allocator must
    # not hand their memory to a later main-stream tensor while the side launch is pending); the stage end joins (end_stage).
    def _side(self, dev):
        if not self.model._hip.fuse["wgside"]:
            return None
        st = self.__dict__.setdefault("_side_streams", {})
        if dev not in st:
            st[dev] = torch.cuda.Stream(device=dev)
        return st[dev]

    def _wg(self, name, *args):
        """a weight-gradient launch: on the side stream when enabled, else in line"""
        dev = next(a.device for a in args if isinstance(a, torch.Tensor))
        side = self._side(dev)
        if side is None:
            return self._call(name, *args)
        side.wait_stream(torch.cuda.current_stream(dev))
        for a in args:
            if isinstance(a, torch.Tensor):
                a.record_stream(side)
        lib.spread_lane(1)
        try:
            with torch.cuda.stream(side):
                self._call(name, *args)
        finally:
            lib.spread_lane(0)

    # The MFMA-bound weight gradients of a dual-path stage (two Toeplitz launches, ~360 us each alone) become ready in the middle of the chain's own
    # MFMA-bound stretch (SRU layer adjoints, rtfs_fold_gemm_bwd): launched there, they and the chain's GEMMs slow each other down by what the overlap
    # was meant to save.  _wg_later parks such a launch (its operands stay referenced); _wg_flush issues the parked ones where the chain turns
    # bandwidth-bound (the depth-wise / gateway adjoints behind the dual paths) - a side-stream launch waits for the chain's position AT ISSUE.
    def _wg_later(self, name, *args):
        if self.model._hip.fuse["wgdefer"] and self.model._hip.fuse["wgside"]:
            self.__dict__.setdefault("_parked", []).append((name, args))
        else:
            self._wg(name, *args)

    def _wg_flush(self):
        parked, self._parked = self.__dict__.get("_parked", []), []
        for name, args in parked:
            self._wg(name, *args)

    def begin_stage(self, dev):
        """deferred finish of the parameter-gradient reducers (csrc/spread.hip) on both lanes for one backward stage"""
        self._parked = []  # (a stage that raised may have left launches parked: they belong to a dead step)
        lib.spread_defer(True, dev)
        side = self._side(dev)
        if side is not None:
            lib.spread_lane(1)
            try:
                with torch.cuda.stream(side):
                    lib.spread_defer(True, dev)
            finally:
                lib.spread_lane(0)

    def end_stage(self, dev):
        """flush both lanes; the main stream then waits for the side stream: every parameter gradient of the stage is complete, stream-ordered"""
        self._wg_flush()
        side = self._side(dev)
        try:
            if side is not None:
                lib.spread_lane(1)
                try:
                    with torch.cuda.stream(side):
                        lib.spread_defer(False, dev)
                finally:
                    lib.spread_lane(0)
                torch.cuda.current_stream(dev).wait_stream(side)
        finally:
            lib.spread_defer(False, dev)

    # ================================================= forward =================================================
    def _dual_path_fwd(self, G, d, B, T2, dim, save):
        S, npos = (B * T2, F2) if dim == 4 else (B * F2, T2)
        L = npos - 7
        dev = G.device
        save.G_in = G  # kept for the adjoint: the stage's output goes to a new buffer (rtfs_dp_convt_fwd_to; in place, G had to be copied first)
        save.U, save.h, save.c = [], [], []
        U0 = torch.empty(S * L * 256, device=dev)
        self._call("rtfs_dp_unfold_gemm_fwd", G, d["g"], d["b"], d["w0"], U0, B, T2, dim, 0)
        h = torch.empty(S * L * 64, device=dev)
        c = torch.empty_like(h)
        l0 = d["layers"][0]
        self._call("rtfs_sru_scan_train_fwd", U0, None, l0["wc"], l0["bias"], l0["scale_x"], h, c, S, L, 4)
        save.U.append(U0), save.h.append(h), save.c.append(c)
        for lw in d["layers"][1:]:
            U = torch.empty(S * L * 192, device=dev)
            h2, c2 = torch.empty_like(h), torch.empty_like(h)
            self._call("rtfs_sru_layer_fwd", h, lw["w"], lw["wc"], lw["bias"], lw["scale_x"], h2, c2, U, S, L)  # projection fused, U / c saved
            save.U.append(U), save.h.append(h2), save.c.append(c2)
            h = h2
        Gout = torch.empty_like(G)
        self._call("rtfs_dp_convt_fwd_to", h, d["ct_w"], d["ct_b"], G, Gout, B, T2, dim)
        return Gout

    def _block_fwd(self, s_in, out, a0_or_none, bw, st, B, T, T2, y0=None, next_proj=None):
        """`y0`: this block's projection output when the previous block's residual kernel already produced it; `next_proj` = (y0 buffer,
        statistics slot) of the NEXT block: its gateway + projection then ride in this block's residual kernel (rtfs_resid_proj_fwd,
        shared block weights) - as in the inference path (hip_path.HipForward._block)"""
        dev = s_in.device
        TF = T * F_BINS
        full = lambda: torch.empty(B * TF * H, device=dev)  # noqa: E731
        low = lambda: torch.empty(B * T2 * F2 * H, device=dev)  # noqa: E731
        k = Ctx()
        k.s_in, k.st, k.has_a0 = s_in, st, a0_or_none is not None
        k.y0 = y0
        if y0 is None:
            k.y0 = full()
            self._call("rtfs_proj_fwd", s_in, bw["gw"], bw["gb"], bw["gslope"], bw["pw"], bw["pb"], k.y0, st[0], B, TF)
        d0w, d0b, d0g, d0be = bw["d0"]
        d1w, d1b, d1g, d1be = bw["d1"]
        k.D0, k.D1 = full(), low()
        self._call("rtfs_dwconv_fwd", k.y0, st[0], bw["pg"], bw["pbe"], bw["pslope"], 2, 1, 1, [d0w], [d0b], [k.D0], [st[1]], B, T, F_BINS)
        G = low()
        # D1's stride-2 convolution, the pooling and fusion_layers[0]'s local embedding in one pass over D0 (as in hip_path.HipForward._block)
        k.l0, pooled = full(), low()
        self._call("rtfs_dwconv_trio_fwd", k.D0, st[1], d0g, d0be, bw["fusion_layers.0.local_embedding"][0], k.l0, st[3], d1w, d1b, k.D1, st[2], pooled,
                   B, T, T2)
        self._call("rtfs_pool_add_fwd", pooled, k.D1, st[2], d1g, d1be, G, B, T2)
        del pooled
        k.dp = [Ctx(), Ctx()]
        G = self._dual_path_fwd(G, bw["dp0"], B, T2, 4, k.dp[0])
        G = self._dual_path_fwd(G, bw["dp1"], B, T2, 3, k.dp[1])
        a = bw["attn"]
        k.G2 = G  # the attention's input, kept for the adjoint (rtfs_attn_out_fwd_to writes a new buffer)
        k.Q = torch.empty(B * 4 * T2 * 256, device=dev)
        k.K = torch.empty_like(k.Q)
        k.V = torch.empty(B * 4 * T2 * 1024, device=dev)
        k.Ypre96 = torch.empty(B * T2 * 64 * 96, device=dev)
        self._call("rtfs_attn_qkv_fwd", G, a["w"], a["bias"], a["slope"], a["gq"], a["bq"], a["gk"], a["bk"], a["gv"], a["bv"], k.Q, k.K, k.V, k.Ypre96, B, T2)
        k.O = torch.empty(B * T2 * 4096, device=dev)
        k.LSE = torch.empty(B * 4 * T2, device=dev)
        self._call("rtfs_attn_core_fwd", k.Q, k.K, k.V, k.O, k.LSE, B, T2)
        k.Ypre_o = torch.empty(B * T2 * 4096, device=dev)
        G = torch.empty_like(k.G2)
        self._call("rtfs_attn_out_fwd_to", k.O, a["ow"], a["ob"], a["oslope"], a["og"], a["obe"], k.G2, G, k.Ypre_o, B, T2)
        k.G3 = G
        f0l, f0g, f0gate = bw["fusion_layers.0.local_embedding"], bw["fusion_layers.0.global_embedding"], bw["fusion_layers.0.global_gate"]
        f1l, f1g, f1gate = bw["fusion_layers.1.local_embedding"], bw["fusion_layers.1.global_embedding"], bw["fusion_layers.1.global_gate"]
        cl_, cg_, cgate_ = bw["concat_layers.0.local_embedding"], bw["concat_layers.0.global_embedding"], bw["concat_layers.0.global_gate"]
        k.l1 = low()
        self._call("rtfs_dwconv_fwd", k.D1, st[2], d1g, d1be, 0.0, 1, 1, 1, [f1l[0]], [None], [k.l1], [st[4]], B, T2, F2)
        k.g0, k.gg0, k.g1, k.gg1 = low(), low(), low(), low()
        self._call("rtfs_dwconv_fwd", G, None, None, None, 0.0, 0, 1, 4, [f0g[0], f0gate[0], f1g[0], f1gate[0]], [None] * 4, [k.g0, k.gg0, k.g1, k.gg1],
                 [st[5], st[6], st[7], st[8]], B, T2, F2)
        k.cl, k.cg, k.cgate = full(), low(), low()
        fz = self.model._hip.fuse
        if fz["dwadj"] and fz["mixgln"] and fz["mix"]:
            # round 6: the fusion layers' outputs F0 / F1 are not materialised for the step either - the concat layer's convolutions form them in their staging
            # as in the inference path, and their adjoint re-forms them per pixel (rtfs_dw_adjoint, input mode 3)
            k.F0 = k.F1 = None
            self._call("rtfs_dwconv_mix_fwd", k.l0, st[3], f0l[2], f0l[3], k.gg0, st[6], f0gate[2], f0gate[3], k.g0, st[5], f0g[2], f0g[3], 1, [cl_[0]], [None],
                       [k.cl], [st[9]], B, T, F_BINS, T2, F2)
            self._call("rtfs_dwconv_mix_fwd", k.l1, st[4], f1l[2], f1l[3], k.gg1, st[8], f1gate[2], f1gate[3], k.g1, st[7], f1g[2], f1g[3], 2, [cg_[0], cgate_[0]],
                       [None, None], [k.cg, k.cgate], [st[10], st[11]], B, T2, F2, T2, F2)
        else:
            k.F0, k.F1 = full(), low()
            self._call("rtfs_tfar_mix_fwd", k.l0, st[3], f0l[2], f0l[3], k.gg0, st[6], f0gate[2], f0gate[3], k.g0, st[5], f0g[2], f0g[3], k.F0, B, T, F_BINS, T2, F2)
            self._call("rtfs_tfar_mix_fwd", k.l1, st[4], f1l[2], f1l[3], k.gg1, st[8], f1gate[2], f1gate[3], k.g1, st[7], f1g[2], f1g[3], k.F1, B, T2, F2, T2, F2)
            self._call("rtfs_dwconv_fwd", k.F0, None, None, None, 0.0, 0, 1, 1, [cl_[0]], [None], [k.cl], [st[9]], B, T, F_BINS)
            self._call("rtfs_dwconv_fwd", k.F1, None, None, None, 0.0, 0, 1, 2, [cg_[0], cgate_[0]], [None, None], [k.cg, k.cgate], [st[10], st[11]], B, T2, F2)
        if next_proj is not None and a0_or_none is not None:
            self._call("rtfs_resid_proj_fwd", k.cl, st[9], cl_[2], cl_[3], k.D0, st[1], d0g, d0be, k.cg, st[10], cg_[2], cg_[3], k.cgate, st[11], cgate_[2],
                       cgate_[3], bw["rw"], bw["rb"], s_in, bw["gw"], bw["gb"], bw["gslope"], a0_or_none, out, bw["pw"], bw["pb"], next_proj[0], next_proj[1],
                       B, T, T2, 0)
            k.fused_next = True
        else:
            self._call("rtfs_resid_fwd", k.cl, st[9], cl_[2], cl_[3], k.D0, st[1], d0g, d0be, k.cg, st[10], cg_[2], cg_[3], k.cgate, st[11], cgate_[2],
                       cgate_[3], bw["rw"], bw["rb"], s_in, bw["gw"], bw["gb"], bw["gslope"], a0_or_none, out, B, T, T2)
            k.fused_next = False
        return k

    def forward(self, wav, att, rsz):
        """wav [B,L]; att, rsz [B,Tv,256] (video side of the CAF cell, torch glue).  Returns (out [B,1,L], ctx)."""
        c = self.forward_a(wav)
        return self.forward_b(c, att, rsz), c

    def forward_a(self, wav):
        """STFT, encoder conv, bottleneck, RTFS block 0 (everything before the CAF cell) -> ctx with x0, a0, a_emb."""
        m = self.model
        wav = wav.to(torch.float32).contiguous()
        B, L = wav.shape
        T = 1 + L // 128
        T2 = (T - 2) // 2 + 1
        if T2 < 8:
            raise ValueError("input too short for the HIP path: need at least 16 STFT frames (L >= 1920 samples)")
        if T * F_BINS * C * 4 > 2 ** 30:  # (the inference path's guard, hip_path.HipForward: 32-bit offsets inside an utterance, tested envelope)
            raise ValueError(f"training segment too long for the HIP path: {L} samples (about 65 s at most)")
        pw = self.weights(defer_scalars=True)  # (scalar slots are read from block 0 on: finish_scalars() below, behind the first three launches)
        w = pw.w
        TF = T * F_BINS
        dev = wav.device
        R = m.refinement_module.audio_net.repeats
        c = Ctx()
        c.pw = pw  # the SAME prepared weights serve forward_b and both backward stages of this step (parameters do not change inside a step)
        c.pw_version = pw.version  # ... and _same_weights() holds the later stages to it: GatherTrainWeights refreshes its buffers IN PLACE
        c.B, c.L, c.T, c.T2, c.R = B, L, T, T2, R
        c.stats = torch.zeros(1 + 12 * R, B, lib.STAT_STRIDE, dtype=torch.float64, device=dev)
        stats = c.stats
        c.spec = torch.empty(B * TF * 2, device=dev)
        self._call("rtfs_stft_fwd", wav, c.spec, B, L)
        c.a_emb = torch.empty(B * TF * C, device=dev)
        self._call("rtfs_enc_conv_fwd", c.spec, w["enc"], c.a_emb, stats[0], B, T)
        c.a0 = torch.empty_like(c.a_emb)
        self._call("rtfs_bottleneck_fwd", c.a_emb, stats[0], w["bn_g"], w["bn_b"], w["bn_w"], w["bn_bias"], c.a0, B, TF)
        if isinstance(pw, GatherTrainWeights):
            pw.finish_scalars()  # the device has 1.3 ms of work queued: the host waits for the scalars without starving it
        blocks = pw.blocks
        bw = lambda i: blocks[0] if len(blocks) == 1 else blocks[i]  # noqa: E731
        c.blk = []
        x = torch.empty_like(c.a_emb)
        c.blk.append(self._block_fwd(c.a0, x, None, bw(0), stats[1:13], B, T, T2))
        c.x0 = x
        return c

    def forward_b(self, c, att, rsz):
        """CAF cell, RTFS blocks 1..R-1, S3 mask, decoder, iSTFT -> out [B,1,L]."""
        m = self.model
        self._same_weights(c, "second forward stage")
        pw = c.pw
        w = pw.w
        B, L, T, T2, R = c.B, c.L, c.T, c.T2, c.R
        TF = T * F_BINS
        dev = c.x0.device
        stats, x = c.stats, c.x0
        blocks = pw.blocks
        bw = lambda i: blocks[0] if len(blocks) == 1 else blocks[i]  # noqa: E731
        c.Tv = att.shape[1]
        # CAF with training-mode BatchNorm2d (batch statistics over B,T,F of the depth-wise conv output)
        if getattr(self, "video_stream", None) is not None:  # att / rsz were produced on the glue stream (AVNet._forward_autograd)
            torch.cuda.current_stream().wait_stream(self.video_stream)
            self.video_stream = None
        c.att, c.rsz = att.contiguous(), rsz.contiguous()
        c.caf = self._caf_coeffs(x, w, B * TF, m.training)
        s = torch.empty_like(c.a_emb)
        last = R == 1
        self._call("rtfs_caf_fuse_fwd", x, c.caf["ks"], c.caf["kb"], c.caf["vs"], c.caf["vb"], c.att, c.rsz, None if last else c.a0, s, B, T, c.Tv)
        fuse = len(blocks) == 1  # shared block weights: block i+1's gateway / projection weights are block i's
        y0_next = None
        for i in range(1, R):
            last = i == R - 1
            nxt = torch.empty_like(c.a_emb)
            np_ = (torch.empty(B * TF * H, device=dev), stats[1 + 12 * (i + 1)]) if (fuse and not last) else None
            blk = self._block_fwd(s, nxt, None if last else c.a0, bw(i), stats[1 + 12 * i: 13 + 12 * i], B, T, T2, y0=y0_next, next_proj=np_)
            y0_next = np_[0] if blk.fused_next else None
            c.blk.append(blk)
            s = nxt
        c.refined = s
        c.masked = torch.empty_like(c.a_emb)
        c.m = torch.empty_like(c.a_emb)
        self._call("rtfs_mask_fwd", s, w["mask_slope"], w["mask_w"], w["mask_b"], c.a_emb, c.masked, c.m, B, TF)
        tapbuf = torch.empty(B * TF * 32, device=dev)
        self._call("rtfs_gemm_rows", c.masked, w["dec_w"], None, tapbuf, B * TF, 256, 32, 0)
        frames = torch.empty(B * T * 256, device=dev)
        out = torch.empty(B, L, device=dev)
        self._call("rtfs_istft_fwd", tapbuf, frames, out, B, L)
        return out.view(B, 1, L)

    def _caf_coeffs(self, x, w, rows, training):
        """folded (scale, shift) of key/value BatchNorm2d; training: batch statistics (+ running-stat update)."""
        m = self.model
        cell = m.refinement_module.crossmodal_fusion.get_fusion_block(0).audio_lstm
        out = {"training": training}
        bk, bv = cell.key_embed.full_layer[3], cell.value_embed.full_layer[3]
        if (training and m._hip.fuse.get("cafbn", True) and bk.momentum == bv.momentum and bk.eps == bv.eps and bk.running_mean is not None
                and bk.running_mean.dtype == torch.float32):
            # one launch for the per-channel arithmetic (csrc/optim.hip caf_bn_prepare_kernel; ~47 tiny torch launches on the critical path before)
            sums = torch.zeros(2, C, dtype=torch.float64, device=x.device)
            self._call("rtfs_chan_stats", x, sums[0], sums[1], rows)
            out["sync"] = (torch.distributed.is_available() and torch.distributed.is_initialized() and isinstance(bk, torch.nn.SyncBatchNorm))
            if out["sync"]:  # train.py:145 sync_batchnorm=True: statistics of the union of all ranks
                buf = torch.cat([sums.reshape(-1), sums.new_full((1,), float(rows))])
                torch.distributed.all_reduce(buf)
                glob, nptr = buf[:2 * C], buf[2 * C:]
            else:
                glob, nptr = sums, sums.new_full((1,), float(rows))
            o = torch.empty(12, C, device=x.device)
            lib.call("rtfs_caf_bn_prepare", glob, sums, nptr,
                     w["caf_key_dw"], w["caf_key_g"], w["caf_key_be"], bk.running_mean, bk.running_var, bk.num_batches_tracked, o[0], o[1], o[2], o[3],
                     w["caf_value_dw"], w["caf_value_g"], w["caf_value_be"], bv.running_mean, bv.running_var, bv.num_batches_tracked, o[4], o[5], o[6], o[7],
                     float(bk.momentum if bk.momentum is not None else 0.1), float(bk.eps), o[8], o[9], o[10], o[11])
            # (written through raw pointers: the inference path keys its folded CAF weights on these buffers' version counters)
            torch.autograd.graph.increment_version([bk.running_mean, bk.running_var, bk.num_batches_tracked, bv.running_mean, bv.running_var, bv.num_batches_tracked])
            out.update({"key_inv": o[0], "key_mean_u": o[1], "ks": o[2], "kb": o[3], "value_inv": o[4], "value_mean_u": o[5], "vs": o[6], "vb": o[7],
                        "mean_x": o[8], "var_x": o[9], "lsum": o[10], "lsq": o[11], "n": nptr, "fused": True})
            return out
        if training:
            sums = torch.zeros(2, C, dtype=torch.float64, device=x.device)
            self._call("rtfs_chan_stats", x, sums[0], sums[1], rows)
            n = float(rows)
            out["sync"] = (torch.distributed.is_available() and torch.distributed.is_initialized()
                           and isinstance(cell.key_embed.full_layer[3], torch.nn.SyncBatchNorm))  # train.py:145 sync_batchnorm=True
            mean_x, var_x, n, lsum, lsq = caf_bn_batch_stats(sums, n, out["sync"])
            out["mean_x"], out["var_x"], out["n"], out["lsum"], out["lsq"] = mean_x, var_x, n, lsum, lsq  # lsq: sum_local (x-mean) x
        for tag, mod in (("key", cell.key_embed), ("value", cell.value_embed)):
            dw, g, be = w[f"caf_{tag}_dw"], w[f"caf_{tag}_g"], w[f"caf_{tag}_be"]
            bn = mod.full_layer[3]
            if training:
                mean_u, var_u = dw * out["mean_x"], dw * dw * out["var_x"]
                with torch.no_grad():
                    mom = bn.momentum if bn.momentum is not None else 0.1
                    bn.running_mean.mul_(1 - mom).add_(mom * mean_u)
                    nn_ = out["n"]
                    unbias = nn_ / torch.clamp(nn_ - 1, min=1) if isinstance(nn_, torch.Tensor) else nn_ / max(nn_ - 1, 1)
                    bn.running_var.mul_(1 - mom).add_((mom * var_u * unbias).to(bn.running_var.dtype))
                    bn.num_batches_tracked += 1
            else:
                mean_u, var_u = bn.running_mean.float(), bn.running_var.float()
            inv = torch.rsqrt(var_u + bn.eps)
            out[tag + "_inv"], out[tag + "_mean_u"] = inv, mean_u
            s = dw * g * inv
            out["ks" if tag == "key" else "vs"] = s.contiguous()
            out["kb" if tag == "key" else "vb"] = (be - mean_u * g * inv).contiguous()
        return out

    # ================================================= backward ================================================
    def _gln_bwd(self, dN, X, st, gamma, beta, dX, accumulate, gr, key, B, rows, Cc=H, act=0, slope=0.0, dslope=None):
        """dN: gradient w.r.t. (act of) the normalised tensor; X: pre-norm; result dX (= or +=); gamma/beta grads into gr[key]."""
        dev = dN.device
        dg, db = _acc(gr, key + ".g", Cc, dev), _acc(gr, key + ".b", Cc, dev)
        red = gr["_pool"].take(B * lib.STAT_STRIDE, torch.float64).view(B, lib.STAT_STRIDE)
        self._call("rtfs_gln_bwd_reduce", dN, X, st, gamma, beta, act, slope, red, dg, db, dslope, B, rows, Cc)
        self._call("rtfs_gln_bwd_apply", dN, X, st, gamma, beta, act, slope, red, dX, 1 if accumulate else 0, B, rows, Cc)

    def _mix_gln_bwd(self, dOut, loc, gate, glob, dLoc, dGate, dGlob, gr, B, T, F, Tg, Fg, defer_apply=False):
        """adjoint of `n(loc) * sigmoid(n(gate))^ + n(glob)^` (fusion.py:59-67) together with the gLN adjoints of its three embeddings.
        loc / gate / glob: (pre-norm tensor, statistics slot, conv tuple (.., .., gamma, beta), gr key); dLoc / dGate / dGlob: gradients w.r.t.
        the three convs' OUTPUTS; gamma / beta grads into gr[key].
        defer_apply: the apply passes of all three branches' gLN adjoints are left to the consumers (rtfs_dw_adjoint / rtfs_dw_adjoint_mix apply them on load):
        dLoc is not written at all (the local branch's consumer re-forms dOut * sigmoid(n(gate))^ itself), dGate / dGlob receive the gradients w.r.t. the NORMALISED
        outputs, and the call returns the three (S1, S2) slots -> (red_loc, red_gate, red_glob)."""
        dev = dOut.device
        dgb = [_acc(gr, br[3] + sfx, H, dev) for br in (loc, gate, glob) for sfx in (".g", ".b")]
        low_rows = Tg * Fg
        if self.model._hip.fuse["mixgln"]:
            red = gr["_pool"].take(3 * B * lib.STAT_STRIDE, torch.float64).view(3, B, lib.STAT_STRIDE)
            if defer_apply:
                sig = torch.empty_like(dGate)  # sigmoid(gLN(gate)): the reduce pass forms it, rtfs_dw_adjoint_mix multiplies the mix's gradient by it on load
                self._call("rtfs_mix_gln_bwd_sig", dOut, loc[0], loc[1], loc[2][2], loc[2][3], gate[0], gate[1], gate[2][2], gate[2][3], glob[0], glob[1], glob[2][2],
                           glob[2][3], None, dGate, dGlob, sig, red, dgb, B, T, F, Tg, Fg)
                return (red[0], sig), red[1], red[2]
            dNgate, dNglob = torch.empty_like(dGate), torch.empty_like(dGlob)
            self._call("rtfs_mix_gln_bwd", dOut, loc[0], loc[1], loc[2][2], loc[2][3], gate[0], gate[1], gate[2][2], gate[2][3], glob[0], glob[1], glob[2][2],
                       glob[2][3], dLoc, dNgate, dNglob, red, dgb, B, T, F, Tg, Fg)
            self._call("rtfs_gln_bwd_apply", dNgate, gate[0], gate[1], gate[2][2], gate[2][3], 0, 0.0, red[1], dGate, 0, B, low_rows, H)
            self._call("rtfs_gln_bwd_apply", dNglob, glob[0], glob[1], glob[2][2], glob[2][3], 0, 0.0, red[2], dGlob, 0, B, low_rows, H)
            return
        dNloc, dNgate, dNglob = torch.empty(B * T * F * H, device=dev), torch.empty_like(dGate), torch.empty_like(dGlob)
        self._call("rtfs_mix_bwd", dOut, loc[0], loc[1], loc[2][2], loc[2][3], gate[0], gate[1], gate[2][2], gate[2][3], dNloc, dNgate, dNglob, B, T, F, Tg, Fg)
        self._gln_bwd(dNloc, loc[0], loc[1], loc[2][2], loc[2][3], dLoc, False, gr, loc[3], B, T * F)
        self._gln_bwd(dNgate, gate[0], gate[1], gate[2][2], gate[2][3], dGate, False, gr, gate[3], B, low_rows)
        self._gln_bwd(dNglob, glob[0], glob[1], glob[2][2], glob[2][3], dGlob, False, gr, glob[3], B, low_rows)

    def _dw_bwd(self, dOut, conv, inp, in_st, in_g, in_b, in_slope, mode, stride, dIn, accumulate, gr, key, B, Tin, Fin, has_bias):
        """depth-wise conv adjoint: input gradient (w.r.t. the transformed input) and tap/bias gradients."""
        dev = dOut.device
        dW = _acc(gr, key + ".w", 16 * 64, dev)
        dbias = _acc(gr, key + ".bias", 64, dev) if has_bias else None
        self._wg("rtfs_dwconv_bwd_weight", dOut, inp, in_st, in_g, in_b, in_slope, mode, stride, dW, dbias, B, Tin, Fin)
        if dIn is not None:
            self._call("rtfs_dwconv_bwd_input", dOut, conv[0], dIn, 1 if accumulate else 0, stride, B, Tin, Fin)

    def _dw_adjoint(self, convs, inp, in_st, in_g, in_b, in_slope, mode, dIn, accumulate, gr, B, T, F, bias=False, in_mix=None, in_low=(0, 0)):
        """rtfs_dw_adjoint (csrc/bwd_dw.hip): the whole adjoint of the 1 / 2 / 4 stride-1 depth-wise convolutions `convs` that read `inp` in one launch.
        convs: list of (dY, conv tuple (taps, bias, gamma, beta), gr key, None | (pre-norm output, statistics slot, (S1, S2) slot)) - with the last entry the
        gLN adjoint is applied on load and dY is the gradient w.r.t. the normalised output.  mode 3: `inp` is the local tensor of a TFAR mix, in_mix = [gate, its
        statistics, gamma, beta, glob, its statistics, gamma, beta] at resolution in_low."""
        dev = dIn.device
        gln = convs[0][3] is not None
        dW = [_acc(gr, c[2] + ".w", 16 * 64, dev) for c in convs]
        db = [_acc(gr, c[2] + ".bias", 64, dev) for c in convs] if bias else None
        lib.call("rtfs_dw_adjoint", len(convs), [c[0] for c in convs], [c[3][0] for c in convs] if gln else None, [c[3][1] for c in convs] if gln else None,
                 [c[3][2] for c in convs] if gln else None, [c[1][2] for c in convs] if gln else None, [c[1][0] for c in convs], inp, in_st, in_g, in_b,
                 float(in_slope), mode, in_mix, in_low[0], in_low[1], dIn, 1 if accumulate else 0, dW, db, B, T, F)

    def _dw_adjoint_mix(self, dOut, loc, gate, red_loc, Tg, Fg, inp, in_st, in_g, in_b, mode, dIn, accumulate, gr, B, T, F, in_mix=None, in_low=(0, 0)):
        """rtfs_dw_adjoint_mix: the adjoint of an InjectionMultiSum's LOCAL embedding convolution straight from the gradient of the mix's output (the local branch's
        mix + gLN adjoint on load).  loc / gate: (pre-norm tensor, statistics slot, conv tuple, gr key) as in _mix_gln_bwd."""
        lib.call("rtfs_dw_adjoint_mix", dOut, loc[0], loc[1], red_loc[0], loc[2][2], red_loc[1], Tg, Fg, loc[2][0], inp, in_st, in_g, in_b,
                 0.0, mode, in_mix, in_low[0], in_low[1], dIn, 1 if accumulate else 0, _acc(gr, loc[3] + ".w", 16 * 64, dIn.device), B, T, F)

    def _sru_bwd_work(self, S, dev):
        """the workgroups' partial-dW scratch of rtfs_sru_layer_bwd: one buffer per device, reused by every layer of every step (main stream only)"""
        n = lib.load().rtfs_sru_layer_bwd_work_floats(S)
        cache = self.__dict__.setdefault("_sru_work", {})
        buf = cache.get(dev)
        if buf is None or buf.numel() < n:
            buf = cache[dev] = torch.empty(n, device=dev)
        return buf

    def _dual_path_bwd(self, dG, d, sv, B, T2, dim, gr, key):
        """dG: gradient w.r.t. the stage output (G layout), updated IN PLACE to the gradient w.r.t. the stage input."""
        S, npos = (B * T2, F2) if dim == 4 else (B * F2, T2)
        L = npos - 7
        dev = dG.device
        g = lambda name, n: _acc(gr, f"{key}.{name}", n, dev)  # noqa: E731
        # ConvTranspose1d + bias + residual
        dG_seq = torch.empty(S * npos * 64, device=dev)
        self._call("rtfs_seq_gather", dG, None, None, 0, dG_seq, B, T2, dim)
        dct = g("ct_w", 64 * 512)
        self._wg_later("rtfs_wgrad", dG_seq, 64, sv.h[3], 64, dct, 512, g("ct_b", 64), S * npos, npos, L, -7, 8, 64, 64, 0, None, None, 0.0, None, 0)
        dh = torch.empty(S * L * 64, device=dev)
        self._call("rtfs_convt_bwd_input", dG, d["ctbi_w"], dh, B, T2, dim)
        # SRU layers 3..1.  fp32 (and the six-term mode, whose backward GEMMs are the fp32 kernels): one launch per layer - recurrence adjoint, weight
        # gradient and input gradient with dU held in LDS (rtfs_sru_layer_bwd); the input gradient arrives as two parts (one per scan direction) that
        # the next layer down adds on load.  Otherwise the three launches (dU through HBM).
        one_launch = self.model._hip.fuse["srubwd"] and self.prec in (0, 6) and S * L < 2_790_000
        dh2 = None
        for l in (3, 2, 1):
            lw = d["layers"][l]
            if one_launch:
                dxa, dxb = torch.empty(S * L * 64, device=dev), torch.empty(S * L * 64, device=dev)
                work = self._sru_bwd_work(S, dev)
                lib.call("rtfs_sru_layer_bwd", sv.U[l], sv.h[l - 1], sv.c[l], lw["w"], lw["wc"], lw["bias"], lw["scale_x"], dh, dh2, dxa, dxb, work,
                         g(f"l{l}.w", 192 * 64), g(f"l{l}.wc", 128), g(f"l{l}.bias", 128), S, L)
                dh, dh2 = dxa, dxb
                continue
            dU = torch.empty(S * L * 192, device=dev)
            dx = torch.empty(S * L * 64, device=dev)
            self._call("rtfs_sru_scan_bwd2", sv.U[l], sv.h[l - 1], sv.c[l], lw["wc"], lw["bias"], lw["scale_x"], dh, dh2, dU, dx, g(f"l{l}.wc", 128),
                       g(f"l{l}.bias", 128), S, L, 3)
            self._wg("rtfs_wgrad", dU, 192, sv.h[l - 1], 64, g(f"l{l}.w", 192 * 64), 64, None, S * L, 0, 0, 0, 1, 192, 64, 0, None, None, 0.0, None, 0)
            self._call("rtfs_gemm_rows", dU, lw["wT"], None, dx, S * L, 192, 64, 1)  # dx += dU . W
            dh, dh2 = dx, None
        l0 = d["layers"][0]
        dU0 = torch.empty(S * L * 256, device=dev)
        self._call("rtfs_sru_scan_bwd2", sv.U[0], None, sv.c[0], l0["wc"], l0["bias"], l0["scale_x"], dh, dh2, dU0, None, g("l0.wc", 128), g("l0.bias", 128),
                   S, L, 4)
        # layer-0 GEMM: weight gradient over the Toeplitz windows, input gradient by folding
        xn_seq = torch.empty(S * npos * 64, device=dev)
        self._call("rtfs_seq_gather", sv.G_in, d["g"], d["b"], 1, xn_seq, B, T2, dim)
        dw0 = g("w0", 256 * 512)
        self._wg_later("rtfs_wgrad", dU0, 256, xn_seq, 64, dw0, 512, None, S * L, L, npos, 0, 8, 256, 64, 0, None, None, 0.0, None, 0)
        dxn = torch.empty(B * T2 * F2 * 64, device=dev)
        self._call("rtfs_fold_gemm_bwd", dU0, d["fold_w"], dxn, B, T2, dim)
        self._call("rtfs_ln4d_c_bwd", dxn, sv.G_in, d["g"], dG, g("g", 64), g("b", 64), B * T2 * F2)  # dG += LN adjoint (residual already in dG)

    def _attn_bwd(self, dG, a, k, B, T2, gr, key):
        """dG: gradient w.r.t. the attention output, updated in place to the gradient w.r.t. its input."""
        dev = dG.device
        g = lambda name, n: _acc(gr, f"{key}.{name}", n, dev)  # noqa: E731
        ntok = B * T2
        rows = ntok * 64
        dYo = torch.empty(rows * 64, device=dev)
        self._call("rtfs_attn_out_norm_bwd", dG, k.Ypre_o, a["oslope"], a["og"], dYo, g("og", 4096), g("obe", 4096), g("oslope", 1), ntok)
        Ocl = torch.empty(rows * 64, device=dev)
        self._call("rtfs_transpose_tok", k.O, Ocl, ntok)  # [c][f] -> [f][c]
        self._wg("rtfs_wgrad", dYo, 64, Ocl, 64, g("ow", 64 * 64), 64, g("ob", 64), rows, 0, 0, 0, 1, 64, 64, 0, None, None, 0.0, None, 0)
        dOcl = torch.empty(rows * 64, device=dev)
        self._call("rtfs_gemm_rows", dYo, a["owT"], None, dOcl, rows, 64, 64, 0)
        dO = torch.empty(rows * 64, device=dev)
        self._call("rtfs_transpose_tok", dOcl, dO, ntok)  # back to the O layout [c][f]
        dQ, dK, dV = torch.empty_like(k.Q), torch.empty_like(k.K), torch.empty_like(k.V)
        Dws = torch.empty(B * 4 * T2, device=dev)
        self._call("rtfs_attn_core_bwd", k.Q, k.K, k.V, k.O, dO, k.LSE, Dws, dQ, dK, dV, B, T2)
        dY96 = torch.empty(rows * 96, device=dev)
        self._call("rtfs_attn_qkv_norm_bwd", dQ, dK, dV, k.Ypre96, a["slope"], a["gq"], a["gk"], a["gv"], dY96, g("gq", 1024), g("bq", 1024), g("gk", 1024),
                 g("bk", 1024), g("gv", 4096), g("bv", 4096), g("slope", 12), B, T2)
        self._wg("rtfs_wgrad", dY96, 96, k.G2, 64, g("w", 96 * 64), 64, g("bias", 96), rows, 0, 0, 0, 1, 96, 64, 0, None, None, 0.0, None, 0)
        self._call("rtfs_gemm_rows", dY96, a["wT"], None, dG, rows, 96, 64, 1)  # dG (residual) += dY96 . Wqkv

    def _block_bwd(self, dx, k, bw, B, T, T2, gr, da0, a0_mode, tag="blk.", dE_in=None, next_rwT=None):
        """dx: gradient w.r.t. the block output [B,TF,256] (overwritten).  Returns ds (gradient w.r.t. the block input).
        tag: prefix of this block's gradient accumulators in `gr` ("blk." for the shared block, "blk<i>." for block i of a non-shared stack).
        a0_mode: how ds also enters the running d(a0) sum `da0` -- 0: not at all (the caller sums the blocks' ds itself), 1: da0 = ds, 2: da0 += ds (blocks whose input was
        `previous + a0`), 3: block 0, whose input IS a0: ds is accumulated straight into da0, 4: same but da0 is empty (R = 1)."""
        dev = dx.device
        TF, lo = T * F_BINS, T2 * F2
        st = k.st
        full = lambda: torch.empty(B * TF * H, device=dev)  # noqa: E731
        low = lambda: torch.empty(B * lo * H, device=dev)  # noqa: E731
        g = lambda name, n: _acc(gr, f"{tag}{name}", n, dev)  # noqa: E731
        d0w, d0b, d0g, d0be = bw["d0"]
        d1w, d1b, d1g, d1be = bw["d1"]
        f0l, f0g, f0gate = bw["fusion_layers.0.local_embedding"], bw["fusion_layers.0.global_embedding"], bw["fusion_layers.0.global_gate"]
        f1l, f1g, f1gate = bw["fusion_layers.1.local_embedding"], bw["fusion_layers.1.global_embedding"], bw["fusion_layers.1.global_gate"]
        cl_, cg_, cgate_ = bw["concat_layers.0.local_embedding"], bw["concat_layers.0.global_embedding"], bw["concat_layers.0.global_gate"]
        # residual_conv: bias, weight (needs `expanded`), input gradient
        E = full()
        # (round 6: `expanded` is re-formed ON THE SIDE STREAM - its only reader is the weight-gradient launch behind it there, the adjoint chain does not wait for it)
        self._wg("rtfs_expand_fwd", k.cl, st[9], cl_[2], cl_[3], k.D0, st[1], d0g, d0be, k.cg, st[10], cg_[2], cg_[3], k.cgate, st[11], cgate_[2], cgate_[3], E, B, T, T2)
        self._wg("rtfs_wgrad", dx, C, E, H, g("rw", C * H), H, g("rb", C), B * TF, 0, 0, 0, 1, C, H, 0, None, None, 0.0, None, 0)
        if dE_in is not None:  # formed by the previous block's last kernel from its ds rows (rtfs_proj_gateway_bwd_next)
            dE = dE_in
        else:
            dE = full()
            self._call("rtfs_gemm_rows", dx, bw["rwT"], None, dE, B * TF, C, H, 0)
        # expanded = n(cl)*sigmoid(n(cgate))^ + n(cg)^ + n(D0):  dN_D0 starts as dE itself (dE has no reader after rtfs_mix_bwd: no copy)
        dN_D0 = dE
        # concat layer: mix + gLN adjoints, then conv adjoints (inputs F0 / F1 are raw tensors)
        dcg, dcgate = low(), low()
        # round 6: the adjoint of every stride-1 depth-wise convolution group in ONE launch (rtfs_dw_adjoint: tap gradients + input gradient from one dX tile in LDS,
        # the gLN adjoint's apply pass on load - dX never reaches HBM, convolutions that share an input share the launch); `fuse["dwadj"]` off = the round-5 launches
        dwadj = self.model._hip.fuse["dwadj"] and self.model._hip.fuse["mixgln"]
        c_loc, c_gate = (k.cl, st[9], cl_, tag + "cl"), (k.cgate, st[11], cgate_, tag + "cgate")
        dcl = None if dwadj else full()
        reds = self._mix_gln_bwd(dE, c_loc, c_gate, (k.cg, st[10], cg_, tag + "cg"), dcl, dcgate, dcg, gr, B, T, F_BINS, T2, F2, defer_apply=dwadj)
        dF0, dF1 = full(), low()
        if dwadj:
            convs_b = [(dcg, cg_, tag + "cg", (k.cg, st[10], reds[2])), (dcgate, cgate_, tag + "cgate", (k.cgate, st[11], reds[1]))]
            if k.F0 is None:  # the concat layer's inputs re-formed from the fusion layers' three tensors each (input mode 3)
                self._dw_adjoint_mix(dE, c_loc, c_gate, reds[0], T2, F2, k.l0, st[3], f0l[2], f0l[3], 3, dF0, False, gr, B, T, F_BINS,
                                     in_mix=[k.gg0, st[6], f0gate[2], f0gate[3], k.g0, st[5], f0g[2], f0g[3]], in_low=(T2, F2))
                self._dw_adjoint(convs_b, k.l1, st[4], f1l[2], f1l[3], 0.0, 3, dF1, False, gr, B, T2, F2,
                                 in_mix=[k.gg1, st[8], f1gate[2], f1gate[3], k.g1, st[7], f1g[2], f1g[3]], in_low=(T2, F2))
            else:
                self._dw_adjoint_mix(dE, c_loc, c_gate, reds[0], T2, F2, k.F0, None, None, None, 0, dF0, False, gr, B, T, F_BINS)
                self._dw_adjoint(convs_b, k.F1, None, None, None, 0.0, 0, dF1, False, gr, B, T2, F2)
        else:
            self._dw_bwd(dcl, cl_, k.F0, None, None, None, 0.0, 0, 1, dF0, False, gr, tag + "cl", B, T, F_BINS, False)
            self._dw_bwd(dcg, cg_, k.F1, None, None, None, 0.0, 0, 1, dF1, False, gr, tag + "cg", B, T2, F2, False)
            self._dw_bwd(dcgate, cgate_, k.F1, None, None, None, 0.0, 0, 1, dF1, True, gr, tag + "cgate", B, T2, F2, False)
        # fusion layers' mixes
        dl0, dl1 = (None, None) if dwadj else (full(), low())
        dgs = [low() for _ in range(4)]  # gradients w.r.t. the outputs of f0g, f0gate, f1g, f1gate
        l0_loc, l0_gate = (k.l0, st[3], f0l, tag + "f0l"), (k.gg0, st[6], f0gate, tag + "f0gate")
        l1_loc, l1_gate = (k.l1, st[4], f1l, tag + "f1l"), (k.gg1, st[8], f1gate, tag + "f1gate")
        r0 = self._mix_gln_bwd(dF0, l0_loc, l0_gate, (k.g0, st[5], f0g, tag + "f0g"), dl0, dgs[1], dgs[0], gr, B, T, F_BINS, T2, F2, defer_apply=dwadj)
        r1 = self._mix_gln_bwd(dF1, l1_loc, l1_gate, (k.g1, st[7], f1g, tag + "f1g"), dl1, dgs[3], dgs[2], gr, B, T2, F2, T2, F2, defer_apply=dwadj)
        # conv adjoints: local embeddings feed D0n / D1n, the four global convs feed G3
        dN_D1 = low()
        dG = low()  # gradient w.r.t. G3 (attention output)
        if dwadj:
            self._dw_adjoint_mix(dF0, l0_loc, l0_gate, r0[0], T2, F2, k.D0, st[1], d0g, d0be, 1, dN_D0, True, gr, B, T, F_BINS)
            self._dw_adjoint_mix(dF1, l1_loc, l1_gate, r1[0], T2, F2, k.D1, st[2], d1g, d1be, 1, dN_D1, False, gr, B, T2, F2)
            self._dw_adjoint([(dgs[0], f0g, tag + "f0g", (k.g0, st[5], r0[2])), (dgs[1], f0gate, tag + "f0gate", (k.gg0, st[6], r0[1])),
                              (dgs[2], f1g, tag + "f1g", (k.g1, st[7], r1[2])), (dgs[3], f1gate, tag + "f1gate", (k.gg1, st[8], r1[1]))], k.G3, None, None, None, 0.0, 0,
                             dG, False, gr, B, T2, F2)
        else:
            self._dw_bwd(dl0, f0l, k.D0, st[1], d0g, d0be, 0.0, 1, 1, dN_D0, True, gr, tag + "f0l", B, T, F_BINS, False)
            self._dw_bwd(dl1, f1l, k.D1, st[2], d1g, d1be, 0.0, 1, 1, dN_D1, False, gr, tag + "f1l", B, T2, F2, False)
            for j, (conv, nm) in enumerate(((f0g, "f0g"), (f0gate, "f0gate"), (f1g, "f1g"), (f1gate, "f1gate"))):
                self._dw_bwd(dgs[j], conv, k.G3, None, None, None, 0.0, 0, 1, dG, j > 0, gr, tag + nm, B, T2, F2, False)
        # attention, dual paths (each updates dG in place to the gradient w.r.t. its input)
        self._attn_bwd(dG, bw["attn"], k, B, T2, gr, tag + "attn")
        self._dual_path_bwd(dG, bw["dp1"], k.dp[1], B, T2, 3, gr, tag + "dp1")
        self._dual_path_bwd(dG, bw["dp0"], k.dp[0], B, T2, 4, gr, tag + "dp0")
        # (the chain is bandwidth-bound from here to the next block's attention adjoint.  Same-box sweep of the issue point, ms per step: here 88.1-88.3, behind the
        # D0 adjoints 88.2, half here / half there 88.2, at the next block's start 89.4, half here / half at the next block's start 90.1; not parked at all 88.9)
        self._wg_flush()
        # pooled = avgpool(D0n) + D1n; downsample[1] (stride 2, input D0n) and downsample[0] (stride 1, input P = prelu(n0(y0)))
        self._call("rtfs_axpy", dG, 1.0, dN_D1, B * lo * H)
        dD1, dD0 = low(), full()
        self._gln_bwd(dN_D1, k.D1, st[2], d1g, d1be, dD1, False, gr, tag + "d1", B, lo)
        if self.model._hip.fuse["d0tail"]:
            # the two remaining contributions to d(D0n) and the reduce pass of D0's gLN adjoint in one pass over dN_D0
            self._dw_bwd(dD1, bw["d1"], k.D0, st[1], d0g, d0be, 0.0, 1, 2, None, False, gr, tag + "d1", B, T, F_BINS, True)  # (tap / bias gradients only)
            red = gr["_pool"].take(B * lib.STAT_STRIDE, torch.float64).view(B, lib.STAT_STRIDE)
            self._call("rtfs_d0_tail_bwd", dD1, d1w, dG, dN_D0, k.D0, st[1], d0g, d0be, red, _acc(gr, tag + "d0.g", H, dev), _acc(gr, tag + "d0.b", H, dev), B, T, T2)
            if dwadj:  # D0's gLN apply pass, downsample_layers[0]'s tap / bias gradients and its input gradient in one launch
                dD0 = None
                dP = full()
                self._dw_adjoint([(dN_D0, bw["d0"], tag + "d0", (k.D0, st[1], red))], k.y0, st[0], bw["pg"], bw["pbe"], bw["pslope"], 2, dP, False, gr, B, T, F_BINS, bias=True)
            else:
                self._call("rtfs_gln_bwd_apply", dN_D0, k.D0, st[1], d0g, d0be, 0, 0.0, red, dD0, 0, B, TF, H)
        else:
            self._call("rtfs_pool_bwd", dG, dN_D0, B, T, T2)
            self._dw_bwd(dD1, bw["d1"], k.D0, st[1], d0g, d0be, 0.0, 1, 2, dN_D0, True, gr, tag + "d1", B, T, F_BINS, True)
            self._gln_bwd(dN_D0, k.D0, st[1], d0g, d0be, dD0, False, gr, tag + "d0", B, TF)
        if dD0 is not None:
            dP = full()
            self._dw_bwd(dD0, bw["d0"], k.y0, st[0], bw["pg"], bw["pbe"], bw["pslope"], 2, 1, dP, False, gr, tag + "d0", B, T, F_BINS, True)
        # projection: PReLU + gLN adjoint, then the 1x1 conv
        dy0 = full()
        self._gln_bwd(dP, k.y0, st[0], bw["pg"], bw["pbe"], dy0, False, gr, tag + "p", B, TF, H, 1, bw["pslope"], g("pslope", 1))
        self._wg("rtfs_wgrad", dy0, H, k.s_in, C, g("pw", H * C), C, g("pb", H), B * TF, 0, 0, 0, 1, H, C, 1, bw["gw"], bw["gb"], bw["gslope"], None, 0)
        # d(gateway out) = dx (residual path) + dy0 . Wp, formed inside the gateway adjoint
        if a0_mode >= 3:
            self._call("rtfs_proj_gateway_bwd", dy0, bw["pwT"], dx, k.s_in, bw["gw"], bw["gb"], bw["gslope"], da0, 1 if a0_mode == 3 else 0, None, 0,
                     g("gw", C), g("gb", C), g("gslope", 1), B * TF)
            return da0
        ds = torch.empty(B * TF * C, device=dev)
        if next_rwT is not None and a0_mode == 0 and not self.prec:
            # ... and the NEXT block's residual-conv input gradient dE = ds . Wr^T from the ds rows while they are in LDS (round 6): that block does not read ds for it
            dE_next = full()
            lib.call("rtfs_proj_gateway_bwd_next", dy0, bw["pwT"], dx, k.s_in, bw["gw"], bw["gb"], bw["gslope"], ds, g("gw", C), g("gb", C), g("gslope", 1), next_rwT,
                     dE_next, B * TF)
            return ds, dE_next
        self._call("rtfs_proj_gateway_bwd", dy0, bw["pwT"], dx, k.s_in, bw["gw"], bw["gb"], bw["gslope"], ds, 0, da0, a0_mode, g("gw", C), g("gb", C),
                 g("gslope", 1), B * TF)
        return (ds, None) if next_rwT is not None else ds

    @staticmethod
    def _tag(n_blocks, i):
        """prefix of block i's gradient accumulators: one shared block -> "blk.", a stack of R blocks with their own weights (tdanet.py:170-181) -> "blk<i>." """
        return "blk." if n_blocks == 1 else f"blk{i}."

    def backward(self, c, dout):
        """dout [B,1,L] -> (datt, drsz, grads dict in kernel layout)."""
        dx0, da0, da_emb, datt, drsz = self.backward_b(c, dout)
        return datt, drsz, self.backward_a(c, dx0, da0, da_emb)

    # The public adjoints open and close their own reducer stage (deferred finishes on both scratch lanes, side-stream join): on return every
    # parameter gradient they produced is complete in stream order, whoever the caller is (autograd nodes, tools, tests).  Stages nest: an outer
    # stage (`with trainer.stage(dev)`) keeps the section open across both calls.
    def stage(self, dev):
        return _Stage(self, dev)

    @staticmethod
    def _same_weights(c, stage):
        """GatherTrainWeights.refresh() overwrites the persistent kernel-layout buffers that c.pw aliases: forward(A) -> parameter change + another forward
        (an optimizer step on earlier gradients, an EMA swap) -> backward(A) would run A's adjoint on the NEW weights, silently.  Refuse instead."""
        if c.pw.version != c.pw_version:
            raise RuntimeError(f"rtfs_net_amd: the parameters changed and a newer forward re-laid out the kernel weights between this step's forward and its "
                               f"{stage}; run each step's backward before the next forward after a parameter update (or set RTFS_DISABLE=wgather: "
                               "per-step weight snapshots)")

    def backward_b(self, c, dout):
        with self.stage(dout.device):
            return self._backward_b(c, dout)

    def backward_a(self, c, dx0, da0, da_emb):
        with self.stage(dx0.device):
            return self._backward_a(c, dx0, da0, da_emb)

    def _backward_b(self, c, dout):
        """adjoint of forward_b: dout [B,1,L] -> (d x0, d a0 or None, d a_emb, datt, drsz); parameter gradients go to c.gr."""
        m = self.model
        self._same_weights(c, "backward")
        pw = c.pw
        w = pw.w
        B, L, T, T2, R, Tv = c.B, c.L, c.T, c.T2, c.R, c.Tv
        TF = T * F_BINS
        dev = dout.device
        gr = c.gr = {}
        g = lambda name, n: _acc(gr, name, n, dev)  # noqa: E731
        dout = dout.reshape(B, L).to(torch.float32).contiguous()
        # iSTFT + decoder taps
        dspec = torch.empty(B * TF * 2, device=dev)
        dtaps = torch.empty(B * TF * 32, device=dev)
        self._call("rtfs_istft_bwd", dout, dspec, dtaps, B, L)
        self._wg("rtfs_wgrad", dtaps, 32, c.masked, C, g("dec_w", 32 * C), C, None, B * TF, 0, 0, 0, 1, 32, C, 0, None, None, 0.0, None, 0)
        # decoder input gradient + S3 mask's element-wise adjoint (da_emb: WRITTEN here, the first contribution to d(a_emb))
        da_emb = torch.empty(B * TF * C, device=dev)
        dz = torch.empty(B * TF * C, device=dev)
        if m._hip.fuse["decmask"]:
            self._call("rtfs_decoder_mask_bwd", dtaps, w["dec_wT"], c.a_emb, c.m, dz, da_emb, B * TF)  # d(masked) never stored
        else:
            dmasked = torch.empty(B * TF * C, device=dev)
            self._call("rtfs_gemm_rows", dtaps, w["dec_wT"], None, dmasked, B * TF, 32, C, 0)
            self._call("rtfs_mask_bwd_elem", dmasked, c.a_emb, c.m, dz, da_emb, B * TF)
            del dmasked
        # (parked: the input-gradient GEMM below is MFMA-bound as well; issued behind it, this one runs under the last block's bandwidth-bound adjoints)
        self._wg_later("rtfs_wgrad", dz, C, c.refined, C, g("mask_w", C * C), C, g("mask_b", C), B * TF, 0, 0, 0, 1, C, C, 2, None, None, w["mask_slope"], None, 0)
        dx = torch.empty(B * TF * C, device=dev)  # gradient w.r.t. the refined features
        if m._hip.fuse["actepi"] and not self.prec:  # the PReLU's adjoint in the input-gradient GEMM's epilogue (round 6; fp32 contraction only)
            lib.call("rtfs_gemm_prelu_bwd", dz, w["mask_wT"], c.refined, w["mask_slope"], dx, g("mask_slope", 1), B, TF)
        else:
            dpre = torch.empty(B * TF * C, device=dev)
            self._call("rtfs_gemm_rows", dz, w["mask_wT"], None, dpre, B * TF, C, C, 0)
            self._call("rtfs_prelu_bwd", dpre, c.refined, w["mask_slope"], dx, 0, g("mask_slope", 1), B * TF * C)
            del dpre
        self._wg_flush()
        # RTFS blocks R-1 .. 1, CAF, block 0
        da0 = torch.empty(B * TF * C, device=dev)  # running sum of the gradients of every block input (each is `... + a0`)
        blocks = pw.blocks
        bw = lambda i: blocks[0] if len(blocks) == 1 else blocks[i]  # noqa: E731
        # d(a0) = the sum of the input gradients of blocks R-1 .. 1 (each input was `previous output + a0`).  Until round 6 every block's last kernel kept a running
        # sum (a 2.1 GB read-modify-write per block at 32 utterances); every one of those gradients is the next block's dx and alive anyway, so they are summed
        # ONCE, by one n-ary launch on the weight-gradient side stream - its result is not read before block 0's adjoint in the other backward stage
        sum_once = m._hip.fuse["da0sum"] and 2 <= R - 1 <= 8
        parts = []
        dE = None
        for i in range(R - 1, 0, -1):
            # (block i's last kernel also forms the residual-conv input gradient of block i - 1, whose output gradient its ds is: not for block 1 - the CAF adjoint sits
            # between it and block 0 - and only in the plain form of that kernel, i.e. with d(a0) summed once)
            nxt = bw(i - 1)["rwT"] if (m._hip.fuse["nextde"] and sum_once and i >= 2) else None
            res = self._block_bwd(dx, c.blk[i], bw(i), B, T, T2, gr, None if sum_once else da0, 0 if sum_once else (1 if i == R - 1 else 2),
                                  tag=self._tag(len(blocks), i), dE_in=dE, next_rwT=nxt)
            dx, dE = res if nxt is not None else (res, None)
            parts.append(dx)
        if sum_once:
            side = self._side(dev)
            if side is None:
                lib.call("rtfs_sum_n", parts, len(parts), da0, B * TF * C)
            else:
                side.wait_stream(torch.cuda.current_stream(dev))
                for t in parts + [da0]:
                    t.record_stream(side)
                with torch.cuda.stream(side):
                    lib.call("rtfs_sum_n", parts, len(parts), da0, B * TF * C)
        del parts
        # CAF: out = key*rsz^ + att^*val (+ a0)
        datt, drsz = _zeros(B * Tv * C, dev), _zeros(B * Tv * C, dev)
        Rr = _zeros(4 * C, dev)
        cf = c.caf
        self._call("rtfs_caf_bwd_reduce", dx, c.x0, cf["ks"], cf["kb"], cf["vs"], cf["vb"], c.att, c.rsz, datt, drsz, Rr, B, T, Tv)
        coef = self._caf_bwd_coeffs(cf, w, Rr.view(4, C), gr, m)
        dx0 = torch.empty(B * TF * C, device=dev)
        self._call("rtfs_caf_bwd_apply", dx, c.x0, cf["ks"], cf["kb"], c.att, c.rsz, coef, dx0, 0, B, T, Tv)
        return dx0, (da0 if R > 1 else None), da_emb, datt.view(B, Tv, C), drsz.view(B, Tv, C)

    def _backward_a(self, c, dx0, da0, da_emb):
        """adjoint of forward_a.  dx0: gradient of block 0's output (overwritten); da0: running d(a0) sum of the later blocks (updated in
        place) or None; da_emb: gradient that reached a_emb through the S3 mask (updated in place).  -> grads dict in kernel layout."""
        self._same_weights(c, "backward (audio stage A)")
        pw = c.pw
        w = pw.w
        B, T, T2, R = c.B, c.T, c.T2, c.R
        TF = T * F_BINS
        dev = dx0.device
        gr = c.gr
        g = lambda name, n: _acc(gr, name, n, dev)  # noqa: E731
        blocks = pw.blocks
        if da0 is None:
            da0 = torch.empty(B * TF * C, device=dev)
        self._block_bwd(dx0, c.blk[0], blocks[0], B, T, T2, gr, da0, 3 if R > 1 else 4, tag=self._tag(len(blocks), 0))  # block 0's input is a0 itself
        # bottleneck: a0 = Wb . relu(gLN(a_emb)) + bb
        self._wg("rtfs_wgrad", da0, C, c.a_emb, C, g("bn_w", C * C), C, g("bn_bias", C), B * TF, 0, 0, 0, 1, C, C, 3, w["bn_g"], w["bn_b"], 0.0, c.stats[0], TF)
        dR = torch.empty(B * TF * C, device=dev)
        if self.model._hip.fuse["actepi"] and not self.prec:  # the reduce pass of relu(gLN(a_emb))'s adjoint in the GEMM's epilogue (round 6)
            red = gr["_pool"].take(B * lib.STAT_STRIDE, torch.float64).view(B, lib.STAT_STRIDE)
            lib.call("rtfs_gemm_gln_relu_bwd_reduce", da0, w["bn_wT"], c.a_emb, c.stats[0], w["bn_g"], w["bn_b"], dR, red, _acc(gr, "bn.g", C, dev), _acc(gr, "bn.b", C, dev),
                     B, TF)
            self._call("rtfs_gln_bwd_apply", dR, c.a_emb, c.stats[0], w["bn_g"], w["bn_b"], 2, 0.0, red, da_emb, 1, B, TF, C)
        else:
            self._call("rtfs_gemm_rows", da0, w["bn_wT"], None, dR, B * TF, C, C, 0)
            self._gln_bwd(dR, c.a_emb, c.stats[0], w["bn_g"], w["bn_b"], da_emb, True, gr, "bn", B, TF, C, 2)
        # encoder conv weight
        patches = torch.empty(B * TF * 32, device=dev)
        self._call("rtfs_spec_patches", c.spec, patches, B, T)
        # (the step's LAST weight gradient, in line: on the side stream it queued behind the bottleneck's - 1.5 ms, issued when da0 was final - and the stage end
        # waited ~0.4 ms for it with an idle chain; in line it runs while the side stream finishes)
        (self._call if self.model._hip.fuse["enctail"] else self._wg)("rtfs_wgrad", da_emb, C, patches, 32, g("enc", C * 32), 32, None, B * TF, 0, 0, 0, 1, C, 32, 0, None, None, 0.0, None, 0)
        return gr

    def _caf_bwd_coeffs(self, cf, w, Rr, gr, m):
        """BatchNorm adjoint of the CAF key/value embeddings -> per-channel coefficients for rtfs_caf_bwd_apply + parameter grads."""
        dev = Rr.device
        if cf.get("fused"):  # csrc/optim.hip caf_bn_adjoint_kernel: the arithmetic below (caf_bn_adjoint) as one launch
            Rg = Rr
            if cf.get("sync", False):
                Rg = Rr.clone()
                torch.distributed.all_reduce(Rg)
            coef, gd = torch.empty(6, C, device=dev), torch.empty(6, C, device=dev)
            lib.call("rtfs_caf_bn_adjoint", Rr, Rg, cf["n"], cf["mean_x"], cf["lsum"], cf["lsq"], w["caf_key_dw"], w["caf_key_g"], cf["key_inv"], gd[0], gd[1], gd[2],
                     w["caf_value_dw"], w["caf_value_g"], cf["value_inv"], gd[3], gd[4], gd[5], coef)
            for j, tag in enumerate(("key", "value")):
                gr[f"caf_{tag}_dw"], gr[f"caf_{tag}_g"], gr[f"caf_{tag}_be"] = gd[3 * j], gd[3 * j + 1], gd[3 * j + 2]
            return coef
        coef = torch.zeros(6, C, device=dev)
        for j, tag in enumerate(("key", "value")):
            dw, gm = w[f"caf_{tag}_dw"], w[f"caf_{tag}_g"]
            inv = cf[tag + "_inv"]
            A, Bx = Rr[2 * j], Rr[2 * j + 1]
            if cf["training"]:
                c1, c2, c3, Q, A, gdw = caf_bn_adjoint(A, Bx, cf["lsum"], cf["lsq"], cf["n"], cf["mean_x"], cf["var_x"], dw, gm, inv, cf.get("sync", False))
                gr[f"caf_{tag}_dw"] = gdw
            else:
                mean_u = cf[tag + "_mean_u"]
                Q = inv * (dw * Bx - mean_u * A)
                c1, c2, c3 = dw * gm * inv, torch.zeros_like(A), torch.zeros_like(A)
                gr[f"caf_{tag}_dw"] = gm * inv * Bx
            gr[f"caf_{tag}_g"], gr[f"caf_{tag}_be"] = Q, A.clone()
            coef[3 * j], coef[3 * j + 1], coef[3 * j + 2] = c1, c2, c3
        return coef.contiguous()


def caf_bn_batch_stats(sums, n_local, sync):
    """sums: fp64 [2, C] = (sum x, sum x^2) over this rank's n_local positions.  -> (mean, biased var, n, local sum x, local sum (x-mean)x)
    in fp32; with sync (nn.SyncBatchNorm, train.py:145 sync_batchnorm=True) the statistics are those of the union of all ranks."""
    local, n = sums, float(n_local)
    if sync:
        buf = torch.cat([sums.reshape(-1), sums.new_full((1,), n)])  # (new_full: a fill launch, not a blocking host-to-device copy of a Python list)
        torch.distributed.all_reduce(buf)
        sums, n = buf[:-1].view(2, -1), buf[-1]  # (n stays a 0-d device tensor: .item() here is a host-device synchronisation in the middle of every DDP step)
    mean = sums[0] / n
    var = (sums[1] / n - mean * mean).clamp_min(0)
    lcov = local[1] - mean * local[0]  # sum_local (x - mean) x, differenced in fp64
    return mean.float(), var.float(), n, local[0].float(), lcov.float()


def caf_bn_adjoint(A, Bx, lsum, lcov, n, mean_x, var_x, dw, gm, inv, sync):
    """Adjoint of u = dw*x -> BatchNorm(batch statistics) -> y = gm*uhat + be, given this rank's per-channel reductions
    A = sum dy, Bx = sum dy*x.  Input gradient dx = dy*c1 + c2 + c3*x (coefficients use the statistics of ALL ranks when sync);
    parameter gradients are this rank's share (DDP averages them afterwards, as with nn.SyncBatchNorm).
    -> (c1, c2, c3, dgamma, dbeta, d(dw))"""
    A_g, Bx_g = A, Bx
    if sync:
        AB = torch.stack([A, Bx])
        torch.distributed.all_reduce(AB)
        A_g, Bx_g = AB[0], AB[1]
    Q_g = dw * inv * (Bx_g - mean_x * A_g)  # sum dy * uhat over all ranks
    Q_l = dw * inv * (Bx - mean_x * A)
    c1 = dw * gm * inv
    c3 = -c1 * (Q_g / n) * (dw * inv)
    c2 = -c1 * A_g / n - c3 * mean_x
    # d(dw) = sum_local du*x, du = gm*inv*(dy - A_g/n - uhat*Q_g/n)
    gdw = gm * inv * (Bx - (A_g / n) * lsum - (Q_g / n) * dw * inv * lcov)
    return c1, c2, c3, Q_l, A.clone(), gdw


# ============================================ reference-layout mapping ============================================
def grads_to_reference(model, pw: TrainWeights, gr: dict) -> dict:
    """kernel-layout gradient buffers -> {reference parameter name: gradient tensor of the parameter's shape}"""
    out = {}
    z = lambda name, n: gr.get(name)  # noqa: E731

    def put(name, t, shape):
        if t is not None:
            out[name] = t.reshape(shape).contiguous()

    put("encoder.conv.full_layer.2.weight", gr["enc"].view(C, 32)[:, :18], (C, 2, 3, 3))
    put("audio_bottleneck.full_layer.0.norm.weight", gr["bn.g"], (C,))
    put("audio_bottleneck.full_layer.0.norm.bias", gr["bn.b"], (C,))
    put("audio_bottleneck.full_layer.2.weight", gr["bn_w"], (C, C, 1, 1))
    put("audio_bottleneck.full_layer.2.bias", gr["bn_bias"], (C,))
    n_blocks = len(pw.blocks)
    shared = model.refinement_module.audio_net.shared
    for bi in range(n_blocks):  # one shared block, or R blocks with their own parameters (state-dict prefix blocks.<i>.)
        p = "refinement_module.audio_net.blocks." + ("" if shared else f"{bi}.")
        b = HipTrainer._tag(n_blocks, bi)
        put(p + "gateway.full_layer.2.weight", gr[b + "gw"], (C, 1, 1, 1))
        put(p + "gateway.full_layer.2.bias", gr[b + "gb"], (C,))
        put(p + "gateway.full_layer.4.weight", gr[b + "gslope"], (1,))
        put(p + "projection.full_layer.2.weight", gr[b + "pw"], (H, C, 1, 1))
        put(p + "projection.full_layer.2.bias", gr[b + "pb"], (H,))
        put(p + "projection.full_layer.3.norm.weight", gr[b + "p.g"], (H,))
        put(p + "projection.full_layer.3.norm.bias", gr[b + "p.b"], (H,))
        put(p + "projection.full_layer.4.weight", gr[b + "pslope"], (1,))

        def dw(prefix, key, bias):
            put(prefix + "full_layer.2.weight", gr[key + ".w"].view(16, 64).t(), (64, 1, 4, 4))
            if bias:
                put(prefix + "full_layer.2.bias", gr[key + ".bias"], (64,))
            put(prefix + "full_layer.3.norm.weight", gr[key + ".g"], (64,))
            put(prefix + "full_layer.3.norm.bias", gr[key + ".b"], (64,))

        dw(p + "downsample_layers.0.", b + "d0", True)
        dw(p + "downsample_layers.1.", b + "d1", True)
        for name, key in (("fusion_layers.0.local_embedding", "f0l"), ("fusion_layers.0.global_embedding", "f0g"), ("fusion_layers.0.global_gate", "f0gate"),
                          ("fusion_layers.1.local_embedding", "f1l"), ("fusion_layers.1.global_embedding", "f1g"), ("fusion_layers.1.global_gate", "f1gate"),
                          ("concat_layers.0.local_embedding", "cl"), ("concat_layers.0.global_embedding", "cg"), ("concat_layers.0.global_gate", "cgate")):
            dw(f"{p}{name}.", b + key, False)
        put(p + "residual_conv.full_layer.2.weight", gr[b + "rw"], (C, H, 1, 1))
        put(p + "residual_conv.full_layer.2.bias", gr[b + "rb"], (C,))
        for j in (0, 1):
            q, k = f"{p}globalatt.{j}.", f"{b}dp{j}."
            put(q + "norm.gamma", gr[k + "g"], (1, 64, 1, 1))
            put(q + "norm.beta", gr[k + "b"], (1, 64, 1, 1))
            put(q + "rnn.rnn_lst.0.weight", gr[k + "w0"].view(256, 8, 64).permute(2, 1, 0), (512, 256))
            for l in range(4):
                put(q + f"rnn.rnn_lst.{l}.weight_c", gr[k + f"l{l}.wc"], (128,))
                put(q + f"rnn.rnn_lst.{l}.bias", gr[k + f"l{l}.bias"], (128,))
                if l > 0:
                    put(q + f"rnn.rnn_lst.{l}.weight", gr[k + f"l{l}.w"].view(3, 64, 64).permute(2, 1, 0), (64, 192))
            put(q + "linear.weight", gr[k + "ct_w"].view(64, 8, 64).permute(2, 0, 1).flip(2), (64, 64, 8))
            put(q + "linear.bias", gr[k + "ct_b"], (64,))
        q, k = p + "globalatt.2.", b + "attn."
        wq = gr[k + "w"].view(96, 64)
        bq = gr[k + "bias"]
        sl = gr[k + "slope"]
        off = 0
        for mi, (name, nch, gk, bk) in enumerate((("Queries", 4, "gq", "bq"), ("Keys", 4, "gk", "bk"), ("Values", 16, "gv", "bv"))):
            for h in range(4):
                put(f"{q}{name}.{h}.conv.weight", wq[off:off + nch], (nch, 64, 1, 1))
                put(f"{q}{name}.{h}.conv.bias", bq[off:off + nch], (nch,))
                put(f"{q}{name}.{h}.act.weight", sl[mi * 4 + h:mi * 4 + h + 1], (1,))
                put(f"{q}{name}.{h}.norm.gamma", gr[k + gk].view(4, -1)[h], (1, nch, 1, 64))
                put(f"{q}{name}.{h}.norm.beta", gr[k + bk].view(4, -1)[h], (1, nch, 1, 64))
                off += nch
        put(q + "attn_concat_proj.conv.weight", gr[k + "ow"], (64, 64, 1, 1))
        put(q + "attn_concat_proj.conv.bias", gr[k + "ob"], (64,))
        put(q + "attn_concat_proj.act.weight", gr[k + "oslope"], (1,))
        put(q + "attn_concat_proj.norm.gamma", gr[k + "og"].view(64, 64).t(), (1, 64, 1, 64))
        put(q + "attn_concat_proj.norm.beta", gr[k + "obe"].view(64, 64).t(), (1, 64, 1, 64))
    caf = pw.caf_prefix
    for tag in ("key", "value"):
        put(f"{caf}{tag}_embed.full_layer.2.weight", gr[f"caf_{tag}_dw"], (C, 1, 1, 1))
        put(f"{caf}{tag}_embed.full_layer.3.weight", gr[f"caf_{tag}_g"], (C,))
        put(f"{caf}{tag}_embed.full_layer.3.bias", gr[f"caf_{tag}_be"], (C,))
    put("mask_generator.mask_generator.0.weight", gr["mask_slope"], (1,))
    put("mask_generator.mask_generator.1.full_layer.2.weight", gr["mask_w"], (C, C, 1, 1))
    put("mask_generator.mask_generator.1.full_layer.2.bias", gr["mask_b"], (C,))
    put("decoder.decoder.weight", gr["dec_w"].view(32, C)[:18].t(), (C, 2, 3, 3))
    return out


class AVNetHipFunction(torch.autograd.Function):
    """out = AVNet audio branch (wav, att, rsz; audio parameters) as ONE node (kept for tools / reports)."""

    @staticmethod
    def forward(ctx, trainer, names, wav, att, rsz, *params):
        with torch.no_grad():
            out, saved = trainer.forward(wav, att, rsz)
        ctx.trainer, ctx.names, ctx.saved = trainer, names, saved
        return out

    @staticmethod
    def backward(ctx, dout):
        trainer = ctx.trainer
        with torch.no_grad():
            with trainer.stage(dout.device):
                datt, drsz, gr = trainer.backward(ctx.saved, dout)
            ref = grads_to_reference(trainer.model, ctx.saved.pw, gr)
        grads = tuple(ref.get(n) for n in ctx.names)
        ctx.saved = None
        return (None, None, None, datt, drsz) + grads


class AVNetHipStageA(torch.autograd.Function):
    """(wav; audio parameters) -> (x0, a0, a_emb): everything before the CAF cell.  Its backward runs AFTER AVNetHipStageB's and returns
    the gradients of ALL audio parameters (stage B leaves its share in the shared step context), so that autograd can run the video
    glue's backward - which only needs stage B's datt / drsz - on its own stream underneath this node's kernels."""

    @staticmethod
    def forward(ctx, trainer, names, step, wav, *params):
        with torch.no_grad():
            c = trainer.forward_a(wav)
        step.c = c
        ctx.trainer, ctx.names, ctx.step = trainer, names, step
        n = c.B * c.T * F_BINS
        # fresh view objects: autograd attaches this node to the RETURNED tensors; the context keeps the plain buffers (no reference cycle)
        return c.x0.view(n, C), c.a0.view(n, C), c.a_emb.view(n, C)

    @staticmethod
    def backward(ctx, dx0, da0, da_emb):
        trainer, c = ctx.trainer, ctx.step.c
        with torch.no_grad():
            n = c.B * c.T * F_BINS * C
            fix = lambda t: None if t is None else t.contiguous().view(n)  # noqa: E731
            dx0, da0, da_emb = fix(dx0), fix(da0), fix(da_emb)
            if dx0 is None:
                dx0 = torch.zeros(n, device=c.x0.device)
            if da_emb is None:
                da_emb = torch.zeros(n, device=c.x0.device)
            gr = trainer.backward_a(c, dx0, da0, da_emb)  # (own reducer stage: flushed + joined, every parameter gradient complete in stream order)
            ref = grads_to_reference(trainer.model, c.pw, gr)
        grads = tuple(ref.get(name) for name in ctx.names)
        c.__dict__.clear()
        ctx.step.c = None
        return (None, None, None, None) + grads


class AVNetHipStageB(torch.autograd.Function):
    """(x0, a0, a_emb, att, rsz) -> out: CAF cell, blocks 1..R-1, S3 mask, decoder, iSTFT."""

    @staticmethod
    def forward(ctx, trainer, step, x0, a0, a_emb, att, rsz):
        with torch.no_grad():
            out = trainer.forward_b(step.c, att, rsz)
        ctx.trainer, ctx.step = trainer, step
        return out

    @staticmethod
    def backward(ctx, dout):
        with torch.no_grad():
            dx0, da0, da_emb, datt, drsz = ctx.trainer.backward_b(ctx.step.c, dout)  # (own reducer stage, csrc/spread.hip)
        n = dx0.numel() // C
        return None, None, dx0.view(n, C), (None if da0 is None else da0.view(n, C)), da_emb.view(n, C), datt, drsz


class StepCtx:
    """holder shared by the two stages of one training step"""
    c = None
