"""Training step of the VP (video) block on HIP kernels (csrc/vp_train.hip).

The reference runs the 1-D `TDANetBlock` (separators/tdanet.py:106-133, config yaml:74-92) as ~100 small PyTorch ops forward and ~300
backward, with `BatchNorm1d` batch statistics (SyncBatchNorm under DDP, train.py:145) at 26 places.  Here the convolution / BatchNorm
chain is two `torch.autograd.Function`s around the 13-token `GlobalAttention` (LayerNorm + nn.MultiheadAttention + FFN, dropout / DropPath,
layers/attention.py:28-73,192-220), which is a third one:

    x [B,512,T] --VPStageA--> g [B,64,Tg] --VPAttnFn--> g' --VPStageB--> out [B,512,T]

* stage A: gateway, projection (+BN+PReLU), the four down-sampling convolutions (+BN), pooled sum;
* GlobalAttention: since round 3 one forward and one backward launch of csrc/vp_attn.hip (`VPAttnFn`; the stochastic layers take one draw of
  keep-masks per step from torch.rand - the only ATen kernels left in the block);
* stage B: the 4 + 3 InjectionMultiSum units (21 convolutions + BN), residual conv + gateway residual.

BatchNorm never runs as an op: producers accumulate per-channel (sum, sum of squares) into a slot of one statistics tensor, consumers
normalise on read; under SyncBatchNorm the slots of a dependency level are all-reduced between the two launches (9 small all-reduces
forward, 10 backward).  Running statistics are updated with two `torch._foreach` calls per step.  In `eval()` mode (running statistics,
autograd on) the same kernels run with slots synthesised from the running statistics and the batch-coupling terms of the adjoint off.
"""
from __future__ import annotations

import torch

try:
    from .. import lib
except ImportError:  # relocated copy of the models sub-package (train.py:95 / test.py:33-36)
    from rtfs_net_amd import lib

NS = 26  # statistics slots: 0 projection, 1-4 down-sampling, 5 + 3 i + {0 local, 1 embedding, 2 gate} fusion layer i, 17 + 3 j + {0,1,2} concat layer j
EPS = 1e-5


def supported(vb) -> bool:
    """the RTFS-Net family's video block: 512 -> 64, k = 3, stride 2, depth 4, BatchNorm1d / SyncBatchNorm, one GlobalAttention"""
    bn = vb.projection.full_layer[3]
    return (vb.in_chan == 512 and vb.hid_chan == 64 and vb.kernel_size == 3 and vb.stride == 2 and vb.upsampling_depth == 4 and not vb.is2d
            and isinstance(bn, torch.nn.modules.batchnorm._BatchNorm) and abs(bn.eps - EPS) < 1e-12 and len(vb.globalatt) == 1)


class _Step:
    """everything the two stages of one step share (saved tensors, statistics, gradient buffers)"""


def _bn_modules(vb):
    mods = [vb.projection.full_layer[3]] + [vb.downsample_layers[i].full_layer[3] for i in range(4)]
    for unit in list(vb.fusion_layers) + list(vb.concat_layers):
        mods += [unit.local_embedding.full_layer[3], unit.global_embedding.full_layer[3], unit.global_gate.full_layer[3]]
    return mods


def _lengths(T):
    Ts = [T]
    for _ in range(3):
        Ts.append((Ts[-1] - 1) // 2 + 1)
    return Ts


class BatchCountProbe:
    """SyncBatchNorm's utterance count, checked without a stall: SUM all-reduce of this rank's batch size (issued by EVERY rank on EVERY step),
    asynchronous copy to pinned host memory, event; `verify()` waits for the event only and raises ValueError on unequal per-rank batches."""

    def __init__(self, B, dev):
        self.B, self.world = int(B), torch.distributed.get_world_size()
        count = torch.full((1,), float(B), device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(count)
        self.event = None
        if count.is_cuda:
            self.host = torch.empty(1, dtype=torch.float64, pin_memory=True)
            self.host.copy_(count, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record()
        else:
            self.host = count

    def verify(self):
        if self.event is not None:
            self.event.synchronize()
        total = float(self.host[0])
        if total != float(self.B * self.world):
            raise ValueError(f"SyncBatchNorm in the VP block's HIP training step needs equal per-rank batch sizes: this rank holds {self.B} utterances, "
                             f"the {self.world} ranks together {total:g}: use drop_last / DistributedSampler padding, or RTFS_DISABLE=vp_hip for the PyTorch modules")


class VPTrainer:
    def __init__(self, vb):
        self.vb = vb
        self.bns = _bn_modules(vb)
        self._ncache = {}
        self._pending = None  # SyncBatchNorm: the step's all-reduced utterance count, not yet compared with B x world (BatchCountProbe)

    # ---- helpers ----------------------------------------------------------------------------------------------------------
    def _sync(self):
        bn = self.bns[0]
        return (isinstance(bn, torch.nn.SyncBatchNorm) and self.vb.training and torch.distributed.is_available() and torch.distributed.is_initialized()
                and torch.distributed.get_world_size() > 1)

    def check_equal_batches(self):
        probe, self._pending = self._pending, None
        if probe is not None:
            probe.verify()

    def _allreduce(self, st, t):
        if st.sync:
            torch.distributed.all_reduce(t)

    def _prepare(self, x, slopes=None):
        vb = self.vb
        st = _Step()
        st.B, st.T = x.shape[0], x.shape[-1]
        st.Ts = _lengths(st.T)
        st.Tg = st.Ts[3]
        st.train = vb.training
        st.sync = self._sync()
        dev = x.device
        # utterances behind the statistics: all ranks' under SyncBatchNorm.  Equal per-rank batches (what DistributedSampler delivers) make that
        # B x world without a host read-back in front of the first kernel.  The assumption is CHECKED every step: every rank issues the same
        # one-element SUM all-reduce of its own B (the decision to communicate never depends on rank-local state - a per-rank cache of verified
        # batch sizes would pair this collective with another rank's statistics all-reduce on an unequal last batch), the result travels to
        # pinned host memory behind an event, and `check_equal_batches` reads it at the start of the step's backward (long complete by then: no
        # stall) and raises before any gradient of the skewed statistics can reach the optimizer (torch's SyncBatchNorm all-reduces counts)
        st.Bn = float(st.B * (torch.distributed.get_world_size() if st.sync else 1))
        self.check_equal_batches()  # a forward-only step that was never followed by a backward
        if st.sync:
            self._pending = BatchCountProbe(st.B, dev)
        st.stats = torch.zeros(NS, 2, 64, device=dev, dtype=torch.float64)  # float64 slots (order-independent sums; E[x^2] - mean^2 differenced in float64)
        # the two scalar PReLU slopes are kernel arguments: taken from the caller (AVNet reads every scalar of the model in ONE transfer per
        # optimizer step, hip_path.PreparedWeights) - fetching them here would be a host synchronisation in the middle of the step
        if slopes is None:
            slopes = torch.cat([vb.gateway.full_layer[4].weight.detach().reshape(1), vb.projection.full_layer[4].weight.detach().reshape(1)]).tolist()
        st.gslope, st.pslope = slopes
        if not st.train:  # running statistics as slots: sum = mean n, sum of squares = (var + mean^2) n with n = 1
            with torch.no_grad():
                rm = torch.stack([b.running_mean.double() for b in self.bns])
                rv = torch.stack([b.running_var.double() for b in self.bns])
                st.stats[:, 0], st.stats[:, 1] = rm, rv + rm * rm
        st.gam = [b.weight.detach().float().contiguous() for b in self.bns]
        st.bet = [b.bias.detach().float().contiguous() for b in self.bns]
        return st

    def _inv_n(self, st, T):
        return 1.0 / (st.Bn * T) if st.train else 1.0

    def _bn(self, st, idx, T):
        """(stats slot, gamma, beta, inv_n) of BatchNorm `idx` over tensors of length T"""
        return st.stats[idx], st.gam[idx], st.bet[idx], self._inv_n(st, T)

    @staticmethod
    def _w3(conv):
        return conv.weight.detach().float().reshape(64, 3).contiguous()

    def _conv(self, st, src, in_bn, in_act, in_slope, conv0, idx0, Tin, Tout, stride, conv1=None, idx1=None):
        """one launch: one or two depth-wise k = 3 convolutions of the (normalised-on-read) tensor src -> raw outputs + statistics"""
        dev = src.device
        o0 = torch.empty(st.B, 64, Tout, device=dev)
        o1 = torch.empty(st.B, 64, Tout, device=dev) if conv1 is not None else None
        ins = in_bn if in_bn is not None else (None, None, None, 1.0)
        b0 = conv0.bias.detach().float().contiguous() if conv0.bias is not None else None
        # in eval mode the output slots already hold the running statistics: the kernel's accumulation goes to a scratch slot
        s0 = st.stats[idx0] if st.train else torch.zeros(2, 64, device=dev, dtype=torch.float64)
        s1 = (st.stats[idx1] if st.train else torch.zeros(2, 64, device=dev, dtype=torch.float64)) if conv1 is not None else None
        lib.call("rtfs_vp_dwconv_fwd", src, ins[0], ins[1], ins[2], ins[3], in_act, in_slope, self._w3(conv0), b0, o0, s0,
                 self._w3(conv1) if conv1 is not None else None, o1, s1, st.B, Tin, Tout, stride)
        return o0, o1

    # ---- forward ----------------------------------------------------------------------------------------------------------
    def forward_a(self, x, slopes=None):
        vb = self.vb
        x = x.detach().float().contiguous()
        st = self._prepare(x, slopes)
        dev, B, T = x.device, st.B, st.T
        st.x = x
        st.r = torch.empty(B, 512, T, device=dev)
        st.y = torch.empty(B, 64, T, device=dev)
        gconv, pconv = vb.gateway.full_layer[2], vb.projection.full_layer[2]
        st.gw, st.gb = gconv.weight.detach().float().reshape(512).contiguous(), gconv.bias.detach().float().contiguous()
        st.Wp, st.bp = pconv.weight.detach().float().reshape(64, 512).contiguous(), pconv.bias.detach().float().contiguous()
        s0 = st.stats[0] if st.train else torch.zeros(2, 64, device=dev, dtype=torch.float64)
        lib.call("rtfs_vp_gate_proj_fwd", x, st.gw, st.gb, st.gslope, st.Wp, st.bp, st.r, st.y, s0, B, T)
        self._allreduce(st, st.stats[0])
        st.raw = []
        src, in_bn, act, slope, Tin = st.y, self._bn(st, 0, T), 1, st.pslope, T
        for i in range(4):
            To = st.Ts[i]
            o, _ = self._conv(st, src, in_bn, act, slope, vb.downsample_layers[i].full_layer[2], 1 + i, Tin, To, 1 if i == 0 else 2)
            self._allreduce(st, st.stats[1 + i])
            st.raw.append(o)
            src, in_bn, act, slope, Tin = o, self._bn(st, 1 + i, To), 0, 0.0, To
        g = torch.empty(B, 64, st.Tg, device=dev)
        lib.call("rtfs_vp_pool_fwd", st.raw, [st.stats[1 + i] for i in range(4)], [st.gam[1 + i] for i in range(4)], [st.bet[1 + i] for i in range(4)],
                 st.Ts[0], st.Ts[1], st.Ts[2], st.Ts[3], *[self._inv_n(st, st.Ts[i]) for i in range(4)], g, B, st.Tg)
        return g, st

    def _ims_fwd(self, st, unit, base, local, local_bn, Tn, glob, To, res_raw=None, res_bn=None):
        """one InjectionMultiSum (layers/fusion.py:54-69): three convolutions + statistics, then the mix.  Returns (mixed, saved)."""
        loc, _ = self._conv(st, local, local_bn, 0, 0.0, unit.local_embedding.full_layer[2], base, Tn, Tn, 1)
        emb, gate = self._conv(st, glob, None, 0, 0.0, unit.global_embedding.full_layer[2], base + 1, To, To, 1, unit.global_gate.full_layer[2], base + 2)
        return (loc, emb, gate)

    def _mix(self, st, base, convs, Tn, To, res_raw=None, res_idx=None):
        loc, emb, gate = convs
        out = torch.empty(st.B, 64, Tn, device=loc.device)
        lb, eb, gb_ = self._bn(st, base, Tn), self._bn(st, base + 1, To), self._bn(st, base + 2, To)
        rb = self._bn(st, res_idx, Tn) if res_raw is not None else (None, None, None, 1.0)
        lib.call("rtfs_vp_mix_fwd", loc, lb[0], lb[1], lb[2], lb[3], gate, gb_[0], gb_[1], gb_[2], emb, eb[0], eb[1], eb[2], eb[3], res_raw, rb[0], rb[1], rb[2],
                 out, st.B, Tn, To)
        return out

    def forward_b(self, st, g2):
        vb = self.vb
        g2 = g2.detach().float().contiguous()
        st.g2 = g2
        Ts, Tg = st.Ts, st.Tg
        # fusion layers: all 12 convolutions first (one dependency level = one statistics all-reduce), then the four mixes
        st.fconv = [self._ims_fwd(st, vb.fusion_layers[i], 5 + 3 * i, st.raw[i], self._bn(st, 1 + i, Ts[i]), Ts[i], g2, Tg) for i in range(4)]
        self._allreduce(st, st.stats[5:17])
        st.fused = [self._mix(st, 5 + 3 * i, st.fconv[i], Ts[i], Tg) for i in range(4)]
        # concat layers, coarse to fine (tdanet.py:127-129): exp_j = concat_j(fused_j, exp_{j+1}) + ds_j
        st.cconv, st.exp = [None] * 3, [None] * 3
        glob, To = st.fused[3], Ts[3]
        for j in (2, 1, 0):
            st.cconv[j] = self._ims_fwd(st, vb.concat_layers[j], 17 + 3 * j, st.fused[j], None, Ts[j], glob, To)
            self._allreduce(st, st.stats[17 + 3 * j:20 + 3 * j])
            st.exp[j] = self._mix(st, 17 + 3 * j, st.cconv[j], Ts[j], To, st.raw[j], 1 + j)
            glob, To = st.exp[j], Ts[j]
        rconv = vb.residual_conv.full_layer[2]
        st.Wr, st.br = rconv.weight.detach().float().reshape(512, 64).contiguous(), rconv.bias.detach().float().contiguous()
        out = torch.empty(st.B, 512, st.T, device=g2.device)
        lib.call("rtfs_vp_resid_fwd", st.exp[0], st.Wr, st.br, st.r, out, st.B, st.T)
        if st.train:
            self._update_running(st)
        return out

    def _update_running(self, st):
        """BatchNorm running statistics (momentum, unbiased variance) of all 26 layers with a handful of launches"""
        with torch.no_grad():
            key = (st.Bn, st.T, str(st.stats.device))
            n = self._ncache.get(key)
            if n is None:  # (built once per shape: a host -> device copy of a Python list is a blocking transfer)
                n = self._ncache[key] = torch.tensor([st.Bn * T for T in self._slot_lengths(st)], device=st.stats.device, dtype=torch.float64).view(NS, 1)
            mean = st.stats[:, 0] / n
            var = ((st.stats[:, 1] / n - mean * mean).clamp_min(0) * (n / (n - 1).clamp_min(1))).float()
            mean = mean.float()
            mom = self.bns[0].momentum if self.bns[0].momentum is not None else 0.1
            rms, rvs = [b.running_mean for b in self.bns], [b.running_var for b in self.bns]
            torch._foreach_mul_(rms, 1 - mom)
            torch._foreach_add_(rms, list(mean.unbind(0)), alpha=mom)
            torch._foreach_mul_(rvs, 1 - mom)
            torch._foreach_add_(rvs, list(var.unbind(0)), alpha=mom)
            torch._foreach_add_([b.num_batches_tracked for b in self.bns], 1)

    @staticmethod
    def _slot_lengths(st):
        """length of the tensor behind every statistics slot"""
        Ts, Tg = st.Ts, st.Tg
        lens = [Ts[0]] + list(Ts)
        for i in range(4):
            lens += [Ts[i], Tg, Tg]
        for j in range(3):  # concat j: local embedding at Ts[j], global embedding / gate at the coarser Ts[j + 1]
            lens += [Ts[j], Ts[j + 1], Ts[j + 1]]
        assert len(lens) == NS
        return lens

    # ---- backward ---------------------------------------------------------------------------------------------------------
    def _bn_adjoint(self, st, dyhat, raw, idx, T, level):
        """register BatchNorm `idx` for the reduction of its dependency level"""
        level.append((dyhat, raw, idx, T))

    def _reduce_level(self, st, level):
        """(sum dyhat, sum dyhat xhat) of every BatchNorm of a level -> st.sums (local copy kept as dbeta / dgamma), all-reduced under SyncBatchNorm"""
        for dyhat, raw, idx, T in level:
            b = self._bn(st, idx, T)
            lib.call("rtfs_vp_bn_bwd_reduce", dyhat, raw, b[0], b[1], b[2], b[3], st.sums[idx], st.B, T)
        if st.sync:
            idxs = [idx for _, _, idx, _ in level]
            st.local_sums[idxs] = st.sums[idxs]
            buf = st.sums[idxs].contiguous()
            torch.distributed.all_reduce(buf)
            st.sums[idxs] = buf

    def _conv_bwd(self, st, dyhat, raw, idx, Tout, src, in_bn, in_act, in_slope, conv, dsrc, accumulate, Tin, stride, dslope=None):
        b = self._bn(st, idx, Tout)
        ins = in_bn if in_bn is not None else (None, None, None, 1.0)
        dW = st.dW[idx]
        dbias = st.dbias[idx] if conv.bias is not None else None
        lib.call("rtfs_vp_dwconv_bwd", dyhat, raw, b[0], b[1], b[2], b[3], st.sums[idx], 1.0 / (st.Bn * Tout), 1 if st.train else 0, src, ins[0], ins[1], ins[2],
                 ins[3], in_act, in_slope, self._w3(conv), dW, dbias, dsrc, 1 if accumulate else 0, dslope, st.B, Tin, Tout, stride)

    def _ims_bwd(self, st, unit, base, dout, convs, Tn, To, local_src, local_bn, glob_src, d_local, d_local_acc, d_glob, d_glob_acc, dres_acc):
        loc, emb, gate = convs
        dev = dout.device
        dloc, dgate, demb = torch.empty_like(loc), torch.empty_like(gate), torch.empty_like(emb)
        lb, gbn = self._bn(st, base, Tn), self._bn(st, base + 2, To)
        lib.call("rtfs_vp_mix_bwd", dout, loc, lb[0], lb[1], lb[2], lb[3], gate, gbn[0], gbn[1], gbn[2], gbn[3], dloc, dgate, demb, dres_acc, st.B, Tn, To)
        level = [(dloc, loc, base, Tn), (demb, emb, base + 1, To), (dgate, gate, base + 2, To)]
        self._reduce_level(st, level)
        self._conv_bwd(st, dloc, loc, base, Tn, local_src, local_bn, 0, 0.0, unit.local_embedding.full_layer[2], d_local, d_local_acc, Tn, 1)
        self._conv_bwd(st, demb, emb, base + 1, To, glob_src, None, 0, 0.0, unit.global_embedding.full_layer[2], d_glob, d_glob_acc, To, 1)
        self._conv_bwd(st, dgate, gate, base + 2, To, glob_src, None, 0, 0.0, unit.global_gate.full_layer[2], d_glob, True, To, 1)

    def backward_b(self, st, dout):
        vb = self.vb
        dev = dout.device
        dout = dout.detach().float().contiguous()
        st.dout = dout
        B, Ts, Tg = st.B, st.Ts, st.Tg
        st.sums = torch.zeros(NS, 2, 64, device=dev, dtype=torch.float64)
        st.local_sums = st.sums if not st.sync else torch.zeros(NS, 2, 64, device=dev, dtype=torch.float64)
        st.dW = torch.zeros(NS, 3, 64, device=dev)
        st.dbias = torch.zeros(NS, 64, device=dev)
        st.dds = [torch.zeros(B, 64, Ts[i], device=dev) for i in range(4)]  # gradients w.r.t. the BatchNorm outputs of the four down-sampled tensors
        st.dWr, st.dbr = torch.zeros(512, 64, device=dev), torch.zeros(512, device=dev)
        d_exp = torch.empty(B, 64, Ts[0], device=dev)
        lib.call("rtfs_vp_resid_bwd", dout, st.exp[0], st.Wr, d_exp, st.dWr, st.dbr, B, st.T)
        d_fused = [torch.empty(B, 64, Ts[i], device=dev) for i in range(4)]
        for j in (0, 1, 2):  # concat layers, fine to coarse: exp_j = concat_j(fused_j, glob) + ds_j, glob = exp_{j+1} (fused_3 for j = 2)
            glob = st.exp[j + 1] if j < 2 else st.fused[3]
            d_glob = torch.empty(B, 64, Ts[j + 1], device=dev) if j < 2 else d_fused[3]
            self._ims_bwd(st, vb.concat_layers[j], 17 + 3 * j, d_exp, st.cconv[j], Ts[j], Ts[j + 1], st.fused[j], None, glob, d_fused[j], False, d_glob, False,
                          st.dds[j])
            d_exp = d_glob
        dg2 = torch.empty(B, 64, Tg, device=dev)
        for i in range(4):
            self._ims_bwd(st, vb.fusion_layers[i], 5 + 3 * i, d_fused[i], st.fconv[i], Ts[i], Tg, st.raw[i], self._bn(st, 1 + i, Ts[i]), st.g2, st.dds[i], True,
                          dg2, i > 0, None)
        return dg2

    def backward_a(self, st, dg):
        vb = self.vb
        dev = dg.device
        dg = dg.detach().float().contiguous()
        B, T, Ts = st.B, st.T, st.Ts
        lib.call("rtfs_vp_pool_bwd", dg, st.dds, Ts[0], Ts[1], Ts[2], Ts[3], B, st.Tg)
        dyhat_y = torch.empty(B, 64, T, device=dev)
        st.dpslope = torch.zeros(1, device=dev)
        for i in (3, 2, 1, 0):
            self._reduce_level(st, [(st.dds[i], st.raw[i], 1 + i, Ts[i])])
            conv = vb.downsample_layers[i].full_layer[2]
            if i > 0:
                self._conv_bwd(st, st.dds[i], st.raw[i], 1 + i, Ts[i], st.raw[i - 1], self._bn(st, i, Ts[i - 1]), 0, 0.0, conv, st.dds[i - 1], True, Ts[i - 1], 2)
            else:
                self._conv_bwd(st, st.dds[0], st.raw[0], 1, Ts[0], st.y, self._bn(st, 0, T), 1, st.pslope, conv, dyhat_y, False, T, 1, st.dpslope)
        self._reduce_level(st, [(dyhat_y, st.y, 0, T)])
        dx = torch.empty(B, 512, T, device=dev)
        st.dWp, st.dbp = torch.zeros(64, 512, device=dev), torch.zeros(64, device=dev)
        st.dgw, st.dgb, st.dgslope = torch.zeros(512, device=dev), torch.zeros(512, device=dev), torch.zeros(1, device=dev)
        b = self._bn(st, 0, T)
        lib.call("rtfs_vp_gate_proj_bwd", dyhat_y, st.y, b[0], b[1], b[2], b[3], st.sums[0], 1.0 / (st.Bn * T), 1 if st.train else 0, st.dout, st.x, st.r, st.gw,
                 st.gb, st.gslope, st.Wp, st.dWp, st.dbp, st.dgw, st.dgb, st.dgslope, dx, B, T)
        return dx

    # ---- parameter lists / gradient mapping ---------------------------------------------------------------------------------
    def params_a(self):
        vb = self.vb
        ps = [vb.gateway.full_layer[2].weight, vb.gateway.full_layer[2].bias, vb.gateway.full_layer[4].weight, vb.projection.full_layer[2].weight,
              vb.projection.full_layer[2].bias, vb.projection.full_layer[3].weight, vb.projection.full_layer[3].bias, vb.projection.full_layer[4].weight]
        for i in range(4):
            fl = vb.downsample_layers[i].full_layer
            ps += [fl[2].weight, fl[2].bias, fl[3].weight, fl[3].bias]
        return ps

    def grads_a(self, st):
        s = st.local_sums
        gs = [st.dgw.view(512, 1, 1), st.dgb, st.dgslope, st.dWp.view(64, 512, 1), st.dbp, s[0, 1].float(), s[0, 0].float(), st.dpslope]
        for i in range(4):
            gs += [st.dW[1 + i].t().reshape(64, 1, 3).contiguous(), st.dbias[1 + i].clone(), s[1 + i, 1].float(), s[1 + i, 0].float()]
        return gs

    def params_b(self):
        vb = self.vb
        ps = []
        for unit in list(vb.fusion_layers) + list(vb.concat_layers):
            for m in (unit.local_embedding, unit.global_embedding, unit.global_gate):
                ps += [m.full_layer[2].weight, m.full_layer[3].weight, m.full_layer[3].bias]
        ps += [vb.residual_conv.full_layer[2].weight, vb.residual_conv.full_layer[2].bias]
        return ps

    def grads_b(self, st):
        s = st.local_sums
        gs = []
        for idx in range(5, NS):
            gs += [st.dW[idx].t().reshape(64, 1, 3).contiguous(), s[idx, 1].float(), s[idx, 0].float()]
        gs += [st.dWr.view(512, 64, 1), st.dbr]
        return gs


class VPStageA(torch.autograd.Function):
    """x [B,512,T] -> pooled g [B,64,Tg]; its backward runs after VPStageB's (through the GlobalAttention glue) and returns dx + stage-A parameter gradients"""

    @staticmethod
    def forward(ctx, trainer, holder, x, *params):
        with torch.no_grad():
            g, st = trainer.forward_a(x, holder.slopes)
        holder.st = st
        ctx.trainer, ctx.holder = trainer, holder
        return g

    @staticmethod
    def backward(ctx, dg):
        trainer, st = ctx.trainer, ctx.holder.st
        with torch.no_grad():
            dx = trainer.backward_a(st, dg)
            grads = trainer.grads_a(st)
        ctx.holder.st = None
        return (None, None, dx) + tuple(grads)


class VPStageB(torch.autograd.Function):
    """(g' [B,64,Tg]; saved stage-A tensors) -> block output [B,512,T]"""

    @staticmethod
    def forward(ctx, trainer, holder, g2, *params):
        with torch.no_grad():
            out = trainer.forward_b(holder.st, g2)
        ctx.trainer, ctx.holder = trainer, holder
        return out

    @staticmethod
    def backward(ctx, dout):
        trainer, st = ctx.trainer, ctx.holder.st
        trainer.check_equal_batches()
        with torch.no_grad():
            dg2 = trainer.backward_b(st, dout)
            grads = trainer.grads_b(st)
        return (None, None, dg2) + tuple(grads)


class _Holder:
    st = None
    slopes = None


# ---- GlobalAttention on HIP (csrc/vp_attn.hip) ------------------------------------------------------------------------------------
def attn_supported(ga) -> bool:
    """the RTFS-Net family's GlobalAttention: 64 channels, 8 heads, positional encoding, FFN 64 -> 128 -> 64 with a k = 3 depth-wise refiner"""
    try:
        m, f = ga.MHSA, ga.FFN
        return (type(ga).__name__ == "GlobalAttention" and m.attention.embed_dim == 64 and m.attention.num_heads == 8 and m.attention.in_proj_weight is not None
                and f.encoder.out_chan == 128 and f.refiner.kernel_size == 3 and abs(m.norm1.eps - EPS) < 1e-12 and abs(m.norm2.eps - EPS) < 1e-12
                # the kernels add the positional table and the packed in-projection bias unconditionally and index tokens batch-first
                and hasattr(m.pos_enc, "pe") and m.attention.in_proj_bias is not None and m.attention.batch_first
                and max(float(m.attention.dropout), float(m.dropout_layer.p), float(m.drop_path_layer.p), float(f.dropout_layer.p)) < 1.0)
    except (AttributeError, TypeError):
        return False


def attn_params(ga):
    """the 16 parameter tensors in the packing order of csrc/vp_attn.hip (VaOff)"""
    m, f = ga.MHSA, ga.FFN
    return [m.norm1.weight, m.norm1.bias, m.attention.in_proj_weight, m.attention.in_proj_bias, m.attention.out_proj.weight, m.attention.out_proj.bias,
            m.norm2.weight, m.norm2.bias, f.encoder.full_layer[2].weight, f.encoder.full_layer[3].norm.weight, f.encoder.full_layer[3].norm.bias,
            f.refiner.full_layer[2].weight, f.refiner.full_layer[2].bias, f.decoder.full_layer[2].weight, f.decoder.full_layer[3].norm.weight,
            f.decoder.full_layer[3].norm.bias]


_MASK_CACHE = {}


def attn_masks(ga, B, Tg, dev):
    """One draw of the stochastic layers of a training step as multiplicative keep-masks [B][8 Tg^2 + 64 Tg + 3] (0 or 1 / keep probability):
    nn.MultiheadAttention's dropout on the attention probabilities, nn.Dropout on the attention output (attention.py:67-68), DropPath after the
    MHSA (:73), after the FFN refiner and after the FFN decoder (conv_layers.py:253-256; per-utterance Bernoulli, scaled by 1 / keep as timm does).
    None in eval mode or when every probability is 0."""
    m, f = ga.MHSA, ga.FFN
    if not ga.training:
        return None
    ps = (float(m.attention.dropout), float(m.dropout_layer.p), float(m.drop_path_layer.p), float(f.dropout_layer.p))
    if max(ps) == 0.0:
        return None
    key = (Tg, ps, str(dev))
    tab = _MASK_CACHE.get(key)
    if tab is None:  # per-column drop probability and scale, built once per shape (host -> device copies are blocking transfers)
        na, ne = 8 * Tg * Tg, Tg * 64
        thr = torch.tensor([ps[0]] * na + [ps[1]] * ne + [ps[2], ps[3], ps[3]])
        tab = _MASK_CACHE[key] = (thr.to(dev), (1.0 / (1.0 - thr)).to(dev))
    thr, scl = tab
    return (torch.rand(B, thr.numel(), device=dev) >= thr).to(torch.float32) * scl


class VPAttnFn(torch.autograd.Function):
    """g [B,64,Tg] -> GlobalAttention(g): one HIP launch forward, one backward (which recomputes the forward in LDS: only g and the masks are kept)"""

    @staticmethod
    def forward(ctx, ga, masks, g, *params):
        with torch.no_grad():
            g = g.detach().float().contiguous()
            B, _, Tg = g.shape
            packed = torch.cat([p.detach().float().reshape(-1) for p in params])
            pe = ga.MHSA.pos_enc.pe[0, :Tg].float().contiguous()
            out = torch.empty_like(g)
            lib.call("rtfs_vp_attn_fwd", g, packed, pe, masks, out, B, Tg)
        ctx.ga, ctx.saved, ctx.shapes = ga, (g, packed, pe, masks), [p.shape for p in params]
        return out

    @staticmethod
    def backward(ctx, dout):
        if ctx.saved is None:
            raise RuntimeError("VPAttnFn: second backward through the same step (retain_graph=True): the saved tensors are released after the first one")
        g, packed, pe, masks = ctx.saved
        B, _, Tg = g.shape
        with torch.no_grad():
            dg = torch.empty_like(g)
            dpar = torch.zeros_like(packed)
            lib.call("rtfs_vp_attn_bwd", g, packed, pe, masks, dout.detach().float().contiguous(), dg, dpar, B, Tg)
            grads, o = [], 0
            for shp in ctx.shapes:
                n = int(torch.Size(shp).numel())
                grads.append(dpar[o:o + n].view(shp))
                o += n
        ctx.saved = None
        return (None, None, dg) + tuple(grads)


def global_attention_train(ga, g: torch.Tensor) -> torch.Tensor:
    """GlobalAttention of the VP block under autograd on the HIP kernels; other configurations / more than 16 tokens: the PyTorch module"""
    if not attn_supported(ga) or not (2 <= g.shape[-1] <= 16) or g.shape[1] != 64:
        return ga(g)
    params = attn_params(ga)
    if packed_count() != sum(p.numel() for p in params):
        raise RuntimeError("GlobalAttention parameter packing does not match csrc/vp_attn.hip (VaOff)")
    return VPAttnFn.apply(ga, attn_masks(ga, g.shape[0], g.shape[-1], g.device), g, *params)


def packed_count() -> int:
    return lib.load().rtfs_vp_attn_param_count()


def vp_block_train(trainer: VPTrainer, x: torch.Tensor, slopes=None) -> torch.Tensor:
    """the VP block of one training step: HIP stage A -> GlobalAttention (HIP, csrc/vp_attn.hip) -> HIP stage B.
    slopes: (gateway PReLU slope, projection PReLU slope) as Python floats if the caller already has them on the host."""
    holder = _Holder()
    holder.slopes = slopes
    g = VPStageA.apply(trainer, holder, x, *trainer.params_a())
    g2 = global_attention_train(trainer.vb.globalatt[0], g)
    return VPStageB.apply(trainer, holder, g2, *trainer.params_b())




@torch.no_grad()
def vp_block_eval(vb, x, slopes=None):
    """The VP block in eval mode WITHOUT autograd on the multi-launch kernels (any length up to 4096 frames; the one-kernel inference form
    csrc/vp.hip holds 4 s): stage A, GlobalAttention (HIP: the one-workgroup LDS form up to 16 pooled tokens, its workspace form up to 1024; other configurations: the module), stage B; BatchNorm from the running statistics."""
    tr = vb.__dict__.get("_hip_eval_trainer")  # kept on the module itself (not a parameter / buffer / child: invisible to state_dict), dies with it
    if tr is None or tr.bns[0] is not vb.projection.full_layer[3]:
        if not supported(vb):
            return vb(x).contiguous()
        tr = vb.__dict__["_hip_eval_trainer"] = VPTrainer(vb)
    if vb.training:
        raise RuntimeError("vp_block_eval is the inference path (model.eval())")
    g, st = tr.forward_a(x, slopes)
    ga = vb.globalatt[0]
    if attn_supported(ga) and 2 <= g.shape[-1] <= 16:
        g2 = torch.empty_like(g)
        packed = torch.cat([p.detach().float().reshape(-1) for p in attn_params(ga)])
        lib.call("rtfs_vp_attn_fwd", g, packed, ga.MHSA.pos_enc.pe[0, :g.shape[-1]].float().contiguous(), None, g2, g.shape[0], g.shape[-1])
    elif attn_supported(ga) and 16 < g.shape[-1] <= 1024 and ga.MHSA.pos_enc.pe.shape[1] >= g.shape[-1]:
        # more than 16 pooled tokens (utterances longer than 5.1 s): the workspace form of the same arithmetic (csrc/vp_attn.hip vp_attn_long_fwd_kernel)
        B, _, Tg = g.shape
        g2 = torch.empty_like(g)
        packed = torch.cat([p.detach().float().reshape(-1) for p in attn_params(ga)])
        work = torch.empty(B * lib.load().rtfs_vp_attn_long_work_floats(Tg), device=g.device)
        lib.call("rtfs_vp_attn_long_fwd", g, packed, ga.MHSA.pos_enc.pe[0, :Tg].float().contiguous(), g2, work, B, Tg)
    else:
        g2 = ga(g)
    return tr.forward_b(st, g2)


# ---- video side of the CAF cell under autograd (csrc/tfar.hip caf_video_kernel / caf_video_bwd_kernel) -------------------------------
class CAFVideoFn(torch.autograd.Function):
    """v1 [B,512,Tv] -> (att, rsz) [B,Tv,256]: attention_embed / resize grouped 1x1 convolutions + gLN, head mean, softmax over Tv
    (layers/fusion.py:255,262-265).  No BatchNorm on this side, so the forward is the inference kernel; the adjoint recomputes it."""

    @staticmethod
    def forward(ctx, v1, *params):
        with torch.no_grad():
            v1 = v1.detach().float().contiguous()
            B, _, Tv = v1.shape
            ps = [p.detach().float().contiguous() for p in params]
            att = torch.empty(B, Tv, 256, device=v1.device)
            rsz = torch.empty_like(att)
            lib.call("rtfs_caf_video_fwd", v1, *ps, att, rsz, B, Tv)
        ctx.saved = (v1, ps)
        return att, rsz

    @staticmethod
    def backward(ctx, datt, drsz):
        if ctx.saved is None:
            raise RuntimeError("CAFVideoFn: second backward through the same step (retain_graph=True): the saved tensors are released after the first one")
        v1, ps = ctx.saved
        B, _, Tv = v1.shape
        with torch.no_grad():
            dv = torch.empty_like(v1)
            flat = torch.zeros(sum(p.numel() for p in ps), device=v1.device)
            grads, o = [], 0
            for p in ps:
                grads.append(flat[o:o + p.numel()])
                o += p.numel()
            lib.call("rtfs_caf_video_bwd", v1, *ps, datt.detach().float().contiguous(), drsz.detach().float().contiguous(), dv, *grads, B, Tv)
        ctx.saved = None
        return (dv,) + tuple(g.view(p.shape) for g, p in zip(grads, ps))


def caf_video_train(cell, v1):
    """(att, rsz) of ATTNFusionCell's video side on HIP kernels, forward and backward; parameter order of rtfs_caf_video_fwd"""
    a, r = cell.attention_embed.full_layer, cell.resize.full_layer
    if v1.shape[1] != 512 or cell.in_chan_a != 256 or cell.kernel_size != 4:
        B = v1.shape[0]
        att = cell.attention_embed(v1).reshape(B, cell.in_chan_a, cell.kernel_size, -1).mean(2)
        return torch.softmax(att, -1).transpose(1, 2).contiguous(), cell.resize(v1).transpose(1, 2).contiguous()
    return CAFVideoFn.apply(v1, a[2].weight, a[2].bias, a[3].norm.weight, a[3].norm.bias, r[2].weight, r[2].bias, r[3].norm.weight, r[3].norm.bias)
