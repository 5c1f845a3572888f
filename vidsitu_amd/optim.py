"""Flat-arena parameters, gradients and Adam for the data-parallel step.

The reference wraps the model in torch DDP (`main_dist.py:68-79`: bucketed fp32
all-reduce of ~39 M gradient elements, plus a BN-buffer broadcast and
`find_unused_parameters`) and steps `torch.optim.Adam(betas=(0.9, 0.99))`
(`main_dist.py:50`).  Here every parameter and every gradient is a view into ONE
fp32 buffer each, so a step is: zero the gradient arena (one memset) ->
forward/backward (HIP kernels write gradients in place) -> ONE all-reduce over
RCCL (`torch.distributed`, backend "nccl") -> ONE fused Adam launch
(`vs_adam_step`, the 1/world_size averaging folded in) -> refresh the bf16
kernel-layout weight copies.  No unused parameters exist (the upstream 2304->400
head is never built) and BN running statistics are not re-broadcast (SURVEY.md
section 5: rank 0's are the ones checkpointed).
"""
import torch
import torch.distributed as dist

from . import ops


def _dense_view(buf, off, p):
    """A view of buf[off: off+n] with p's shape AND p's memory order."""
    n = p.numel()
    flat = buf[off : off + n]
    if p.dim() == 5 and not p.is_contiguous():  # channels-last conv weight
        co, ci, kt, kh, kw = p.shape
        return flat.view(co, kt, kh, kw, ci).permute(0, 4, 1, 2, 3)
    return flat.view(p.shape)


def reference_param_order(model):
    """[(name, parameter | None)] in the order of the reference model's `parameters()`: this build's
    `named_parameters()` with the parameters only the reference constructs inserted where the reference
    registers them.  A module announces those through `reference_only_params() -> [(suffix, shape)]`
    (VideoTrunk: the upstream `head.projection` that `SlowFast_FeatModel` / `ResNet_FeatModel` build but
    `forward_features` never calls, `mdl_sf_base.py:21-34,133-135`)."""
    phantoms = {}
    for mname, m in model.named_modules():
        fn = getattr(m, "reference_only_params", None)
        if callable(fn):
            lst = fn()
            if lst:
                phantoms[mname] = lst
    named = list(model.named_parameters())
    out = []
    for i, (n, p) in enumerate(named):
        out.append((n, p))
        for prefix, lst in phantoms.items():
            pre = prefix + "." if prefix else ""
            nxt = named[i + 1][0] if i + 1 < len(named) else None
            if n.startswith(pre) and not (nxt is not None and nxt.startswith(pre)):
                out.extend((pre + suffix, None) for suffix, _ in lst)
    return out


class ParamArena:
    def __init__(self, model, adopt_conv=True):
        """adopt_conv=False keeps only the flat fp32 parameter / gradient bookkeeping (what the
        CPU gloo tests exercise); True also moves the bf16 kernel weights into arenas (GPU)."""
        self.model = model
        self.params = [p for p in model.parameters() if p.requires_grad]
        # every parameter in `model.parameters()` order, trainable or not: the index space of the
        # reference's `torch.optim.Adam(mdl.parameters())` state (ArenaAdam.state_dict)
        self.all_named = list(model.named_parameters())
        dev = self.params[0].device
        sizes = [(p.numel() + 3) // 4 * 4 for p in self.params]  # keep 16-byte alignment
        self.offsets = [0]
        for s in sizes:
            self.offsets.append(self.offsets[-1] + s)
        total = self.offsets[-1]
        self.data = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                v = _dense_view(self.data, off, p)
                v.copy_(p.detach())
                p.data = v
                p.grad = _dense_view(self.grad, off, p)
                p._vs_direct_grad = True  # HIP backward kernels may write p.grad in place
        self.numel = total
        self.grad16 = None  # bf16 image of the gradient arena (bf16 all-reduce payload), on demand
        self._tr_table, self._loose_convs, self.data_bf16 = None, [], None
        self._tr_stream, self._tr_pending = None, False
        self._lin_table, self.wt_f32 = None, None
        if adopt_conv:
            self._adopt_conv_weights(dev)
            self._adopt_linear_weights(dev)

    def _adopt_conv_weights(self, dev):
        """bf16 kernel-layout weights and their transposed dgrad images become views into two
        bf16 arenas with the fp32 arena's offsets: one cast launch + one batched transpose
        launch refresh every conv of the model."""
        from .trunk import Conv3dP

        self.data_bf16 = torch.zeros(self.numel, dtype=ops.BF16, device=dev)
        self.wt_bf16 = torch.zeros(self.numel, dtype=ops.BF16, device=dev)
        off_of = {id(p): off for p, off in zip(self.params, self.offsets)}
        rows, first = [], 0
        self._loose_convs = []
        for m in self.model.modules():
            if not isinstance(m, Conv3dP):
                continue
            off = off_of.get(id(m.weight))
            if off is None or m.cin_pad != m.cin:
                self._loose_convs.append(m)  # stems (Cin 3 -> 8 padding) keep their own copy
                continue
            n = m.weight.numel()
            kt, kh, kw = m.k
            m.w_bf16 = self.data_bf16[off : off + n].view(m.cout, kt, kh, kw, m.cin).permute(0, 4, 1, 2, 3)
            m.wt_bf16 = self.wt_bf16[off : off + n].view(m.cin, kt, kh, kw, m.cout).permute(0, 4, 1, 2, 3)
            m.arena_managed = True
            rows.append([off, m.cout, kt * kh * kw, m.cin, first])
            first += n
        self._tr_total = first
        self._tr_table = torch.tensor(rows, dtype=torch.int64, device=dev) if rows else None
        self.refresh()

    def _adopt_linear_weights(self, dev):
        """fp32 nn.Linear weights (heads, TxEncoder) get a transposed image [K][N] in a second fp32
        arena with the same offsets: the operand of dx = dy @ W in LinearFn.backward, refreshed for
        ALL linears by one launch instead of one small transpose per linear per step."""
        from torch import nn

        from .transformer_code import MultiHead

        off_of = {id(p): off for p, off in zip(self.params, self.offsets)}
        rows, first = [], 0
        mods, fused_w = [], {}
        # self-attention blocks whose three bias-free projections sit back to back in the arena (they do:
        # registration order wq, wk, wv): ONE [3d, d] matrix for the forward / weight-gradient GEMMs and
        # ONE transposed image [d][3d] for dx = dqkv @ W (transformer_code.FusedQKVAttnFn)
        for mh in self.model.modules():
            if not isinstance(mh, MultiHead):
                continue
            ws = [mh.wq.weight, mh.wk.weight, mh.wv.weight]
            if any(id(w) not in off_of or w.dtype != torch.float32 or w.shape != ws[0].shape for w in ws) \
                    or any(l.bias is not None for l in (mh.wq, mh.wk, mh.wv)):
                continue
            n, k = ws[0].shape
            o0 = off_of[id(ws[0])]
            if off_of[id(ws[1])] != o0 + n * k or off_of[id(ws[2])] != o0 + 2 * n * k:
                continue
            rows.append([o0, 3 * n, 1, k, first])
            first += 3 * n * k
            fused_w[id(ws[0])] = fused_w[id(ws[1])] = fused_w[id(ws[2])] = (mh, o0, n, k, ws)
        for m in self.model.modules():
            if isinstance(m, nn.Linear) and id(m.weight) in off_of and m.weight.dtype == torch.float32 \
                    and id(m.weight) not in fused_w:
                n, k = m.weight.shape
                rows.append([off_of[id(m.weight)], n, 1, k, first])
                first += n * k
                mods.append(m)
        if not rows:
            return
        rows.sort(key=lambda r: r[0])  # (the batched kernels search by first index: keep it ascending)
        first = 0
        for r in rows:
            r[4] = first
            first += r[1] * r[3]
        self.wt_f32 = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self._lin_total = first
        self._lin_table = torch.tensor(rows, dtype=torch.int64, device=dev)
        self._lin_weights = [m.weight for m in mods]
        self._fused_qkv = []
        off_row = {r[0]: r for r in rows}
        for m in mods:
            n, k = m.weight.shape
            o = off_of[id(m.weight)]
            m.weight._vs_wt = self.wt_f32[o: o + n * k].view(k, n)
        done = set()
        for mh, o0, n, k, ws in fused_w.values():
            if id(mh) in done:
                continue
            done.add(id(mh))
            for w in ws:  # no per-projection image any more: LinearFn transposes on demand if ever asked
                w._vs_wt = None
                w._vs_wt_version = None
            mh._qkv = {"w": self.data[o0: o0 + 3 * n * k].view(3 * n, k),
                       "dw": self.grad[o0: o0 + 3 * n * k].view(3 * n, k),
                       "wt": self.wt_f32[o0: o0 + 3 * n * k].view(k, 3 * n),
                       "weights": ws, "grads": [w.grad for w in ws], "versions": [w._version for w in ws]}
            self._fused_qkv.append(mh._qkv)
        ops.transpose_f32_batched(self.data, self.wt_f32, self._lin_table, self._lin_total)
        for w in self._lin_weights:
            w._vs_wt_version = w._version

    def zero_grad(self, fill=True):
        """fill=False: skip the 300 MB memset.  Valid when every parameter's gradient is WRITTEN
        (overwrite semantics) by the backward -- true for the HIP modules of this package (conv /
        BN / linear / LayerNorm kernels write `param.grad` in place; checked by
        tests/test_gpu_trunk.py::test_every_gradient_is_overwritten) -- and the arena was zeroed
        once, so the alignment gaps between parameters stay zero."""
        if fill:
            self.grad.zero_()
        for p, off in zip(self.params, self.offsets):  # re-attach if something replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                p.grad = _dense_view(self.grad, off, p)

    def all_reduce(self):
        """Sum gradients over ranks (averaging happens inside the Adam kernel)."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM)
            return dist.get_world_size()
        return 1

    def bucket_ranges(self, module_groups):
        """[start, end) element ranges of the arena covered by each group of modules (gradient
        buckets for an all-reduce that overlaps the rest of the backward); parameters that belong
        to no group form a final bucket.  Ranges must be contiguous in the arena (parameter
        registration order) or ValueError is raised."""
        owner = {}
        for gi, mods in enumerate(module_groups):
            for m in mods:
                for p in m.parameters():
                    owner[id(p)] = gi
        spans = {}
        for p, off, end in zip(self.params, self.offsets, self.offsets[1:]):
            gi = owner.get(id(p), len(module_groups))
            lo, hi, n = spans.get(gi, (off, off, 0))
            spans[gi] = (min(lo, off), max(hi, end), n + (end - off))
        out = []
        for gi in range(len(module_groups) + 1):
            if gi not in spans:
                out.append((0, 0))
                continue
            lo, hi, n = spans[gi]
            if hi - lo != n:
                raise ValueError(f"gradient bucket {gi} is not contiguous in the arena")
            out.append((lo, hi))
        return out

    def pack_grad_bf16(self, lo, hi):
        """grad[lo:hi] (fp32) -> grad16[lo:hi] (bf16), the payload of a bf16 gradient all-reduce.
        GPU: one launch of the cast kernel (captured at the end of a segment's hipGraph); the CPU
        bookkeeping mode of the gloo tests uses a torch cast."""
        if hi <= lo:
            return
        if self.grad16 is None:
            self.grad16 = torch.zeros(self.numel, dtype=torch.bfloat16, device=self.grad.device)
        if self.grad.is_cuda:
            ops.cast_bf16(self.grad[lo:hi], self.grad16[lo:hi])
        else:
            self.grad16[lo:hi].copy_(self.grad[lo:hi])

    def unpack_grad_bf16(self, lo, hi):
        """grad16[lo:hi] -> grad[lo:hi] (consumers that read fp32 gradients: the CPU mode and
        torch optimizers; the GPU Adam kernel reads grad16 directly)."""
        if hi > lo:
            self.grad[lo:hi].copy_(self.grad16[lo:hi])

    def all_reduce_range(self, lo, hi, async_op=False, bf16=False, packed=False):
        """All-reduce grad[lo:hi] (SUM).  Returns the work handle when async_op (None if nothing
        to do: single process or empty range).  bf16: the payload is the bf16 image grad16[lo:hi]
        (`packed`: the caller has already run pack_grad_bf16 for the range); the sum stays in grad16."""
        if hi <= lo or not (dist.is_available() and dist.is_initialized()):
            return None
        if bf16:
            if not packed:
                self.pack_grad_bf16(lo, hi)
            return dist.all_reduce(self.grad16[lo:hi], op=dist.ReduceOp.SUM, async_op=async_op)
        return dist.all_reduce(self.grad[lo:hi], op=dist.ReduceOp.SUM, async_op=async_op)

    def broadcast_params(self, src=0):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.broadcast(self.data, src=src)
            self.refresh()

    def refresh(self, cast=True, transposes=True):
        """After the fp32 arena changed (optimizer step, broadcast, load): one cast, one
        batched transpose, the two padded stems, and the trunks are marked up to date.
        cast=False: the bf16 arena was already written (fused into the Adam kernel);
        transposes=False: the dgrad images are refreshed later by `transposes_async`."""
        if self.data_bf16 is None:
            return
        if cast:
            ops.cast_bf16(self.data, self.data_bf16)
        if transposes:
            self._join_transposes()
            self._run_transposes()
        for m in self._loose_convs:
            m.refresh()
        for m in self.model.modules():
            if hasattr(m, "_version_key") and hasattr(m, "_weights_version"):
                m._weights_version = m._version_key()

    def _run_transposes(self):
        if self._tr_table is not None:
            ops.weight_transpose_batched(self.data_bf16, self.wt_bf16, self._tr_table, self._tr_total)
        if self._lin_table is not None:
            ops.transpose_f32_batched(self.data, self.wt_f32, self._lin_table, self._lin_total)
            for w in self._lin_weights:
                w._vs_wt_version = w._version
            for f in getattr(self, "_fused_qkv", ()):
                f["versions"] = [w._version for w in f["weights"]]

    # The transposed (dgrad) weight images are only read by the backward pass: refresh them on a
    # side stream at the START of a step, beside the forward pass; the first dgrad joins.
    def transposes_async(self):
        if self._tr_table is None and self._lin_table is None:
            return
        from .trunk import Conv3dP

        self._join_transposes()
        main = torch.cuda.current_stream()
        if self._tr_stream is None:
            self._tr_stream = torch.cuda.Stream(device=main.device)
        self._tr_stream.wait_stream(main)
        with torch.cuda.stream(self._tr_stream):
            self._run_transposes()
        self._tr_pending = True
        Conv3dP._pending_arenas.add(self)  # every arena with a refresh in flight is joined by wt()

    def _join_transposes(self):
        if self._tr_pending:
            torch.cuda.current_stream().wait_stream(self._tr_stream)
            self._tr_pending = False
            from .trunk import Conv3dP

            Conv3dP._pending_arenas.discard(self)

class ArenaAdam:
    """torch.optim.Adam semantics (lr, betas, eps; no weight decay) on the arena."""

    def __init__(self, arena, lr=1e-4, betas=(0.9, 0.99), eps=1e-8):
        self.arena, self.lr, self.betas, self.eps = arena, lr, betas, eps
        self.m = torch.zeros_like(arena.data)
        self.v = torch.zeros_like(arena.data)
        self.t = torch.zeros(1, dtype=torch.int32, device=arena.data.device)  # device-side: graph safe

    def zero_grad(self, fill=True):
        self.arena.zero_grad(fill)

    # ---- torch.optim.Adam's checkpoint format (`utils/trn_utils.py:699-716` saves
    #      `optimizer.state_dict()`; `:689-697` loads it).  The reference builds
    #      `Adam(mdl.parameters())` (`main_dist.py:50`): index i = position in `mdl.parameters()`
    #      order over ALL parameters -- frozen ones (`embed_tokens.weight` of TxEncoderOld / new_conc)
    #      and the never-used upstream `sf_mdl.head.projection.{weight,bias}` included -- and a state
    #      entry {"step", "exp_avg", "exp_avg_sq"} exists only for parameters that received a gradient.
    def _reference_index(self):
        """[(name, parameter | None)] in the reference's `mdl.parameters()` order; None marks a
        parameter the reference model owns and this build never constructs."""
        return reference_param_order(self.arena.model)

    def state_dict(self):
        a = self.arena
        off_of = {id(p): off for p, off in zip(a.params, a.offsets)}
        step = float(int(self.t.item()))
        state, index = {}, self._reference_index()
        for i, (_, p) in enumerate(index):
            if p is None or id(p) not in off_of:  # reference-only or frozen: no state, index kept
                continue
            off = off_of[id(p)]
            state[i] = {"step": torch.tensor(step),
                        "exp_avg": _dense_view(self.m, off, p).detach().clone().contiguous(),
                        "exp_avg_sq": _dense_view(self.v, off, p).detach().clone().contiguous()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0,
                 "amsgrad": False, "params": list(range(len(index)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        a = self.arena
        groups = sd["param_groups"]
        order = [i for g in groups for i in g["params"]]
        index = self._reference_index()
        real = [(n, p) for n, p in index if p is not None]
        if len(order) == len(index):
            slots = index
        elif len(order) == len(real):  # a state written over the parameters this build constructs
            slots = real
        else:
            raise ValueError(f"optimizer state has {len(order)} parameters; the model has {len(index)} in the "
                             f"reference's order ({len(real)} of them built here)")
        g0 = groups[0]
        self.lr, self.betas, self.eps = g0["lr"], tuple(g0["betas"]), g0["eps"]
        if any(g.get("weight_decay", 0) != 0 or g.get("amsgrad", False) for g in groups):
            raise ValueError("weight_decay / amsgrad are not implemented by the fused Adam kernel")
        off_of = {id(p): off for p, off in zip(a.params, a.offsets)}
        self.m.zero_()
        self.v.zero_()
        steps = set()
        with torch.no_grad():
            for key, (name, p) in zip(order, slots):
                st = sd["state"].get(key)
                if st is None or p is None:  # never received a gradient / not built here
                    continue
                if id(p) not in off_of:
                    continue  # frozen here: its moments are never read
                off = off_of[id(p)]
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise ValueError(f"optimizer state of {name}: shape {tuple(st['exp_avg'].shape)} "
                                     f"!= parameter {tuple(p.shape)}")
                _dense_view(self.m, off, p).copy_(st["exp_avg"].to(self.m.device))
                _dense_view(self.v, off, p).copy_(st["exp_avg_sq"].to(self.v.device))
                steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): one shared counter here")
        self.t.fill_(steps.pop() if steps else 0)

    def tick(self):
        """Start of a ranged step: the step count (device memory) goes up once."""
        ops.adam_tick(self.t)

    def step_range(self, lo, hi, world=1, grad_bf16=False):
        """The update of arena elements [lo, hi) against the count `tick()` set; no refresh of derived weight
        images (the caller runs `finish_ranged()` once every range is done)."""
        a = self.arena
        if hi <= lo:
            return
        g = (a.grad16 if grad_bf16 else a.grad)[lo:hi]
        pb = a.data_bf16[lo:hi] if a.data_bf16 is not None else None
        ops.adam_step_dev_range(a.data[lo:hi], g, self.m[lo:hi], self.v[lo:hi], pb, self.lr, self.betas[0],
                                self.betas[1], self.eps, self.t, grad_scale=1.0 / world)

    def finish_ranged(self, defer_transposes=False):
        a = self.arena
        if a.data_bf16 is not None:
            a.refresh(cast=False, transposes=not defer_transposes)
        else:
            a.refresh()

    def step(self, world=1, defer_transposes=False, grad_bf16=False):
        """defer_transposes: leave the dgrad weight images stale; the caller refreshes them with
        `arena.transposes_async()` at the start of the next step (bench.py).
        grad_bf16: the (all-reduced) gradients are read from the bf16 arena `arena.grad16`."""
        a = self.arena
        if a.data_bf16 is not None:
            if grad_bf16:
                ops.adam_step_dev_cast_g16(a.data, a.grad16, self.m, self.v, a.data_bf16, self.lr,
                                           self.betas[0], self.betas[1], self.eps, self.t,
                                           grad_scale=1.0 / world)
            else:
                ops.adam_step_dev_cast(a.data, a.grad, self.m, self.v, a.data_bf16, self.lr, self.betas[0],
                                       self.betas[1], self.eps, self.t, grad_scale=1.0 / world)
            a.refresh(cast=False, transposes=not defer_transposes)
        else:
            if grad_bf16:
                a.unpack_grad_bf16(0, a.numel)
            ops.adam_step_dev(a.data, a.grad, self.m, self.v, self.lr, self.betas[0], self.betas[1],
                              self.eps, self.t, grad_scale=1.0 / world)
            a.refresh()
