"""The data-parallel training step: forward + backward + gradient all-reduce + Adam.

What the reference gets from `torch.nn.parallel.DistributedDataParallel` (`main_dist.py:68-79`:
bucketed gradient all-reduce overlapped with the backward pass) and `Learner.train_epoch`
(`utils/trn_utils.py:590-615`: zero_grad -> forward -> loss -> backward -> optimizer.step), built for
one process per MI355X over RCCL:

  * the step is cut into SEGMENTS -- [forward + heads/TxEncoder backward], trunk s5, trunk s4, the
    rest -- each captured in its own hipGraph; the gradient bucket a segment completes (a contiguous
    range of the fp32 gradient arena, `ParamArena.bucket_ranges`) is all-reduced asynchronously on
    RCCL's stream while the next segment computes; Adam (one more graph) waits for the buckets;
  * `overlap=False` is the same mechanism with ONE segment (whole forward + backward, one all-reduce);
  * without a process group the whole step, Adam included, is one hipGraph.  `adam_overlap=True` /
    `VS_ADAM_OVERLAP=1` uses the same segments to start Adam EARLY -- the step count ticks once, and the update
    of a segment's arena range runs on a side stream as soon as that segment's gradients are final -- bitwise the
    plain step (tests/test_gpu_train_step.py).  Measured on MI355X, batch 8: 13.39-13.41 ms against 13.11-13.13
    for one launch at the end (the segmented backward alone: 13.13-13.15): the update is a 2.3 GB HBM stream
    and the kernels it runs beside lose more to the contention than the overlap hides, so it is OFF by default;
  * `grad_bf16=True`: the bucket payload is bf16 (half the xGMI bytes: 153 MB instead of 305 MB for
    SlowFast-R50 + 6-layer TxEncoder) -- a segment's graph ends with the cast of its bucket into a
    bf16 arena, RCCL sums that, and the Adam kernel reads the bf16 sums against its fp32 master
    parameters and moments (`vs_adam_step_dev_cast_g16`).

A failed hipGraph capture RAISES (bench.py then exits non-zero): a silently eager multi-GPU line
would be a different measurement under the same name.
"""
import os

import torch
import torch.distributed as dist


class TrainStep:
    def __init__(self, mdl, loss_fn, arena, opt, batch, world=1, overlap=None, use_dist=None,
                 grad_bf16=False, adam_overlap=None, grad_fill=True):
        self.mdl, self.loss_fn, self.arena, self.opt, self.batch = mdl, loss_fn, arena, opt, batch
        self.world = world
        self.use_dist = (dist.is_available() and dist.is_initialized()) if use_dist is None else use_dist
        self.collectives = self.use_dist  # bench.py's rank-0-only instrumented pass switches them off
        self.grad_bf16 = bool(grad_bf16) and self.use_dist
        self.trunk = getattr(mdl, "sf_mdl", None)
        can_overlap = self.trunk is not None and hasattr(self.trunk, "BWD_SEGMENTS")
        self.overlap = (self.use_dist and world > 1 if overlap is None else bool(overlap)) and can_overlap
        # single process: ranged Adam beside the backward pass (needs the segmented backward and the arena on a GPU)
        want = os.environ.get("VS_ADAM_OVERLAP", "0") not in ("0", "") if adam_overlap is None else bool(adam_overlap)
        self.adam_overlap = (want and not self.use_dist and can_overlap and arena.data.is_cuda
                             and hasattr(opt, "step_range"))
        self._adam_stream = None
        self.loss = None
        # The per-step memset of the gradient arena (300 MB for SlowFast-R50 + TxEncoder) is only needed by
        # parameters whose gradient arrives through autograd's AccumulateGrad (p.grad += g).  The HIP modules
        # WRITE p.grad in place and return None for it, so AccumulateGrad never runs for them.
        #   True (default): memset every step -- what `optimizer.zero_grad()` means (utils/trn_utils.py:596).
        #   "learn" (bench.py): the first step fills the arena and watches which parameters AccumulateGrad
        #       touches (tensor hooks: called with None when a backward returned no gradient); later steps zero
        #       only those, and every later python-level step -- eager steps and the pass a hipGraph is captured
        #       from -- keeps watching: a gradient arriving for a parameter outside the learned set RAISES
        #       instead of being added onto the previous step's (a replayed graph is exactly its captured pass,
        #       so a clean capture covers every replay).
        #   False: never.
        # VS_GRAD_FILL=0 / 1 / learn overrides (A/B runs).
        env = os.environ.get("VS_GRAD_FILL", "")
        if env in ("0", "1", "learn"):
            grad_fill = {"0": False, "1": True, "learn": "learn"}[env]
        if grad_fill is None:  # round-2 spelling of "learn"
            grad_fill = "learn"
        self.grad_fill = grad_fill
        self._accumulated = None  # learned: parameters that need a zero gradient before every backward pass
        self.graphs = None  # segment graphs + the Adam graph, or [whole-step graph]
        self._eager_stream = None
        self.segments = self._build_segments()

    # ---- pieces ------------------------------------------------------------------------------
    def fwd_bwd(self):
        a = self.arena
        mod = getattr(self.mdl, "vid_feat_encoder", None)
        fired, tr_hook = [], None
        if TrainStep.late_transposes and isinstance(mod, torch.nn.Module):
            def refresh(m, inp):  # (a forward pre-hook's return value replaces the input: return None)
                if not fired:
                    fired.append(1)
                    a.transposes_async()
            tr_hook = mod.register_forward_pre_hook(refresh)
        else:
            a.transposes_async()  # dgrad weight images of the last update, beside the forward pass
        hooks = self._zero_grads()
        out = self.mdl(self.batch)
        if tr_hook is not None:
            tr_hook.remove()
            if not fired:
                a.transposes_async()
        loss = self.loss_fn(out, self.batch)["loss"]
        loss.backward()
        for h in hooks:
            h.remove()
        a._join_transposes()  # no-op unless no dgrad ran (keeps a captured graph closed)
        self.loss = loss.detach()
        return loss

    def _zero_grads(self):
        """Before the forward pass: what `optimizer.zero_grad()` has to do for this model (see `grad_fill`).
        Returns the hook handles of a learning step."""
        if self.grad_fill is True:
            self.opt.zero_grad()
            return []
        if self.grad_fill is False:
            self.opt.zero_grad(fill=False)  # only re-attaches replaced gradients
            return []
        if self._accumulated is not None:
            self.opt.zero_grad(fill=False)
            known = self._accumulated
            for p in known:
                p.grad.zero_()

            def guard(g, q):
                if g is not None and not any(q is r for r in known):
                    name = next((n for n, r in self.mdl.named_parameters() if r is q), "?")
                    raise RuntimeError(
                        f"TrainStep(grad_fill='learn'): parameter {name} received its gradient through autograd "
                        "on this step but not on the step the zero-fill set was learned from; its gradient would "
                        "accumulate across steps.  Use grad_fill=True, or build a new TrainStep after changing "
                        "which modules run.")
            return [p.register_hook(lambda g, q=p: guard(g, q)) for p in self.arena.params
                    if not any(p is r for r in known)]
        if self.arena.data.is_cuda and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("TrainStep: run one eager step() before capture() (it learns which gradients "
                               "need the per-step zero fill)")
        self.opt.zero_grad()
        seen = []
        self._accumulated = seen

        def note(g, q):  # autograd calls a tensor hook with None when the backward returned no gradient for q
            if g is not None and not any(q is r for r in seen):
                seen.append(q)
        return [p.register_hook(lambda g, q=p: note(g, q)) for p in self.arena.params]

    # The refresh of the transposed weight images (0.2 ms of HBM-bound side-stream kernels) has no reader before the
    # backward pass.  Beside the stems it competes with the heaviest streaming kernels of the step; the 8-token encoder
    # section between the trunk's forward and backward leaves the chip idle -- so, when the model has a
    # `vid_feat_encoder` (the first module behind the trunk), the refresh is launched from a forward pre-hook on it
    # (and right after the forward pass if that module never ran).  VS_TRANSPOSE_LATE=0: at the start of the step.
    late_transposes = os.environ.get("VS_TRANSPOSE_LATE", "1") != "0"

    def _adam(self):
        self.opt.step(world=self.world, defer_transposes=True, grad_bf16=self.grad_bf16)

    def _build_segments(self):
        """[(callable, (lo, hi))]: after `callable` the gradients in arena.grad[lo:hi] are final."""
        a = self.arena
        if not (self.overlap or self.adam_overlap):
            if self.trunk is not None:
                self.trunk.defer_backward = False
            return [(self.fwd_bwd, (0, a.numel))]
        trunk = self.trunk
        trunk.defer_backward = True
        segs = list(trunk.BWD_SEGMENTS)
        ranges = a.bucket_ranges([trunk.backward_segment_modules(sg) for sg in segs])
        out = [(self.fwd_bwd, ranges[-1])]  # everything outside the trunk is done after autograd
        for sg, rg in zip(segs, ranges[:-1]):
            out.append((lambda sg=sg: trunk.run_backward_segment(sg), rg))
        return out

    def _reduce(self, lo, hi):
        if not (self.collectives and self.use_dist):
            return None
        return self.arena.all_reduce_range(lo, hi, async_op=True, bf16=self.grad_bf16, packed=True)

    def _pack(self, lo, hi):
        if self.grad_bf16:
            self.arena.pack_grad_bf16(lo, hi)

    # ---- eager ---------------------------------------------------------------------------------
    def _step_adam_overlapped(self):
        main = torch.cuda.current_stream()
        if self._adam_stream is None:
            self._adam_stream = torch.cuda.Stream(device=self.arena.data.device)
        side = self._adam_stream
        self.opt.tick()
        late = os.environ.get("VS_ADAM_OVERLAP") == "2"  # A/B: the segmented backward, every range updated at the end
        for fn, (lo, hi) in self.segments:
            fn()
            if late:
                continue
            side.wait_stream(main)  # this range's gradients are final; its parameters have no reader left
            with torch.cuda.stream(side):
                self.opt.step_range(lo, hi, world=self.world)
        if late:
            for _, (lo, hi) in self.segments:
                self.opt.step_range(lo, hi, world=self.world)
        main.wait_stream(side)
        self.opt.finish_ranged(defer_transposes=True)
        return self.loss

    def step(self):
        if self.arena.data.is_cuda and not torch.cuda.is_current_stream_capturing():
            self._eager_stream = torch.cuda.current_stream()  # capture() records on the same stream (see there)
        if self.adam_overlap:
            return self._step_adam_overlapped()
        works = []
        for fn, (lo, hi) in self.segments:
            fn()
            self._pack(lo, hi)
            works.append(self._reduce(lo, hi))
        for w in works:
            if w is not None:
                w.wait()
        self._adam()
        return self.loss

    # ---- hipGraph ------------------------------------------------------------------------------
    def capture(self):
        """Capture the step.  Call after at least one eager `step()` on a side stream (allocator and
        lane-stream warm-up).  Raises RuntimeError when a capture fails."""
        # A process group's watchdog thread polls its events while this thread captures; the default
        # (global) capture mode treats that as an illegal call and invalidates the capture.
        mode = "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"
        inner = []  # an exception raised by the step itself: ending the aborted capture fails too, with a less useful error
        # Capture on the stream the eager warm-up step ran on: the kernels' scratch buffers are cached per stream
        # (ops._workspace), and on a stream of its own the capture would allocate them again INSIDE the graph -- the
        # zero fill of the arrival-counter workspaces then replays with every step (73 us per step, measured).
        st = getattr(self, "_eager_stream", None)
        on = {} if st is None or st == torch.cuda.default_stream(st.device) else {"stream": st}

        def guarded(fn):
            try:
                fn()
            except Exception as e:  # noqa: BLE001
                inner.append(e)
                # The step died half way: streams it forked into the capture (transposes, the fast pathway's, the
                # weight-gradient lane) are still un-joined, and ending such a capture fails in a way that leaves the
                # process unusable for any later capture (ROCm 7.2: "legacy stream depends on a capturing blocking
                # stream" from then on).  Join them so that the capture ends legally; the graph is discarded.
                try:
                    self._join_forked_streams()
                except Exception:  # noqa: BLE001
                    pass
                raise
        try:
            if not self.use_dist:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode=mode, **on):
                    guarded(self.step)
                self.graphs = [g]
            else:
                graphs, pool = [], None
                fns = [(lambda fn=fn, r=r: (fn(), self._pack(*r))) for fn, r in self.segments] + [self._adam]
                for fn in fns:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, pool=pool, capture_error_mode=mode, **on):
                        guarded(fn)
                    pool = g.pool()
                    graphs.append(g)
                self.graphs = graphs
            torch.cuda.synchronize()
        except Exception as e:
            e = inner[0] if inner else e
            self.graphs = None
            if self.trunk is not None:
                self.trunk._deferred = None
            raise RuntimeError(f"hipGraph capture of the training step failed: {e!r}") from e

    def _join_forked_streams(self):
        """Make the current (capturing) stream wait for every side stream of this package that is part of the capture."""
        from . import trunk as _trunk

        main = torch.cuda.current_stream()
        cands = [getattr(self.arena, "_tr_stream", None), self._adam_stream]
        cands += list(_trunk.VideoTrunk._side_streams.values()) + list(_trunk.VideoTrunk._reduce_streams.values())
        cands += [lane[1] for lane in _trunk._WgradLanes.lanes.values()]
        for st in cands:
            if st is None or st.cuda_stream == main.cuda_stream:
                continue
            with torch.cuda.stream(st):
                capturing = torch.cuda.is_current_stream_capturing()
            if capturing:
                main.wait_stream(st)
        _trunk._WgradLanes.join_all()
        if getattr(self.arena, "_tr_pending", False):
            self.arena._join_transposes()

    def replay(self):
        if self.graphs is None:
            raise RuntimeError("replay() before capture()")
        if not self.use_dist:
            self.graphs[0].replay()
            return
        works = []
        for g, (_, (lo, hi)) in zip(self.graphs[:-1], self.segments):
            g.replay()
            works.append(self._reduce(lo, hi))
        for w in works:
            if w is not None:
                w.wait()
        self.graphs[-1].replay()

    def run(self):
        if self.graphs is not None:
            self.replay()
        else:
            self.step()
