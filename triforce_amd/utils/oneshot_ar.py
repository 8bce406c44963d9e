"""One-shot all-reduce for the decode-sized messages of the tensor-parallel path (csrc/allreduce.hip, C ABI
tf_allreduce_oneshot) — replaces dist.all_reduce at the reference's models/tensor_op.py:179,326,359 for messages up to
``max_elems`` fp16 values; larger ones (prefill chunks) stay on RCCL.

Per rank: one fine-grained staging buffer + one control block, exported to the peers with hipIpc handles (exchanged
through torch.distributed) and mapped once.  The producer GEMM writes this rank's partial straight into ``staging()``;
``reduce(staged, out)`` then runs the READY / reduce / DONE kernel on the current stream.  Results are bit-identical on
every rank (same values, same order, fp32 accumulation, one rounding).

``alternate=True`` (TRIFORCE_AR_ALTERNATE=1 through DistributedLlama) allocates two staging halves and uses them in
turn (tf_allreduce_oneshot_alt): the closing DONE handshake — one more cross-device round trip per exchange — is not
needed then.  ``staging()`` hands the producer the half the NEXT exchange will read; the kernel takes the half from
its device-side epoch and a disagreement with the half passed here is a sticky error (3), so a captured forward must
hold an even number of exchanges (the decode layer's two per layer do).
"""
import ctypes
import os

import torch

from .. import hip


class _RawBuffer:
    """Exposes a raw device pointer to torch through the CUDA array interface (no copy, no ownership)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"data": (int(ptr), False), "shape": tuple(shape), "typestr": typestr, "version": 2,
                                         "strides": None}


def reference_sum(partials, resid=None):
    """The arithmetic of the kernel: fp32 accumulation in rank order, one rounding to fp16; then (optionally) the
    residual added in fp16."""
    acc = torch.zeros_like(partials[0], dtype=torch.float32)
    for p in partials:
        acc = acc + p.float()
    out = acc.to(torch.float16)
    return out if resid is None else resid + out


class OneShotAllReduce:
    def __init__(self, rank, world, device, max_elems, peer_data=None, peer_flags=None, own=None, connect=True,
                 alternate=False):
        """Collective constructor (every rank of the default process group calls it) unless ``peer_data`` /
        ``peer_flags`` are given (single-process groups of virtual ranks: tests).  ``connect=False`` only allocates this
        rank's buffers — no collective — and leaves the handle exchange to ``connect()``, so a caller can let the ranks
        agree that every allocation succeeded before any of them enters the exchange (DistributedLlama does)."""
        self.rank, self.world, self.device, self.max_elems = rank, world, torch.device(device), int(max_elems)
        assert self.max_elems % 8 == 0
        self.alternate, self._issued = bool(alternate), 0          # _issued: exchanges launched or captured so far
        L = hip.lib()
        self._opened = []
        self._data = self._flags = None
        if own is None:
            own = (self._alloc(self.max_elems * 2 * (2 if self.alternate else 1)), self._alloc(L.tf_ar_flags_bytes()))
            self._owned = own
        else:
            self._owned = ()
        self.data_ptr, self.flags_ptr = own
        self._stage = torch.as_tensor(_RawBuffer(self.data_ptr, (self.max_elems * (2 if self.alternate else 1),), "<f2"),
                                      device=self.device)
        # the sticky error word is mirrored into pinned host memory: error() is then a plain host read (valid after any
        # stream synchronisation or host read of later work of the stream), not a blocking copy of the control block
        self._err_host = None
        try:
            word = torch.zeros(1, dtype=torch.int32).pin_memory()
            hip.check(L.tf_ar_set_error_mirror(ctypes.c_void_p(self.flags_ptr), ctypes.c_void_p(word.data_ptr())),
                      "tf_ar_set_error_mirror")
            self._err_host = word
        except Exception:                              # no pinned memory / no mapping: fall back to the copying read
            self._err_host = None
        if peer_data is not None:
            self._set_peers(peer_data, peer_flags)
        elif connect:
            self.connect()

    def _set_peers(self, peer_data, peer_flags):
        self._data = (ctypes.c_void_p * self.world)(*peer_data)
        self._flags = (ctypes.c_void_p * self.world)(*peer_flags)

    def connect(self):
        """The collective half of construction: export this rank's buffers, all-gather the handles, map the peers'.
        EVERY rank reaches the all-gather even when its own export failed (it contributes None), so the ranks cannot
        end up in different collectives; the failure is raised afterwards, on every rank that saw it."""
        self._set_peers(*self._exchange())

    @staticmethod
    def _alloc(nbytes):
        p = ctypes.c_void_p()
        hip.check(hip.lib().tf_ar_alloc(int(nbytes), ctypes.byref(p)), "tf_ar_alloc")
        return p.value

    def _exchange(self):
        import torch.distributed as dist
        L = hip.lib()
        n = L.tf_ar_ipc_handle_bytes()
        mine, failure = [], None
        try:
            for ptr in (self.data_ptr, self.flags_ptr):
                buf = ctypes.create_string_buffer(n)
                hip.check(L.tf_ar_get_ipc_handle(ctypes.c_void_p(ptr), buf), "tf_ar_get_ipc_handle")
                mine.append(bytes(buf.raw))
        except Exception as ex:                       # still take part in the all-gather below
            mine, failure = None, ex
        everyone = [None] * self.world
        dist.all_gather_object(everyone, mine)
        if failure is not None:
            raise failure
        if any(e is None for e in everyone):
            raise RuntimeError(f"rank(s) {[r for r, e in enumerate(everyone) if e is None]} could not export their "
                               "one-shot all-reduce buffers")
        data, flags = [], []
        for r, (hd, hf) in enumerate(everyone):
            if r == self.rank:
                data.append(self.data_ptr)
                flags.append(self.flags_ptr)
                continue
            for handle, dst in ((hd, data), (hf, flags)):
                p = ctypes.c_void_p()
                hip.check(L.tf_ar_open_ipc_handle(ctypes.create_string_buffer(handle, n), ctypes.byref(p)),
                          "tf_ar_open_ipc_handle")
                self._opened.append(p.value)
                dst.append(p.value)
        return data, flags

    # ------------------------------------------------------------------------------------------------------
    def _half(self):
        """Half of the staging buffer the next exchange reads: its epoch (exchanges so far + 1) & 1; 0 without
        alternation."""
        return (self._issued + 1) & 1 if self.alternate else 0

    def staging(self, rows, cols, packed=False):
        """(rows, cols) fp16 view of this rank's staging buffer: the producer kernel's output tensor — or, ``packed``,
        the same elements as a k-octet-major ops.Act block (the layout is the producer's and the consumer's business:
        the sum is element-wise)."""
        assert rows * cols <= self.max_elems and (rows * cols) % 8 == 0
        lo = self._half() * self.max_elems
        flat = self._stage[lo:lo + rows * cols]
        if packed:
            from ..ops import Act
            return Act.over(flat, rows, cols)
        return flat.view(rows, cols)

    def is_staged(self, t):
        """True when ``t`` starts where the next exchange will read this rank's partial."""
        return t.data_ptr() == self.data_ptr + 2 * self.max_elems * self._half()

    def fits(self, t):
        return t.dtype == torch.float16 and t.numel() <= self.max_elems and t.numel() % 8 == 0

    def reduce(self, staged, out, resid=None, ss_out=None):
        """out <- [resid +] sum over ranks of their staged partials.  ``staged`` must be (a prefix view of)
        ``staging()``; ``resid`` (fp16, may be ``out`` itself) is added in fp16 to the rounded sum; ``ss_out``
        (hidden / 16, 32) fp32 receives the per-panel sums of squares of the result rows (ops.ss_buffer)."""
        assert self._data is not None, "OneShotAllReduce.connect() has not run"
        from ..ops import Act
        pack_rows = 0
        if isinstance(out, Act):                       # k-octet-major residual stream: every operand in the same form
            assert isinstance(staged, Act) and staged.R == out.R == out.M and ss_out is not None
            assert resid is None or (isinstance(resid, Act) and resid.R == out.R)
            pack_rows = out.R
        else:
            assert not isinstance(staged, Act) and not isinstance(resid, Act)
            assert out.is_contiguous() and (resid is None or resid.is_contiguous())
        assert self.is_staged(staged) and out.dtype == torch.float16
        assert out.numel() == staged.numel() and out.data_ptr() != staged.data_ptr()
        if resid is not None:
            assert resid.dtype == torch.float16 and resid.numel() == out.numel()
        rp = ctypes.c_void_p(resid.data_ptr()) if resid is not None else None
        st = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        hidden = out.shape[-1]
        if ss_out is not None:
            assert ss_out.dtype == torch.float32 and ss_out.is_contiguous() and ss_out.shape == (hidden // 16, 32)
        if pack_rows:
            hip.check(hip.lib().tf_allreduce_oneshot_act(self._data, self._flags, self.rank, self.world, rp,
                                                         ctypes.c_void_p(out.data_ptr()), staged.numel(), hidden, pack_rows,
                                                         ctypes.c_void_p(ss_out.data_ptr()),
                                                         self.max_elems if self.alternate else 0,
                                                         self._half() if self.alternate else 0, st),
                      "tf_allreduce_oneshot_act")
            self._issued += 1
            return out
        # _issued counts exchanges that were actually enqueued (or captured): it advances only after the launch call
        # returned success, so a refused launch leaves host parity and device epoch in step
        if self.alternate:
            hip.check(hip.lib().tf_allreduce_oneshot_alt(self._data, self._flags, self.rank, self.world, rp,
                                                         ctypes.c_void_p(out.data_ptr()), staged.numel(), hidden,
                                                         ctypes.c_void_p(ss_out.data_ptr()) if ss_out is not None else None,
                                                         self.max_elems, self._half(), st), "tf_allreduce_oneshot_alt")
        elif ss_out is not None:
            hip.check(hip.lib().tf_allreduce_oneshot_add_ss(self._data, self._flags, self.rank, self.world, rp,
                                                            ctypes.c_void_p(out.data_ptr()), staged.numel(), hidden,
                                                            ctypes.c_void_p(ss_out.data_ptr()), st),
                      "tf_allreduce_oneshot_add_ss")
        else:
            hip.check(hip.lib().tf_allreduce_oneshot_add(self._data, self._flags, self.rank, self.world, rp,
                                                         ctypes.c_void_p(out.data_ptr()), staged.numel(), st),
                      "tf_allreduce_oneshot_add")
        self._issued += 1
        return out

    def resync(self):
        """Set the host-side exchange count from the device epoch (blocking; the stream must be idle — the caller
        synchronises first).  Needed after anything that may have counted an exchange the device never ran: a forward
        that raised midway, a graph capture that failed or was discarded (captured exchanges advance ``_issued`` but
        not the epoch).  Returns the count."""
        torch.cuda.synchronize(self.device)
        e = hip.lib().tf_ar_epoch(ctypes.c_void_p(self.flags_ptr))
        if e < 0:
            raise hip.TriforceHipError(f"tf_ar_epoch failed: {e}")
        self._issued = int(e)
        return self._issued

    def error(self):
        """0, or which wait timed out (1 READY, 2 DONE) at some point since creation (3: the alternating form found the
        partial staged in the other half than its epoch selects — an odd number of exchanges in a captured forward).  Sticky: after a timeout every
        later reduce returns at once and fills ``out`` with NaN (csrc/allreduce.hip) — callers poll this once per
        decode step (``check()``) and stop instead of emitting tokens computed from a reduction that never happened."""
        if self._err_host is not None:
            return int(self._err_host[0])
        return hip.lib().tf_ar_error(ctypes.c_void_p(self.flags_ptr))

    def error_device(self):
        """The error word as the control block holds it (blocking copy) — what error() mirrors."""
        return hip.lib().tf_ar_error(ctypes.c_void_p(self.flags_ptr))

    def inject_error(self, code):
        """Fault injection (tests, bench.py TRIFORCE_BENCH_INJECT_AR_ERROR): set / clear the sticky error word."""
        hip.check(hip.lib().tf_ar_inject_error(ctypes.c_void_p(self.flags_ptr), int(code)), "tf_ar_inject_error")

    def check(self, where=""):
        e = self.error()
        if e:
            what = ("read the staging half its producer did not write (exchanges out of step with the device epoch)"
                    if e == 3 else
                    f"timed out waiting for a peer's {'READY' if e == 1 else 'DONE' if e == 2 else e} flag")
            raise RuntimeError(f"rank {self.rank}: one-shot all-reduce {what}{(' (' + where + ')') if where else ''}; "
                               "its outputs since then are NaN-filled. Restart with TRIFORCE_ALLREDUCE=rccl.")

    def close(self):
        L = hip.lib()
        for p in self._opened:
            L.tf_ar_close_ipc_handle(ctypes.c_void_p(p))
        if self._err_host is not None and self._owned:
            L.tf_ar_set_error_mirror(ctypes.c_void_p(self.flags_ptr), None)
        for p in self._owned:
            L.tf_ar_free(ctypes.c_void_p(p))
        self._opened, self._owned = [], ()

    # ------------------------------------------------------------------------------------------------------
    @classmethod
    def local_group(cls, world, device, max_elems, alternate=False):
        """``world`` virtual ranks inside ONE process on ONE device (their kernels must run on different streams):
        exercises the flag protocol and the arithmetic without peer mappings."""
        L = hip.lib()
        owned = [(cls._alloc(max_elems * 2 * (2 if alternate else 1)), cls._alloc(L.tf_ar_flags_bytes()))
                 for _ in range(world)]
        data, flags = [o[0] for o in owned], [o[1] for o in owned]
        group = [cls(r, world, device, max_elems, peer_data=data, peer_flags=flags, own=owned[r], alternate=alternate)
                 for r in range(world)]
        for g, o in zip(group, owned):
            g._owned = o
        return group


class GemmExchange:
    """GEMM + all-reduce in ONE launch (csrc/gemv.hip SgXchg, C ABI tf_skinny_gemm_xchg): the o_proj / down_proj of the
    tensor-parallel decode layer with their exchange folded into the GEMM's epilogue — every workgroup exchanges its own
    16-column panel with the same workgroup of the other ranks (reference: models/tensor_op.py:175-181,353-360).

    Per rank: a staging buffer of two halves (exchange e uses half e & 1 of the DEVICE-side epoch, so there is no host
    bookkeeping and a captured launch replays correctly) and a control buffer (epoch / ticket / error + per-panel flags),
    both fine-grained and mapped by every peer through hipIpc.  Construction is collective like OneShotAllReduce's."""

    def __init__(self, rank, world, device, max_elems, peer_stage=None, peer_ctl=None, own=None, connect=True):
        self.rank, self.world, self.device, self.max_elems = rank, world, torch.device(device), int(max_elems)
        assert self.max_elems % 8 == 0
        L = hip.lib()
        # TRIFORCE_XCHG_TIMEOUT_MS: wall-clock limit of one flag wait (default 5 000).  The value is a kernel argument, i.e. it
        # is FROZEN into every graph captured afterwards — set it before initialize_graphs().  With replicated decisions the
        # ranks no longer meet in broadcasts: a host stall on one rank longer than this (a debugger, SIGSTOP, a long collection)
        # NaN-poisons its peers, and the sticky error word needs the collective reset() — give decode loops that may stall a
        # larger limit, keep the short one for the litmus and the tests (advisor, round 5).
        ms = os.environ.get("TRIFORCE_XCHG_TIMEOUT_MS")
        if ms:
            L.tf_xchg_tune(1, max(1, int(ms)))
        self._opened, self._stage, self._ctl = [], None, None
        if own is None:
            own = (OneShotAllReduce._alloc(self.max_elems * 2 * 2), OneShotAllReduce._alloc(L.tf_xchg_ctl_bytes()))
            self._owned = own
        else:
            self._owned = ()
        self.stage_ptr, self.ctl_ptr = own
        self._err_host = None
        try:
            word = torch.zeros(1, dtype=torch.int32).pin_memory()
            hip.check(L.tf_xchg_set_error(ctypes.c_void_p(self.ctl_ptr), 0, ctypes.c_void_p(word.data_ptr()), 1),
                      "tf_xchg_set_error")
            self._err_host = word
        except Exception:
            self._err_host = None
        if peer_stage is not None:
            self._set_peers(peer_stage, peer_ctl)
        elif connect:
            self.connect()

    def _set_peers(self, stage, ctl):
        self._stage = (ctypes.c_void_p * self.world)(*stage)
        self._ctl = (ctypes.c_void_p * self.world)(*ctl)

    def connect(self):
        helper = OneShotAllReduce.__new__(OneShotAllReduce)         # reuse the handle exchange (all-gather + hipIpc maps)
        helper.rank, helper.world, helper.data_ptr, helper.flags_ptr, helper._opened = \
            self.rank, self.world, self.stage_ptr, self.ctl_ptr, self._opened
        self._set_peers(*helper._exchange())

    def fits(self, rows, cols):
        return rows * cols <= self.max_elems and cols % 16 == 0 and cols // 16 <= 512

    def linear_reduce(self, a, w, x, ss_out):
        """x <- x + sum over ranks of (a . W^T) (fp16 partials, fp32 sum in rank order, one rounding, fp16 residual add:
        the arithmetic of ops.linear(out=staging) + OneShotAllReduce.reduce(resid=x)); ss_out <- per-panel sums of
        squares of the new x.  a / x: row-major tensors or ops.Act blocks (x's layout is also the staging layout)."""
        from .. import ops
        assert self._stage is not None, "GemmExchange.connect() has not run"
        assert isinstance(w, ops.PackedLinear) and w.wp is not None
        M = a.shape[0]
        assert tuple(x.shape) == (M, w.N) and a.shape[1] == w.K and self.fits(M, w.N)
        assert ss_out.dtype == torch.float32 and ss_out.is_contiguous() and ss_out.shape == (w.N // 16, 32)
        ap, asm, ask = ops._lay(a)
        xp, xsm, xsk = ops._lay(x)
        st = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        hip.check(hip.lib().tf_skinny_gemm_xchg(ops._ptr(w.wp), ap, asm, ask, self._stage, self._ctl, self.rank, self.world,
                                                self.max_elems, xp, xsm, xsk, xp, xsm, xsk,
                                                ctypes.c_void_p(ss_out.data_ptr()), M, w.N, w.K, st), "tf_skinny_gemm_xchg")
        return x

    def error(self):
        if self._err_host is not None:
            return int(self._err_host[0])
        return self.error_device()

    def error_device(self):
        return hip.lib().tf_xchg_error(ctypes.c_void_p(self.ctl_ptr))

    def inject_error(self, code):
        hip.check(hip.lib().tf_xchg_set_error(ctypes.c_void_p(self.ctl_ptr), int(code), None, 0), "tf_xchg_set_error")

    # ---- hardening (round 5; advisor round 4, verdict item 4) -------------------------------------------------------------
    @staticmethod
    def set_fenced(on):
        """Process-wide: later launches (and captures) of the fused exchange use the release / acquire FENCES (True) instead
        of s_waitcnt + system-scope accesses in issue order (False, the default).  Returns the previous setting."""
        return bool(hip.lib().tf_xchg_tune(0, 1 if on else 0))

    @staticmethod
    def set_timeout_ms(ms):
        return hip.lib().tf_xchg_tune(1, int(ms))

    def reset(self):
        """Zero this rank's control block (peer flags, per-panel exchange counts, error word).  COLLECTIVE by contract:
        every rank calls it between two barriers with no exchange in flight (after a time-out the per-panel counts of the
        ranks have drifted apart; nothing else brings them back together)."""
        hip.check(hip.lib().tf_xchg_reset(ctypes.c_void_p(self.ctl_ptr)), "tf_xchg_reset")

    def litmus(self, iters=100_000, rows=7, per_graph=100, delay_every=17, capture_mode="thread_local", hidden=None):
        """Message-passing litmus of THIS kernel on THIS group (the staged exchange kernel has its own:
        tools/xgmi_litmus.py): every iteration an ordinary kernel writes a fresh integer pattern as the activation
        (tf_ar_litmus_stage), tf_skinny_gemm_xchg multiplies it by an identity weight and exchanges the panels, and
        tf_ar_litmus_check compares every element of the sum with the value the SAME iteration's patterns give — a flag that
        overtook its panel, a stale or torn peer read shows up as a COUNT, not a hang.  ``per_graph`` iterations are captured
        as one hipGraph and replayed (no host synchronisation inside; this rank is delayed now and then so the peers run
        ahead as far as the protocol lets them).  Collective: every rank passes the same arguments.  Returns a dict."""
        from .. import ops
        L, dev = hip.lib(), self.device
        # ``hidden``: width of the identity exchange (default: what 32 rows leave of the staging half — the engine sizes it
        # as ONESHOT_MAX_ROWS x hidden_size with 32 rows; pass it explicitly when that is not how max_elems was chosen)
        hidden = self.max_elems // 32 if hidden is None else int(hidden)
        rows = min(rows, 32)
        assert hidden % 16 == 0 and hidden // 16 <= 512 and rows * hidden <= self.max_elems, \
            f"litmus shape {rows} x {hidden} does not fit a staging half of {self.max_elems} elements"
        eye = ops.PackedLinear(torch.eye(hidden, dtype=torch.float16, device=dev))
        a = torch.zeros(rows, hidden, dtype=torch.float16, device=dev)
        out = torch.zeros(rows, hidden, dtype=torch.float16, device=dev)
        ss = ops.ss_buffer(hidden, dev)
        it_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        bad = torch.zeros(1, dtype=torch.int64, device=dev)
        n = rows * hidden

        def vp(t):
            return ctypes.c_void_p(t.data_ptr())

        def one():
            st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            hip.check(L.tf_ar_litmus_stage(vp(a), n, self.rank, vp(it_dev), st), "tf_ar_litmus_stage")
            hip.check(L.tf_skinny_gemm_xchg(ops._ptr(eye.wp), vp(a), hidden, 8, self._stage, self._ctl, self.rank, self.world,
                                            self.max_elems, None, 8, 8, vp(out), hidden, 8, vp(ss), rows, hidden, hidden, st),
                      "tf_skinny_gemm_xchg")
            hip.check(L.tf_ar_litmus_check(vp(out), n, self.world, vp(it_dev), vp(bad), st), "tf_ar_litmus_check")

        for _ in range(2):                                     # eager warm-up (module load, first-touch), also checked
            one()
        torch.cuda.synchronize(dev)
        per = max(1, int(per_graph))
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side, capture_error_mode=capture_mode):
                for _ in range(per):
                    one()
        torch.cuda.current_stream(dev).wait_stream(side)
        replays = max(1, int(iters) // per)
        for j in range(replays):
            if delay_every and j % delay_every == 0 and (j // delay_every) % self.world == self.rank:
                torch.cuda._sleep(200_000)                     # ~0.1 ms: the peers run ahead as far as the protocol allows
            graph.replay()
        torch.cuda.synchronize(dev)
        res = {"iterations": 2 + replays * per, "mismatched_elements": int(bad.item()), "error_word": self.error_device(),
               "rows": rows, "hidden": hidden, "fenced": bool(L.tf_xchg_tune(0, -1))}
        del graph
        return res

    def check(self, where=""):
        e = self.error()
        if e:
            raise RuntimeError(f"rank {self.rank}: GEMM + exchange timed out waiting for a peer's panel flag"
                               f"{(' (' + where + ')') if where else ''}; its outputs since then are NaN-filled. "
                               "Restart with TRIFORCE_TP_GEMM_XCHG=0 or TRIFORCE_ALLREDUCE=rccl.")

    def close(self):
        L = hip.lib()
        for p in self._opened:
            L.tf_ar_close_ipc_handle(ctypes.c_void_p(p))
        if self._owned:
            L.tf_xchg_set_error(ctypes.c_void_p(self.ctl_ptr), 0, None, 1)
        for p in self._owned:
            L.tf_ar_free(ctypes.c_void_p(p))
        self._opened, self._owned = [], ()

    @classmethod
    def local_group(cls, world, device, max_elems):
        """``world`` virtual ranks in one process on one device (kernels on different streams): tests / shard benches."""
        L = hip.lib()
        owned = [(OneShotAllReduce._alloc(max_elems * 2 * 2), OneShotAllReduce._alloc(L.tf_xchg_ctl_bytes()))
                 for _ in range(world)]
        stage, ctl = [o[0] for o in owned], [o[1] for o in owned]
        group = [cls(r, world, device, max_elems, peer_stage=stage, peer_ctl=ctl, own=owned[r]) for r in range(world)]
        for g, o in zip(group, owned):
            g._owned = o
        return group
