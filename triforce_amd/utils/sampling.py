"""Sampling helpers — mirror of the reference's utils/sampling.py (norm_logits :43-60,
top_k_top_p_filter :5-27, sample :63-66, max_fn :68-75), device-agnostic torch (plumbing that
runs inside the captured hipGraphs), plus the explicit uniform stream that feeds the device-side
sampler / accept kernels.

Tie rule: the reference sorts with torch.sort(descending=True), whose order among equal logits is
implementation-defined; here the sort is *stable* (lowest token id first among equals) so greedy
emulation (temperature=1, top_p->0) is deterministic on every device.
"""
import torch
from torch.nn import functional as F

from .. import ops


def top_k_top_p_filter(logits: torch.Tensor, top_k: int = 0, top_p: float = 0.0):
    if top_k > 0:
        kth = torch.topk(logits, min(top_k, logits.size(-1)))[0][:, [-1]]
        logits = logits.masked_fill(logits < kth, float("-inf"))
    if top_p > 0.0:
        sorted_logits, sorted_indices = torch.sort(logits, descending=True, stable=True)
        cumulative = torch.cumsum(F.softmax(sorted_logits, dim=-1), dim=-1)
        # drop a sorted position when the mass BEFORE it already exceeds top_p (rank 0 always stays)
        drop_sorted = torch.zeros_like(cumulative, dtype=torch.bool)
        drop_sorted[..., 1:] = cumulative[..., :-1] > top_p
        drop = torch.zeros_like(drop_sorted).scatter(1, sorted_indices, drop_sorted)
        logits = logits.masked_fill(drop, float("-inf"))
    return logits


def norm_logits(logits: torch.Tensor, temperature=0.6, top_k=-1, top_p=0.9) -> torch.Tensor:
    """(rows, vocab) fp32 logits -> probabilities after temperature / top-k / top-p."""
    assert logits.dim() == 2
    if logits.is_cuda and top_k <= 0 and 0.0 < top_p and logits.shape[-1] <= ops.TOPP_MAX_VOCAB \
            and logits.dtype == torch.float32:
        return ops.topp_probs(logits.contiguous(), temperature, top_p)      # fused HIP kernel, no vocabulary sort
    logits = logits / temperature
    logits = top_k_top_p_filter(logits, top_k=top_k, top_p=top_p)
    return F.softmax(logits, dim=-1)


def max_fn(x):
    """norm(max(x, 0)) — the residual distribution of speculative sampling."""
    x_max = torch.where(x > 0, x, torch.zeros_like(x))
    return x_max / torch.sum(x_max, dim=-1, keepdim=True)


class UniformSource:
    """Device-resident stream of U[0,1) numbers consumed by the sampling / accept kernels.

    ``take(n)`` exposes the next n numbers (a device view) without consuming them; ``advance(k)``
    consumes k.  The reference draws lazily with torch.rand(1)/multinomial per decision
    (decoding.py:97,192); a stream with explicit consumption reproduces that order with zero host
    syncs.  ``values`` injects a fixed sequence (cycled) for parity tests.
    """

    MAX_TAKE = 256        # most numbers one kernel may look at (the Sequoia tree walk: one per examined child)

    def __init__(self, device, seed=None, values=None, block=1 << 16):
        self.device = torch.device(device)
        self.block = block
        self.pos = 0
        self._fixed = values is not None
        # Round 5: the buffer is allocated ONCE and refilled in place, and ``cursor`` is a device-resident int64 copy of
        # ``pos`` — kernels captured inside a hipGraph read their numbers as buf[cursor + k] and the kernel that closes a
        # decision advances the cursor itself (ops.*_cur); the host mirrors it with advance().  ``device_cursor`` says
        # whether the device copy is current: the plain take() / advance() users (eager kernels that get a pointer) leave it
        # stale, cursor_tensor() brings it up to date with one fill when it is.
        self.cursor = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.device_cursor = True
        if self._fixed:
            v = torch.as_tensor(values, dtype=torch.float32).flatten()
            reps = (block + self.MAX_TAKE + v.numel() - 1) // v.numel() + 1
            self._period = v.numel()
            self.buf = v.repeat(reps).to(self.device)
        else:
            self.gen = torch.Generator(device=self.device)
            if seed is not None:
                self.gen.manual_seed(seed)
            else:
                self.gen.seed()
            self.buf = torch.rand(block + self.MAX_TAKE, generator=self.gen, device=self.device)

    def _room(self, n):
        """Make sure buf[pos : pos + n] exists (refill / wrap, in place: captured graphs keep the buffer's address)."""
        assert n <= self.MAX_TAKE
        if self.pos + n > self.block:
            if self._fixed:
                self.pos %= self._period
            else:
                self.buf.copy_(torch.rand(self.block + self.MAX_TAKE, generator=self.gen, device=self.device))
                self.pos = 0
            self.device_cursor = False

    def take(self, n):
        self._room(n)
        return self.buf[self.pos:self.pos + n]

    def advance(self, k):
        """Consume k numbers on the host side only (an eager kernel got them through take())."""
        self.pos += int(k)
        self.device_cursor = False

    def cursor_tensor(self, n):
        """The device cursor, current and with room for n numbers: for kernels that read buf[cursor + k] (ops.*_cur)."""
        self._room(n)
        if not self.device_cursor:
            self.cursor.fill_(self.pos)
            self.device_cursor = True
        return self.cursor

    def advanced_on_device(self, k, at=None):
        """A *_cur kernel consumed k numbers and advanced the device cursor itself; ``at``: the cursor value it reported."""
        if at is not None and int(at) != self.pos:
            raise RuntimeError(f"uniform stream out of step: device cursor {int(at)} != host position {self.pos}")
        self.pos += int(k)


def sample(probs: torch.Tensor, num_samples=1, rng: UniformSource = None):
    """Draw one token id from ``probs`` ((V,) or (1,V)) -> int64 tensor of shape (1,1) on the device.
    Replaces torch.multinomial (sampling.py:63-66) by inverse-CDF sampling with an explicit uniform."""
    assert num_samples == 1
    p = probs.reshape(-1).contiguous()
    if rng is None:
        rng = _default_rng(p.device)
    out = torch.empty(1, dtype=torch.int64, device=p.device)
    ops.sample_inverse_cdf(p, rng.take(1), out)
    rng.advance(1)
    return out.view(1, 1)


_rngs = {}


def _default_rng(device):
    key = str(device)
    if key not in _rngs:
        _rngs[key] = UniformSource(device)
    return _rngs[key]
