"""Flat-buffer Adam for the MI355X trainer (reference: torch.optim.Adam as built by
src/diff_gfdn/trainer.py:152-228).

``FlatAdam`` re-points every parameter of the param groups into ONE flat fp32 device buffer and
gathers every ``.grad`` into a second one (one multi-tensor copy), then updates all of them with a single HIP kernel
(csrc/optim.hip).  Consequences used by the trainer:
  * the data-parallel all-reduce runs on ``flat_grad`` in place -- no pack / unpack copies;
  * ``zero_grad`` is one memset; the update is one launch (torch's capturable Adam issues ~15 tiny
    foreach kernels per group);
  * learning rates live in a device tensor (one float per group), so StepLR can change them while the
    step is replayed from a HIP graph: call :meth:`sync_lr` after ``scheduler.step()``.
It is a ``torch.optim.Optimizer`` (param_groups / lr schedulers work); update maths and defaults are
those of torch.optim.Adam without amsgrad / weight decay.
"""
from typing import Iterable

import torch

from . import hip_ops as ops


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params: Iterable, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 extra_slots: int = 0):
        """``extra_slots``: floats appended to the gradient buffer (``bucket`` = gradients + slots, ``extra`` = the
        slots): a data-parallel trainer parks its per-rank loss terms there, so that ONE all-reduce of ``bucket``
        carries the gradients and the loss scalars ("[all parameter grads | per-loss partial sums]")."""
        defaults = dict(lr=lr, betas=betas, eps=eps)
        super().__init__(params, defaults)
        self.param_groups = [g for g in self.param_groups if len(g['params']) > 0]
        if len(self.param_groups) > 255:
            raise ValueError("at most 255 learning-rate groups")
        plist = [(gi, p) for gi, g in enumerate(self.param_groups) for p in g['params']]
        dev = plist[0][1].device
        if dev.type != 'cuda':
            raise RuntimeError("FlatAdam needs parameters on the GPU (no CPU fallback)")
        for _, p in plist:
            if p.dtype != torch.float32:
                raise TypeError("FlatAdam handles float32 parameters")
        n = sum(p.numel() for _, p in plist)
        self.flat_param = torch.empty(n, dtype=torch.float32, device=dev)
        self.bucket = torch.zeros(n + int(extra_slots), dtype=torch.float32, device=dev)
        self.flat_grad = self.bucket[:n]
        self.extra = self.bucket[n:]
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.step_count = torch.zeros(1, dtype=torch.float32, device=dev)
        self._block_counter = torch.zeros(1, dtype=torch.int32, device=dev)      # see gfdn_adam_step_counted
        # second counter, always equal to step_count after a step: `step_range(..., second=True)` reads and advances
        # this one, so that two ranges can be stepped from two streams without ordering the launches
        self.step_count2 = torch.zeros(1, dtype=torch.float32, device=dev)
        self._block_counter2 = torch.zeros(1, dtype=torch.int32, device=dev)
        seg = torch.empty(n, dtype=torch.uint8)
        self._grad_views, self._params = [], []
        off = 0
        for gi, p in plist:
            k = p.numel()
            self.flat_param[off:off + k].copy_(p.detach().reshape(-1))
            p.data = self.flat_param[off:off + k].view(p.shape)          # parameter = view of the flat buffer
            self._grad_views.append(self.flat_grad[off:off + k].view(p.shape))
            self._params.append(p)
            seg[off:off + k] = gi
            off += k
        self.seg = seg.to(dev)
        self.lr_seg = torch.zeros(len(self.param_groups), dtype=torch.float32, device=dev)
        self._lr_host = None
        self.sync_lr()

    def sync_lr(self):
        """Push the groups' (host) learning rates to the device table the kernel reads."""
        lrs = [float(g['lr']) for g in self.param_groups]
        if lrs != self._lr_host:
            self.lr_seg.copy_(torch.tensor(lrs, dtype=torch.float32), non_blocking=False)
            self._lr_host = lrs

    def zero_grad(self, set_to_none: bool = True):
        """Gradients are dropped (autograd then hands over fresh tensors without an accumulate
        kernel per parameter); :meth:`pack_grads` gathers them into the flat buffer."""
        for p in self._params:
            p.grad = None

    @torch.no_grad()
    def pack_grads(self):
        """flat_grad <- all parameter gradients, one multi-tensor copy (zeros where a parameter got
        no gradient this step).  Call before an all-reduce of ``flat_grad``."""
        views, grads = [], []
        for v, p in zip(self._grad_views, self._params):
            if p.grad is None:
                v.zero_()
            else:
                views.append(v)
                grads.append(p.grad)
        if views:
            torch._foreach_copy_(views, grads)
        self._packed = True

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError
        if not torch.cuda.is_current_stream_capturing():
            self.sync_lr()
        if not getattr(self, '_packed', False):
            self.pack_grads()
        self._packed = False
        b1, b2 = self.defaults['betas']
        ops.adam_step(self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, self.seg,
                      self.lr_seg, self.step_count, b1, b2, self.defaults['eps'], self._block_counter,
                      mirror=self.step_count2)
        # the kernel wrote the parameters through raw pointers: tell autograd / version-keyed caches
        torch.autograd.graph.increment_version(self._params)

    @torch.no_grad()
    def step_range(self, lo: int, hi: int, second: bool = False):
        """Adam update of the flat range [lo, hi) only, on the current stream.  A step split in two covers the buffer
        with ONE ``second=False`` call and ONE ``second=True`` call (any order, any streams): each advances its own
        counter (``step_count`` / ``step_count2``), which the unsplit :meth:`step` keeps equal."""
        b1, b2 = self.defaults['betas']
        sl = slice(lo, hi)
        cnt, blk = (self.step_count2, self._block_counter2) if second else (self.step_count, self._block_counter)
        ops.adam_step(self.flat_param[sl], self.flat_grad[sl], self.exp_avg[sl], self.exp_avg_sq[sl], self.seg[sl],
                      self.lr_seg, cnt, b1, b2, self.defaults['eps'], blk)
        self._packed = False

    def flat_range(self, param) -> tuple:
        """(lo, hi) of ``param`` in the flat buffers."""
        for p, v in zip(self._params, self._grad_views):
            if p is param:
                lo = (v.data_ptr() - self.flat_grad.data_ptr()) // 4
                return lo, lo + v.numel()
        raise KeyError("not a parameter of this optimiser")

    def state_tensors(self):
        return [self.exp_avg, self.exp_avg_sq, self.step_count, self.step_count2]
