"""Adam on flat buffers: the optimizer step of the reference's training loop
(codes/models/VideoSR_AllPair_model_YCbCr_Split.py:122-124 ``torch.optim.Adam(optim_params, lr, weight_decay, betas)``,
``.step()`` at :187) as ONE HIP kernel launch per parameter group.

MI355X-first layout: every parameter of the network is a view into one contiguous f32 buffer, every gradient a view
into a second one laid out identically (the buffer ``realvsr_amd.dist.BucketedGradAllReduce`` all-reduces in
buckets), and the two Adam moments are two more.  The update is then a single streaming pass over 4 x 13 MB
(EDVR-M) instead of torch's multi-tensor chain over 144 tensors; ``zero_grad`` is one memset that also sets every
``p.grad`` to None, so that autograd ADOPTS the gradient tensors of the next backward -- which the fused operators hand over
as the parameters' own views of the flat gradient buffer (no per-parameter ``grad += new`` kernels); gradients that arrive
from elsewhere are copied home (``FlatBuffers.rebind``) before the buffer is reduced or consumed.

Arithmetic = torch's single-tensor Adam (amsgrad=False), op for op (rvsr_adam_step in csrc/train_kernels.hip);
``param_groups`` / ``state`` keep torch's schema, so LR schedulers that edit ``param_groups[i]['lr']``
(base_model.py:36-60) and ``state_dict()`` work unchanged.  Two behaviours of torch.optim.Adam that a dense flat update
would otherwise lose are kept: a parameter whose ``grad`` is None at step time is skipped entirely (no moment decay, no
weight decay, no drift: its three ranges are restored after the launch; the step count that sets the bias corrections is
kept per GROUP, so such a parameter follows its group's schedule afterwards where torch's per-parameter count would lag), and emptying ``optimizer.state`` -- what the
reference's MultiStepLR_Restart(clear_state=True) does at a restart (codes/models/lr_scheduler.py) -- resets the flat
moments and step counters (``reset_state``).
"""
import math

import torch

from . import functional as RF

_ALIGN = 64  # elements: every parameter starts on a 256-byte boundary of the flat buffers


class FlatBuffers:
    """Re-homes parameters (and their gradients) into contiguous buffers, group by group.

    Within a group the parameters are laid out in REVERSE registration order -- roughly the order backward produces
    their gradients -- so that the bucketed all-reduce can send the first buckets while backward is still running."""

    def __init__(self, groups):
        groups = [[p for p in g if p.requires_grad] for g in groups]
        flat = [p for g in groups for p in g]
        if not flat:
            raise ValueError('FlatBuffers: no trainable parameters')
        ref = flat[0]
        for p in flat:
            if p.dtype != torch.float32 or p.device != ref.device:
                raise ValueError('FlatBuffers: parameters must be float32 on one device')
        self.order, self.offset, self.group_range = [], {}, []
        off = 0
        for g in groups:
            start = off
            for p in reversed(g):
                self.order.append(p)
                self.offset[p] = off
                off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
            self.group_range.append((start, off))
        self.numel = off
        self.param = torch.zeros(off, dtype=torch.float32, device=ref.device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=ref.device)
        # ids of the parameters whose flat gradient view a fused backward has been given since the last zero_grad
        # (functional._pgrad hands a home out once; see there)
        self.claimed = set()
        # parameters that had received no gradient when rebind() first ran after the last zero_grad (None = not decided yet):
        # what FlatAdam.step must leave untouched.  Decided BEFORE rebind() hands every parameter a flat view -- with world > 1 the
        # reducer's finish() rebinds ahead of the optimizer, and `p.grad is None` would never be seen by step() otherwise.
        self.no_grad = None
        with torch.no_grad():
            for p in self.order:
                o, n = self.offset[p], p.numel()
                view = self.param[o:o + n].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = self.grad[o:o + n].view(p.shape)
                # where the fused operators of realvsr_amd.functional write this parameter's gradient (see zero_grad)
                p._rvsr_grad_home = (self.grad, o, self.claimed)
        RF.packed_weights.invalidate()   # (parameters moved: weight images packed from the old storage are stale)

    def grad_view(self, p):
        o = self.offset[p]
        return self.grad[o:o + p.numel()].view(p.shape)

    def zero_grad(self):
        """One memset, and every ``p.grad`` becomes None until backward fills it.  With ``p.grad is None`` autograd's
        AccumulateGrad ADOPTS the incoming gradient tensor instead of adding it to an existing one, and the fused operators
        produce that tensor as the parameter's own view of the flat buffer (functional._pgrad): the gradient kernels write
        straight into the flat buffer and the 144 tiny ``grad += new`` kernels (and 144 temporaries) of a step disappear.
        Gradients that arrive from elsewhere (plain torch modules) are copied home by ``rebind``."""
        self.grad.zero_()
        self.claimed.clear()
        self.no_grad = None
        for p in self.order:
            p.grad = None

    def rebind(self, p=None):
        """Make ``p.grad`` (all parameters if None) the flat view again: zeros if no gradient arrived, a copy if autograd
        adopted a tensor that lives elsewhere.  A home that a fused backward wrote but autograd never adopted
        (``torch.autograd.grad``: the result went to the caller, not to ``p.grad``) is not a gradient of this step: zeroed.
        The first whole-buffer call after a zero_grad records which parameters had no gradient (``no_grad``).  With several ranks
        that set must be the same everywhere (the same graph on every rank -- DistributedDataParallel's own requirement without
        find_unused_parameters); RVSR_DIST_CHECK=1 makes BucketedGradAllReduce.finish() verify it."""
        if p is None and self.no_grad is None:
            self.no_grad = [q for q in self.order if q.grad is None]
        for q in (self.order if p is None else (p,)):
            o = self.offset[q]
            if q.grad is None:
                q.grad = self.grad[o:o + q.numel()].view(q.shape)
                if id(q) in self.claimed:
                    q.grad.zero_()
            elif q.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                view = self.grad[o:o + q.numel()].view(q.shape)
                view.copy_(q.grad)
                q.grad = view

    def check_bound(self):
        """Raise if something re-bound a parameter away from the flat buffers (net.to(), load with assign) or left a gradient
        outside them: the flat update would then silently work on stale memory.  (``p.grad is None`` = no gradient yet.)"""
        base_p, base_g = self.param.data_ptr(), self.grad.data_ptr()
        for p in self.order:
            o = 4 * self.offset[p]
            if p.data_ptr() != base_p + o:
                raise RuntimeError('a parameter was moved out of the flat buffer (module.to()/load with assign?)')
            if p.grad is not None and p.grad.data_ptr() != base_g + o:
                raise RuntimeError('a gradient lives outside the flat buffer; FlatBuffers.rebind() copies it home')


class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr, betas, eps, weight_decay) semantics on FlatBuffers."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError('FlatAdam: invalid hyper-parameter')
        super(FlatAdam, self).__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.buffers = FlatBuffers([g['params'] for g in self.param_groups])
        self.exp_avg = torch.zeros_like(self.buffers.param)
        self.exp_avg_sq = torch.zeros_like(self.buffers.param)
        self._steps = [0] * len(self.param_groups)
        self._step_t = [torch.tensor(0.0) for _ in self.param_groups]  # one host tensor per group, shared by its params
        self._bind_state()

    def zero_grad(self, set_to_none=False):
        self.buffers.zero_grad()

    def _bind_state(self):
        for gi, group in enumerate(self.param_groups):
            for p in group['params']:
                if not p.requires_grad:
                    continue
                o, n = self.buffers.offset[p], p.numel()
                self.state[p] = {'step': self._step_t[gi], 'exp_avg': self.exp_avg[o:o + n].view(p.shape),
                                 'exp_avg_sq': self.exp_avg_sq[o:o + n].view(p.shape)}

    def reset_state(self):
        """Forget the moments and step counts (torch: ``optimizer.state = defaultdict(dict)``)."""
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        self._steps = [0] * len(self.param_groups)
        for t in self._step_t:
            t.fill_(0.0)
        self._bind_state()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if len(self.state) == 0:          # a scheduler cleared the state (restart): start the moments over
            self.reset_state()
        # torch skips parameters without a gradient; the flat launch touches every element, so keep what must not move
        if self.buffers.no_grad is None:       # (else: the gradient all-reduce already rebound every parameter and recorded the set)
            self.buffers.no_grad = [p for p in self.buffers.order if p.grad is None]
        skipped = [(self.buffers.offset[p], self.buffers.offset[p] + p.numel()) for p in self.buffers.no_grad]
        saved = [(s, e, self.buffers.param[s:e].clone(), self.exp_avg[s:e].clone(), self.exp_avg_sq[s:e].clone())
                 for s, e in skipped]
        self.buffers.rebind()
        self.buffers.check_bound()
        for gi, (group, (s, e)) in enumerate(zip(self.param_groups, self.buffers.group_range)):
            if e == s:
                continue
            self._steps[gi] += 1
            t = self._steps[gi]
            beta1, beta2 = group['betas']
            bias_correction1 = 1 - beta1 ** t
            bias_correction2 = 1 - beta2 ** t
            step_size = group['lr'] / bias_correction1
            RF.adam_step_(self.buffers.param[s:e], self.buffers.grad[s:e], self.exp_avg[s:e], self.exp_avg_sq[s:e],
                          step_size, beta1, beta2, group['eps'], group['weight_decay'], math.sqrt(bias_correction2))
            self._step_t[gi].fill_(float(t))
        for s, e, pv, m1, m2 in saved:
            self.buffers.param[s:e].copy_(pv)
            self.exp_avg[s:e].copy_(m1)
            self.exp_avg_sq[s:e].copy_(m2)
        # the skip set belongs to THIS step's gradients: the next step decides again (a finish() that runs before it may pre-decide; gradients
        # cleared by nn.Module.zero_grad() / p.grad = None, or two steps without a zero_grad, must not reuse a stale list -- ADVICE r4)
        self.buffers.no_grad = None
        # the kernels above changed every parameter through raw pointers: re-pack all cached bf16 hi/lo weight images, one launch
        RF.packed_weights.repack()
        RF.dcn_offset_stats.advance()   # (the DCN forwards' halo choice lags by optimizer steps, functional.DcnOffsetStats)
        return loss

    def load_state_dict(self, state_dict):
        """torch's loader replaces the state tensors by copies; copy them back into the flat moments."""
        super(FlatAdam, self).load_state_dict(state_dict)
        with torch.no_grad():
            for gi, group in enumerate(self.param_groups):
                for p in group['params']:
                    st = self.state.get(p)
                    if not st or not p.requires_grad:
                        continue
                    o, n = self.buffers.offset[p], p.numel()
                    for name, flat in (('exp_avg', self.exp_avg), ('exp_avg_sq', self.exp_avg_sq)):
                        view = flat[o:o + n].view(p.shape)
                        view.copy_(st[name])
                        st[name] = view
                    self._steps[gi] = int(st['step'])
                    st['step'] = self._step_t[gi]
                self._step_t[gi].fill_(float(self._steps[gi]))
