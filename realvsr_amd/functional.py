"""Autograd glue between PyTorch tensors and the C ABI of librealvsr_hip.so.

PyTorch is used for device memory, streams and the autograd graph only; every FLOP of the hot
path runs in the HIP kernels (realvsr_amd/csrc).  Each ``Function`` below wraps one fused
operator; CPU tensors are refused (NotImplementedError, like the reference's operator:
codes/models/archs/dcn/deform_conv.py:109-110,124-125).
"""
import ctypes
import functools

import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib

ACT_NONE, ACT_RELU, ACT_LRELU = 0, 1, 2
ACT_MASK = 3   # rvsr_conv2d_forward only: out = conv * act'(residual) (include/realvsr_hip.h); internal to the fused backward nodes


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(*ts, any_float=False):
    """any_float: the deformable operators also take f64 / f16 tensors (their general path), everything else is float32."""
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise NotImplementedError('realvsr_amd operators run on MI355X (HIP) tensors only; got a %s tensor'
                                      % t.device.type)
        if t.dtype != torch.float32 and not (any_float and t.dtype in (torch.float64, torch.float16)):
            raise TypeError('realvsr_amd operators are float32 (like the reference training path); got %s' % t.dtype)
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError('realvsr_amd operators need all operands on one device; got %s and %s' % (dev, t.device))


def _c(t):
    return None if t is None else t.contiguous()


_workspaces = {}


def _workspace(nbytes, device):
    """Per-(device, stream) scratch buffer, grown on demand (kernels on one stream are ordered,
    so consecutive operators can share it)."""
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


class GradSink:
    """One gradient buffer for an activation that several fused operators consume.

    Autograd sums the gradients of a tensor with k consumers by k - 1 out-of-place adds (3 passes over the tensor each; the
    40 x 64 x 180 x 320 feature maps of the alignment stage have up to four consumers).  The kernels here can do that
    sum themselves: the DCN backward accumulates atomically into whatever its grad_input buffer holds, and the
    data-gradient convs take a fused residual that may be their own output buffer.  A GradSink is created per forward for such an
    activation and handed to its consumers:
      * a DEPOSITOR (every consumer but the owner) adds its gradient into `buf` in place and returns None to autograd;
      * the OWNER -- the consumer created FIRST in forward, hence run LAST in backward -- adds its own gradient the same way,
        closes the sink and returns the buffer: autograd sees one gradient, no adds, no zero fills.
    If the engine ever ran the owner early, the late depositors see `closed` and return their gradients normally: the
    result is the same sum, just through autograd's adds."""
    __slots__ = ('buf', 'closed', 'shape')

    def __init__(self, shape=None):
        self.buf, self.closed, self.shape = None, False, (tuple(shape) if shape is not None else None)

    def get(self, like):
        """The buffer to accumulate into (zeros on first use), or None once the owner has run."""
        if self.closed:
            return None
        if self.buf is None:
            self.buf = torch.zeros_like(like)
        return self.buf

    def close(self):
        self.closed = True
        buf, self.buf = self.buf, None
        return buf


_ADOPT_FLAT_GRADS = True   # parameter gradients are written in place of the flat buffer (optim.FlatBuffers)
_FUSE_GRAD_MASK = True     # ResBlock backward: relu' of the hidden activation in the epilogue of conv2's data gradient


def _pgrad(p, zero=False):
    """Output buffer for the gradient of parameter `p` inside a fused backward.

    When the parameter lives in optim.FlatBuffers, has no gradient yet this step (``p.grad is None`` after
    FlatBuffers.zero_grad) and its home has not been handed out since that zero_grad, the buffer is the parameter's own view of
    the flat gradient buffer: the kernel writes the gradient where the optimizer and the bucketed all-reduce read it, and
    autograd's AccumulateGrad adopts the returned view as ``p.grad`` (no ``grad += new`` kernel, no temporary).  The home is
    handed out ONCE per zero_grad (FlatBuffers.claimed): a parameter used by several fused operators in one backward (a module
    applied twice, shared weights) gets an ordinary temporary for every later use -- AccumulateGrad has not run between the uses,
    so ``p.grad`` is still None and a second hand-out would alias the first gradient.  Gradient accumulation over several
    backward passes and parameters outside FlatBuffers also take the temporary; autograd sums as usual."""
    home = getattr(p, '_rvsr_grad_home', None) if p is not None else None
    if home is not None and p.grad is None and p.dtype == torch.float32 and _ADOPT_FLAT_GRADS:
        buf, off, claimed = home
        if id(p) not in claimed:
            claimed.add(id(p))
            v = buf[off:off + p.numel()].view(p.shape)
            if zero:
                v.zero_()   # (accumulating kernels: do not rely on the caller having used FlatBuffers.zero_grad)
            return v
    return torch.zeros_like(p) if zero else torch.empty_like(p)



_PACK_VERIFY = False   # debug (set it from a test): on a cache hit, pack the weight again and compare


# ------------------------------------------------------------------------------------------ packed weights, once per step
class PackedWeights:
    """bf16 hi/lo weight images (what the conv / DCN kernels stage into LDS), packed ONCE per optimizer step.

    A conv block needs its weights re-packed ([m-block][chunk][hi|lo][tap][octet][row][8] bf16) -- once for the forward and once,
    transposed and flipped, for the data gradient.  The library does that per call into the workspace (~150 launches of a 5 us
    kernel per training step).  Here the images of parameters that live in optim.FlatBuffers are kept in their own tensors,
    the first use of an image packs it with one call (rvsr_conv2d_pack_weights / rvsr_dcn_pack_weights), and from then on
    ``FlatAdam.step`` re-packs ALL registered images with ONE launch (rvsr_pack_weights_batched) right after the update.
    Validity: an entry is used only while (a) the very same parameter object is alive, (b) its torch version counter is
    unchanged (load_state_dict, in-place edits under no_grad bump it) and (c) it was packed in the current epoch; everything
    that writes parameters behind torch's back (the flat Adam kernel, a broadcast into the flat buffer, a raw copy into it) must
    call ``repack()`` (re-pack now) or ``invalidate()`` (forget).  Foreign weights (not in FlatBuffers) take the per-call path.
    RVSR_PACK_CACHE=0 turns the cache off."""

    def __init__(self):
        self.entries = {}          # key -> [packed tensor, weakref(weight), version, epoch, desc (10 ints), weight.data_ptr()]
        self.slices = {}           # (id(weight), C1) -> [w_a, w_b, weakref(weight), version, epoch, data_ptr]
        self.epoch = 0
        self.table = None          # device copy of the descriptor table, rebuilt when entries were added or dropped
        self.enabled = os.environ.get('RVSR_PACK_CACHE', '1') != '0'
        self.stats = {'hits': 0, 'packs': 0, 'batched': 0}

    def invalidate(self):
        self.entries.clear()
        self.slices.clear()
        self.table = None
        self.epoch += 1

    @staticmethod
    def _version(t):
        """Version of the parameter a weight tensor stands for: its own, or its parent's for a cached input-channel slice."""
        parent = getattr(t, '_rvsr_parent', None)
        if parent is None:
            return t._version
        parent = parent()
        return -1 if parent is None else parent._version

    def split(self, weight, C1):
        """weight[:, :C1] and weight[:, C1:] as PERSISTENT contiguous tensors (conv_cat_bcast convolves the two halves of a concat
        conv separately): copied on first sight and after every optimizer step (repack), so that their packed images can be
        cached like those of whole parameters.  Falls back to fresh copies for weights outside FlatBuffers."""
        if not self.enabled or getattr(weight, '_rvsr_grad_home', None) is None:
            return weight[:, :C1].contiguous(), weight[:, C1:].contiguous()
        import weakref
        key = (id(weight), C1)
        e = self.slices.get(key)
        if e is not None and e[2]() is weight and e[3] == weight._version and e[4] == self.epoch and e[5] == weight.data_ptr():
            return e[0], e[1]
        if e is not None and e[2]() is weight and e[0].device == weight.device:
            w_a, w_b = e[0], e[1]
            with torch.no_grad():
                w_a.copy_(weight[:, :C1])
                w_b.copy_(weight[:, C1:])
        else:
            w_a, w_b = weight[:, :C1].detach().contiguous(), weight[:, C1:].detach().contiguous()
            w_a._rvsr_parent = w_b._rvsr_parent = weakref.ref(weight)
        self.slices[key] = [w_a, w_b, weakref.ref(weight), weight._version, self.epoch, weight.data_ptr()]
        return w_a, w_b

    def get(self, weight, kind, C_in, Co, k, w_mode, nbytes):
        """Packed image tensor for (weight, kind, geometry), or None when the weight is not cacheable."""
        if not self.enabled or _lib.get_gemm_mode() == 'f32':
            return None
        if getattr(weight, '_rvsr_grad_home', None) is None and getattr(weight, '_rvsr_parent', None) is None:
            return None
        key = (id(weight), kind, C_in, Co, k, w_mode)
        e = self.entries.get(key)
        if e is not None and e[1]() is weight and e[2] == self._version(weight) and e[3] == self.epoch and e[5] == weight.data_ptr():
            self.stats['hits'] += 1
            if _PACK_VERIFY:
                self._verify(e, weight, kind, C_in, Co, k, w_mode)
            return e[0]
        import weakref
        L = _lib.lib()
        buf = e[0] if e is not None and e[0].numel() >= nbytes and e[0].device == weight.device else \
            torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=weight.device)
        desc = (ctypes.c_longlong * 20)()   # (second descriptor, desc[10:]: unused since the fourth-generation DCN forward left the library)
        if kind == 'conv':
            got = L.rvsr_conv2d_pack_weights(_p(weight), C_in, Co, k, w_mode, _p(buf), buf.numel(), desc, _stream())
        else:
            got = L.rvsr_dcn_pack_weights(_p(weight), C_in, Co, _p(buf), buf.numel(), desc, _stream())
        if got == 0:
            return None
        self.stats['packs'] += 1
        self.entries[key] = [buf, weakref.ref(weight), self._version(weight), self.epoch, list(desc[:10]), weight.data_ptr()]
        self.table = None
        return buf

    def _verify(self, e, weight, kind, C_in, Co, k, w_mode):
        """functional._PACK_VERIFY = True (debug): on a cache hit, pack the weight again and compare -- catches writes that bypassed both torch's
        version counter and repack()/invalidate() (p.data.copy_, raw writes into FlatBuffers.param, EMA swaps)."""
        L = _lib.lib()
        tmp = torch.empty_like(e[0])
        if kind == 'conv':
            n = L.rvsr_conv2d_pack_weights(_p(weight), C_in, Co, k, w_mode, _p(tmp), tmp.numel(), None, _stream())
        else:
            n = L.rvsr_dcn_pack_weights(_p(weight), C_in, Co, _p(tmp), tmp.numel(), None, _stream())
        n = int(n)
        if not torch.equal(tmp[:n], e[0][:n]):
            raise RuntimeError('PackedWeights: stale bf16 weight image (the parameter was written without a version bump; call '
                               'realvsr_amd.functional.invalidate_weight_cache() after such writes)')

    def repack(self):
        """Re-pack every live image in one launch (the parameters were just updated in place) and start a new epoch."""
        self.epoch += 1
        if not self.enabled or not self.entries:
            return
        for k, e in list(self.slices.items()):     # refresh the persistent input-channel slices first: their images are packed below
            parent = e[2]()
            if parent is None or e[3] != parent._version or e[5] != parent.data_ptr():
                del self.slices[k]
                continue
            with torch.no_grad():
                e[0].copy_(parent[:, :k[1]])
                e[1].copy_(parent[:, k[1]:])
            e[4] = self.epoch
        dead = [k for k, e in self.entries.items() if e[1]() is None or e[2] != self._version(e[1]()) or e[5] != e[1]().data_ptr()]
        for k in dead:
            del self.entries[k]
            self.table = None
        if not self.entries:
            return
        by_dev = {}
        for e in self.entries.values():
            by_dev.setdefault(e[0].device, []).append(e)
        if self.table is None:
            self.table = {}
            for dev, es in by_dev.items():
                # PackDesc = two pointers + eight 32-bit fields (48 bytes)
                raw = torch.empty(len(es), 6, dtype=torch.int64)
                for i, e in enumerate(es):
                    d = e[4]
                    raw[i, 0], raw[i, 1] = d[0], d[1]
                    for j in range(4):
                        raw[i, 2 + j] = (d[2 + 2 * j] & 0xffffffff) | ((d[3 + 2 * j] & 0xffffffff) << 32)
                self.table[dev] = (raw.to(dev), len(es))
        for dev, es in by_dev.items():
            tab, n = self.table[dev]
            with torch.cuda.device(dev):
                _lib.check(_lib.lib().rvsr_pack_weights_batched(_p(tab), n, _stream()), 'pack_weights_batched')
            for e in es:
                e[3] = self.epoch
        self.stats['batched'] += 1


packed_weights = PackedWeights()


def invalidate_weight_cache():
    """Call after writing parameters behind torch's version counter (``p.data.copy_``, raw writes into ``FlatBuffers.param``, an EMA
    swap, a custom re-init after FlatAdam was built): forgets every cached bf16 weight image; the next use packs afresh.
    ``load_state_dict`` / in-place ops under ``no_grad`` bump the version and need nothing; ``FlatAdam.step`` re-packs by itself."""
    packed_weights.invalidate()


def _conv_fwd(L, x1, C1, x2, C2, xact, xact_slope, in_mode, Hs, Ws, weight, bias, residual, out1, Co1, out2, Co2, B, k, stride, w_mode,
              act, slope, ps, Hout, Wout, what, wparam=None):
    """rvsr_conv2d_forward with the weight image from the per-step cache when `wparam` (the nn.Parameter behind `weight`,
    default: none = per-call packing into the shared scratch) lives in FlatBuffers."""
    nbytes = L.rvsr_conv2d_forward_workspace_bytes(C1, C2, Co1 + Co2, k)
    if (_lib.fmt_f16fp8() and k == 3 and stride == 1 and w_mode == 0 and Co1 + Co2 > 32 and not xact and in_mode == 0 and Ws % 4 == 0
            and (C2 == 0 or C1 % 16 == 0) and ((getattr(x1, 'value', None) or 0) | (getattr(x2, 'value', None) or 0)) % 16 == 0):
        w_mode |= 4     # 'f16fp8' mode: this forward conv in the f16 + fp8 product format (weight image and kernel; _lib.set_gemm_mode)
    buf = packed_weights.get(wparam, 'conv', C1 + C2, Co1 + Co2, k, w_mode, nbytes) if wparam is not None else None
    if buf is not None:
        ws, w_mode = buf, w_mode | 2
    else:
        ws = _workspace(nbytes, torch.device('cuda', torch.cuda.current_device()))
    rc = L.rvsr_conv2d_forward(x1, C1, x2, C2, xact, xact_slope, in_mode, Hs, Ws, weight, bias, residual, out1, Co1, out2, Co2, B, k,
                               stride, w_mode, act, slope, ps, Hout, Wout, _p(ws), ws.numel(), _stream())
    if rc == 1 and act == ACT_MASK:     # RVSR_ERR_UNSUPPORTED: the fused gradient mask is an option of one kernel, the caller has a plan B
        return False
    _lib.check(rc, what)
    return True


# ------------------------------------------------------------------------------------------ conv
class _Conv2dFused(Function):
    """act(conv2d(cat(x1, x2), w) + b) [+ residual] [-> PixelShuffle(2)]"""

    @staticmethod
    def forward(ctx, x1, x2, weight, bias, residual, stride, act, slope, pixel_shuffle, sink=None, dep_sink=None, x_premask=None,
                grad_premasked=False):
        # x_premask = (act, slope) of the activation that PRODUCED x1, given when this conv is x1's only consumer: the data gradient then
        # leaves this node already multiplied by act'(x1) (in the dgrad kernel's epilogue where it has one, rvsr_conv2d_forward act = 3),
        # and the producer, called with grad_premasked=True, applies no mask in its own backward -- one read of x1 instead of a read of
        # the producer's output in both of its gradient kernels.  (Multiplying by act' is linear, so it commutes with autograd's sum over
        # consumers; the pairing is still only set up for single-consumer chains, archs/EDVR_arch.py.)
        if x_premask is not None and (x2 is not None or sink is not None or dep_sink is not None or pixel_shuffle or stride != 1):
            raise RuntimeError('conv2d: x_premask is for single-input stride-1 convs without a GradSink')
        if grad_premasked and (act == ACT_NONE or residual is not None):
            # the consumer multiplies by act'(this conv's OUTPUT): with a residual added (or no activation) the output is not an
            # activation output and the pairing would silently drop / misapply the derivative
            raise RuntimeError('conv2d: grad_premasked needs an activation and no residual on the producing conv')
        ctx.x_premask, ctx.grad_premasked = x_premask, bool(grad_premasked)
        if x2 is not None and (sink is not None or dep_sink is not None):
            # an owner that never closes drops what the depositors wrote: sinks are for single-input convs only
            raise RuntimeError('conv2d: a GradSink cannot be combined with a second (concatenated) input')
        _need_cuda(x1, x2, weight, bias, residual)
        ctx.sink = sink           # GradSink of x1: this conv is its OWNER (see GradSink)
        ctx.dep_sink = dep_sink   # GradSink of x1: this conv is a DEPOSITOR (single-input convs only)
        x1, x2, weight, bias, residual = _c(x1), _c(x2), _c(weight), _c(bias), _c(residual)
        B, C1, H, W = x1.shape
        C2 = 0 if x2 is None else x2.shape[1]
        Co, Cw, k, _ = weight.shape
        if Cw != C1 + C2:
            raise RuntimeError('conv2d: weight expects %d input channels, got %d' % (Cw, C1 + C2))
        pad = k // 2
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        if pixel_shuffle:
            out = x1.new_empty(B, Co // 4, 2 * Ho, 2 * Wo)
        else:
            out = x1.new_empty(B, Co, Ho, Wo)
        L = _lib.lib()
        _conv_fwd(L, _p(x1), C1, _p(x2), C2, None, 0.0, 0, H, W, _p(weight), _p(bias), _p(residual), _p(out), Co, None, 0, B,
                  k, stride, 0, act, slope, int(pixel_shuffle), Ho, Wo, 'conv2d_forward', wparam=weight)
        ctx.cfg = (stride, act, slope, bool(pixel_shuffle), C1, C2, H, W, Ho, Wo, k, bias is not None,
                   residual is not None)
        ctx.save_for_backward(x1, x2, weight, out if act != ACT_NONE else None)
        ctx.bias_p = bias   # only to find its gradient buffer (_pgrad)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        x1, x2, weight, act_out = ctx.saved_tensors
        stride, act, slope, ps, C1, C2, H, W, Ho, Wo, k, has_bias, has_res = ctx.cfg
        gout = gout.contiguous()
        B, Co = x1.shape[0], weight.shape[0]
        L = _lib.lib()
        gslope = 0.0 if act == ACT_RELU else slope
        if ctx.grad_premasked:
            act_out = None   # gout arrives multiplied by act'(out) already (the consumer's x_premask)
        need_x1, need_x2, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        gx1 = gx2 = gw = gb = None
        if ctx.sink is not None and not need_x1:
            # the owner always closes: what the depositors wrote is x1's gradient even if autograd wants none from this conv
            ctx.sink.close()
        if need_x1 or (x2 is not None and need_x2):
            # owner of a GradSink (single-input convs only): add this data gradient onto what the other consumers of x1 deposited
            dep = ctx.sink.close() if (ctx.sink is not None and x2 is None) else None
            deposit = ctx.dep_sink is not None and x2 is None and not ctx.dep_sink.closed
            if deposit and ctx.dep_sink.buf is not None:
                dep = ctx.dep_sink.buf            # a later depositor: add in place
            gx1 = dep if dep is not None else torch.empty_like(x1)
            gx2 = torch.empty_like(x2) if x2 is not None else None
            in_mode = 2 if ps else (1 if stride == 2 else 0)
            masked = False
            if ctx.x_premask is not None:
                pslope = 0.0 if ctx.x_premask[0] == ACT_RELU else ctx.x_premask[1]
                masked = _FUSE_GRAD_MASK and k == 3 and _conv_fwd(
                    L, _p(gout), Co, None, 0, _p(act_out), gslope, in_mode, gout.shape[2], gout.shape[3], _p(weight), None, _p(x1), _p(gx1),
                    C1, None, 0, B, k, 1, 1, ACT_MASK, pslope, 0, H, W, 'conv2d_backward_data', wparam=weight)
            if not masked:
                _conv_fwd(L, _p(gout), Co, None, 0, _p(act_out), gslope, in_mode, gout.shape[2], gout.shape[3], _p(weight), None,
                          _p(dep), _p(gx1), C1, _p(gx2), C2, B, k, 1, 1, ACT_NONE, 0.0, 0, H, W, 'conv2d_backward_data',
                          wparam=weight)
                if ctx.x_premask is not None:   # no fused epilogue for this frame / GEMM mode: the mask as a separate pass (the producer applies none)
                    gx1.mul_(torch.where(x1 > 0, 1.0, pslope))
            if deposit:   # the first depositor's output IS the sink's buffer (no zero fill); autograd gets no gradient from here
                ctx.dep_sink.buf = gx1
                gx1 = None
        if need_w or (has_bias and ctx.needs_input_grad[3]):
            gw = _pgrad(weight)
            gb = _pgrad(ctx.bias_p) if has_bias else None
            nbytes = L.rvsr_conv2d_wgrad_workspace_bytes(C1, C2, Co, B, k, stride, Ho, Wo)
            ws = _workspace(nbytes, x1.device)
            _lib.check(L.rvsr_conv2d_backward_weight(_p(x1), C1, _p(x2), C2, H, W, _p(gout), _p(act_out), gslope,
                                                     2 if ps else 0, gout.shape[2], gout.shape[3], _p(gw), _p(gb),
                                                     Co, B, k, stride, Ho, Wo, 0, _p(ws), ws.numel(), _stream()),
                       'conv2d_backward_weight')
        gres = gout if has_res else None
        return gx1, gx2, gw, gb, gres, None, None, None, None, None, None, None, None


class _ResBlockFused(Function):
    """x + conv2(relu(conv1(x))) as ONE autograd node (arch_util.ResidualBlock_noBN, reference
    codes/models/archs/arch_util.py:37-52).  Forward is the same two fused kernels as two conv2d()
    calls; the point is the backward: grad_x = grad_out + dgrad1(...) is produced by conv1's
    data-gradient kernel with grad_out as its fused residual, instead of a separate full-tensor
    add by autograd (3 passes over a B x 64 x H x W tensor per block)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        _need_cuda(x, w1, b1, w2, b2)
        x, w1, b1, w2, b2 = _c(x), _c(w1), _c(b1), _c(w2), _c(b2)
        B, C, H, W = x.shape
        if w1.shape != (C, C, 3, 3) or w2.shape != (C, C, 3, 3):
            raise RuntimeError('res_block: expected two %dx%dx3x3 convs' % (C, C))
        L = _lib.lib()
        h, out = torch.empty_like(x), torch.empty_like(x)
        _conv_fwd(L, _p(x), C, None, 0, None, 0.0, 0, H, W, _p(w1), _p(b1), None, _p(h), C, None, 0, B, 3, 1, 0, ACT_RELU,
                  0.0, 0, H, W, 'res_block conv1', wparam=w1)
        _conv_fwd(L, _p(h), C, None, 0, None, 0.0, 0, H, W, _p(w2), _p(b2), _p(x), _p(out), C, None, 0, B, 3, 1, 0, ACT_NONE,
                  0.0, 0, H, W, 'res_block conv2', wparam=w2)
        ctx.save_for_backward(x, h, w1, w2)
        ctx.has_bias = (b1 is not None, b2 is not None)
        ctx.bias_p = (b1, b2)   # only to find their gradient buffers (_pgrad)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        x, h, w1, w2 = ctx.saved_tensors
        gout = gout.contiguous()
        B, C, H, W = x.shape
        L = _lib.lib()
        need_x, need_w1, need_b1, need_w2, need_b2 = ctx.needs_input_grad
        gx = gw1 = gb1 = gw2 = gb2 = None
        nb = L.rvsr_conv2d_wgrad_workspace_bytes(C, 0, C, B, 3, 1, H, W)
        if need_w2 or need_b2:
            gw2 = _pgrad(w2)
            gb2 = _pgrad(ctx.bias_p[1]) if ctx.has_bias[1] else None
            ws = _workspace(nb, x.device)
            _lib.check(L.rvsr_conv2d_backward_weight(_p(h), C, None, 0, H, W, _p(gout), None, 0.0, 0, H, W, _p(gw2),
                                                     _p(gb2), C, B, 3, 1, H, W, 0, _p(ws), ws.numel(), _stream()),
                       'res_block wgrad2')
        if need_x or need_w1 or need_b1:
            # gradient w.r.t. conv1's output: conv2's data gradient times relu'(h).  The mask is applied in that kernel's epilogue when it
            # has one (8 x 64 tile; one extra read of h instead of a second read of h in BOTH consumers below), else by the consumers.
            gh = torch.empty_like(x)
            masked = _conv_fwd(L, _p(gout), C, None, 0, None, 0.0, 0, H, W, _p(w2), None, _p(h), _p(gh), C, None, 0, B, 3, 1, 1,
                               ACT_MASK, 0.0, 0, H, W, 'res_block dgrad2', wparam=w2) if _FUSE_GRAD_MASK else False
            if not masked:
                _conv_fwd(L, _p(gout), C, None, 0, None, 0.0, 0, H, W, _p(w2), None, None, _p(gh), C, None, 0, B, 3, 1, 1,
                          ACT_NONE, 0.0, 0, H, W, 'res_block dgrad2', wparam=w2)
            hmask = None if masked else _p(h)
            if need_w1 or need_b1:
                gw1 = _pgrad(w1)
                gb1 = _pgrad(ctx.bias_p[0]) if ctx.has_bias[0] else None
                ws = _workspace(nb, x.device)
                _lib.check(L.rvsr_conv2d_backward_weight(_p(x), C, None, 0, H, W, _p(gh), hmask, 0.0, 0, H, W,
                                                         _p(gw1), _p(gb1), C, B, 3, 1, H, W, 0, _p(ws), ws.numel(),
                                                         _stream()), 'res_block wgrad1')
            if need_x:
                gx = torch.empty_like(x)
                _conv_fwd(L, _p(gh), C, None, 0, hmask, 0.0, 0, H, W, _p(w1), None, _p(gout), _p(gx), C, None, 0, B, 3, 1, 1,
                          ACT_NONE, 0.0, 0, H, W, 'res_block dgrad1', wparam=w1)
        return gx, gw1, gb1, gw2, gb2


class _ConvCatBcast(Function):
    """act(conv3x3(cat([x, ref.repeat(N, 1, 1, 1)], 1), w) + b) for a frame-major x [N*B, C1, H, W] and ref [B, C2, H, W],
    WITHOUT the repeat and without convolving the same reference N times:

        conv(cat(x, repeat(ref))) = conv_a(x) + conv_b(ref),     w = [w_a | w_b] along the input channels.

    PCD_Align convolves cat([nbr, ref]) four times per frame (EDVR_arch.py:100, 109, 118, 127) and the reference
    features are those of the window's centre frame for all N frames (:297-303): conv_b runs on B frames instead of
    N*B (fwd, dgrad and wgrad: 2.0 -> 1.2 conv-equivalents each), the 2x-wide conv reads half the channels, and the
    N copies of ref are never materialised.  The glue is one in-place pass (add broadcast + activation) forward and
    one reduction over N backward; act' is taken from the saved output as everywhere else."""

    @staticmethod
    def forward(ctx, x, ref, weight, bias, N, act, slope, x_sink=None, x_owner=False, ref_sink=None, ref_block=0, grad_premasked=False):
        ctx.grad_premasked = bool(grad_premasked)   # the only consumer multiplies its data gradient by act'(out) (conv2d x_premask)
        _need_cuda(x, ref, weight, bias)
        x, ref, weight, bias = _c(x), _c(ref), _c(weight), _c(bias)
        # GradSinks (see GradSink): x_sink collects the gradient of x (this conv deposits, or owns it when x_owner);
        # ref_sink is the sink of the [N*B] tensor whose block `ref_block` IS ref: the ref gradient is added into that block
        ctx.sinks = (x_sink, bool(x_owner), ref_sink, int(ref_block))
        NB, C1, H, W = x.shape
        B, C2 = ref.shape[0], ref.shape[1]
        Co, Cw, k, _ = weight.shape
        if NB != N * B or Cw != C1 + C2 or k != 3 or tuple(ref.shape[2:]) != (H, W):
            raise RuntimeError('conv_cat_bcast: x %s, ref %s, weight %s, N %d do not fit' % (tuple(x.shape), tuple(ref.shape), tuple(weight.shape), N))
        w_a, w_b = packed_weights.split(weight, C1)   # persistent per-step copies when the weight lives in FlatBuffers
        out = x.new_empty(NB, Co, H, W)
        part = x.new_empty(B, Co, H, W)
        L = _lib.lib()
        _conv_fwd(L, _p(x), C1, None, 0, None, 0.0, 0, H, W, _p(w_a), _p(bias), None, _p(out), Co, None, 0, NB, 3, 1, 0,
                  ACT_NONE, 0.0, 0, H, W, 'conv_cat_bcast conv_a', wparam=w_a)
        _conv_fwd(L, _p(ref), C2, None, 0, None, 0.0, 0, H, W, _p(w_b), None, None, _p(part), Co, None, 0, B, 3, 1, 0,
                  ACT_NONE, 0.0, 0, H, W, 'conv_cat_bcast conv_b', wparam=w_b)
        _lib.check(L.rvsr_bcast_add_act(_p(out), _p(part), part.numel(), N, act, slope, _stream()), 'bcast_add_act')
        ctx.cfg = (N, act, slope, bias is not None)
        ctx.w_ab = (w_a, w_b)   # (python references: the cached slices carry the attribute the weight-image cache keys on)
        # the cached slices are PERSISTENT tensors that repack() / split() overwrite in place, behind save_for_backward's version
        # check: remember what they were copied from, backward refuses to run on newer weights
        ctx.w_tag = (weight._version, packed_weights.epoch if getattr(w_a, '_rvsr_parent', None) is not None else None)
        ctx.save_for_backward(x, ref, out if act != ACT_NONE else None)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        x, ref, act_out = ctx.saved_tensors
        w_a, w_b = ctx.w_ab
        if ctx.w_tag[1] is not None:
            parent = w_a._rvsr_parent()
            if parent is None or parent._version != ctx.w_tag[0] or packed_weights.epoch != ctx.w_tag[1]:
                raise RuntimeError('conv_cat_bcast: the weights were modified (optimizer step / in-place write) between this forward '
                                   'and its backward')
        N, act, slope, has_bias = ctx.cfg
        gout = gout.contiguous()
        NB, C1, H, W = x.shape
        B, C2 = ref.shape[0], ref.shape[1]
        Co = w_a.shape[0]
        L = _lib.lib()
        gslope = 0.0 if act == ACT_RELU else slope
        if ctx.grad_premasked:
            act_out = None
        gx = gref = gw = gb = None
        # gradient of the broadcast partial: sum over the N frames of gout * act'
        gpart = x.new_empty(B, Co, H, W)
        _lib.check(L.rvsr_bcast_reduce_act(_p(gout), _p(act_out), _p(gpart), gpart.numel(), N, gslope, _stream()), 'bcast_reduce_act')
        x_sink, x_owner, ref_sink, ref_block = ctx.sinks
        # (the reference gradient first: when this conv owns x's sink and ref is a block of x, the deposit must precede the close)
        if ctx.needs_input_grad[1]:
            blk = None
            if ref_sink is not None and not ref_sink.closed:
                full = ref_sink.buf
                if full is None:
                    full = ref_sink.buf = ref.new_zeros(ref_sink.shape)
                blk = full[ref_block * B:(ref_block + 1) * B]
            gref = blk if blk is not None else torch.empty_like(ref)
            _conv_fwd(L, _p(gpart), Co, None, 0, None, 0.0, 0, H, W, _p(w_b), None, _p(blk), _p(gref), C2, None, 0, B, 3, 1,
                      1, ACT_NONE, 0.0, 0, H, W, 'conv_cat_bcast dgrad_b', wparam=w_b)
            if blk is not None:
                gref = None  # deposited into the block of the full tensor's sink
        if ctx.needs_input_grad[0]:
            dep = None
            if x_sink is not None:
                dep = x_sink.close() if x_owner else x_sink.get(x)
            gx = dep if dep is not None else torch.empty_like(x)
            # a fresh sink buffer is all zeros: adding it as the residual is exact and keeps one code path
            _conv_fwd(L, _p(gout), Co, None, 0, _p(act_out), gslope, 0, H, W, _p(w_a), None, _p(dep), _p(gx), C1, None, 0,
                      NB, 3, 1, 1, ACT_NONE, 0.0, 0, H, W, 'conv_cat_bcast dgrad_a', wparam=w_a)
            if dep is not None and not x_owner:
                gx = None    # deposited: the owner returns the buffer
        if ctx.needs_input_grad[2] or (has_bias and ctx.needs_input_grad[3]):
            gw_a, gw_b = torch.empty_like(w_a), torch.empty_like(w_b)
            gb = w_a.new_empty(Co) if has_bias else None
            ws = _workspace(L.rvsr_conv2d_wgrad_workspace_bytes(C1, 0, Co, NB, 3, 1, H, W), x.device)
            _lib.check(L.rvsr_conv2d_backward_weight(_p(x), C1, None, 0, H, W, _p(gout), _p(act_out), gslope, 0, H, W, _p(gw_a),
                                                     _p(gb), Co, NB, 3, 1, H, W, 0, _p(ws), ws.numel(), _stream()),
                       'conv_cat_bcast wgrad_a')
            ws = _workspace(L.rvsr_conv2d_wgrad_workspace_bytes(C2, 0, Co, B, 3, 1, H, W), x.device)
            _lib.check(L.rvsr_conv2d_backward_weight(_p(ref), C2, None, 0, H, W, _p(gpart), None, 0.0, 0, H, W, _p(gw_b), None,
                                                     Co, B, 3, 1, H, W, 0, _p(ws), ws.numel(), _stream()), 'conv_cat_bcast wgrad_b')
            gw = torch.cat([gw_a, gw_b], 1)
        return gx, gref, gw, gb, None, None, None, None, None, None, None, None


def conv_cat_bcast(x, ref, conv, N, act=ACT_NONE, slope=0.1, x_sink=None, x_owner=False, ref_sink=None, ref_block=0, grad_premasked=False):
    """act(conv(cat([x, ref.repeat(N, 1, 1, 1)], 1))) for frame-major x [N*B, ...] and ref [B, ...] (3x3, stride 1)."""
    return _ConvCatBcast.apply(x, ref, conv.weight, conv.bias, int(N), act, float(slope), x_sink, x_owner, ref_sink, ref_block, grad_premasked)


def res_block(x, conv1, conv2):
    """x + conv2(relu(conv1(x))) with the identity add fused in both directions."""
    return _ResBlockFused.apply(x, conv1.weight, conv1.bias, conv2.weight, conv2.bias)


def conv2d(x, conv, act=ACT_NONE, slope=0.1, x2=None, residual=None, pixel_shuffle=False, sink=None, dep_sink=None, x_premask=None,
           grad_premasked=False):
    """Fused conv block driven by an ``nn.Conv2d`` parameter holder (weight, bias, stride).

    out = act(conv(cat(x, x2))) [+ residual]; with ``pixel_shuffle`` the activation commutes with
    the shuffle, so lrelu(pixel_shuffle(conv(x))) (EDVR_arch.py:311-312) is one kernel."""
    stride = conv.stride[0] if isinstance(conv.stride, tuple) else conv.stride
    if grad_premasked and residual is not None:
        raise RuntimeError('conv2d: grad_premasked cannot be combined with a residual (out + residual is not an activation output)')
    if residual is not None and act != ACT_NONE:
        # act'(.) is recovered from the saved activation output, so the residual is added outside
        out = _Conv2dFused.apply(x, x2, conv.weight, conv.bias, None, stride, act, slope, pixel_shuffle, sink, dep_sink, x_premask,
                                 grad_premasked)
        return out + residual
    return _Conv2dFused.apply(x, x2, conv.weight, conv.bias, residual, stride, act, slope, pixel_shuffle, sink, dep_sink, x_premask,
                              grad_premasked)


def grad_mask_fusable(H, W):
    """Whether the data-gradient kernel of a 3x3 stride-1 conv on H x W frames has the mask epilogue (the 8 x 64 tile: chosen when it
    wastes no more pixels than 16 x 32, conv2_kernels.hip launch_fwd5).  Where it has not, x_premask costs a separate pass."""
    px_n = ((H + 15) // 16 * 16) * ((W + 31) // 32 * 32)
    px_w = ((H + 7) // 8 * 8) * ((W + 63) // 64 * 64)
    return _FUSE_GRAD_MASK and W % 4 == 0 and px_w <= px_n


# ------------------------------------------------------------------------------------------ DCN
class DcnOffsetStats:
    """Per DCN layer: the sampled offset counters of its last backwards (components beyond 2.5 .. 11.5 px), brought to the host with a
    non-blocking copy + event, so that a later forward of the layer can pick its LDS tile halo (3 / 7 / 11 px) on the host and launch
    exactly one kernel.  The forward uses the counters recorded LAG = 3 OPTIMIZER STEPS back (`advance()`, called by FlatAdam.step; the
    layer's last backward of that step) and waits for that copy: the choice is a function of the data, never of host timing (a
    query-and-keep-the-old-decision would make the kernel choice, and with it the last bits of the forward, depend on how far the host
    happens to run ahead), and a host that is up to three steps ahead of the GPU -- which is what absorbs an 80 ms pause of Python's
    garbage collector, tools/cpu_launch_time.py -- never blocks on it, however many backwards per step a layer has (the per-frame PCD path
    has N).  Without an optimizer that calls advance() the lag is counted in backwards of the layer instead.  Offsets of a layer change
    slowly from step to step, and the choice affects speed and the last bits of rounding only (samples beyond the tile gather from global
    memory with the same rules) -- which also means that data-parallel ranks, whose offsets differ, may run different tile sizes: their
    forwards are equal to rounding, not bitwise.  Rule = the device-side rule of rvsr_launch_dcn_fwd3: 3 px while < 8 % of the components
    exceed 3.5 px, 7 px while < 1 % exceed 7.5 px, else 11 px (7 px above 64 output channels)."""
    LAG = 3

    def __init__(self):
        self.layers = {}     # id(weight) -> [weakref, ring of [pinned host counters, event, n_samples, tick], records so far, {tick: halo}]
        self.step = None     # optimizer steps seen (None: nobody advances -- ticks are the layer's own record count)

    def advance(self):
        self.step = 1 if self.step is None else self.step + 1

    def _tick(self, e):
        return self.step if self.step is not None else e[2]

    def record(self, weight, probe_dev, nsamples):
        import weakref
        e = self.layers.get(id(weight))
        if e is None or e[0]() is not weight:
            ring = [[torch.zeros(8, dtype=torch.int32).pin_memory(), torch.cuda.Event(), 0, -1] for _ in range(self.LAG + 1)]
            e = [weakref.ref(weight), ring, 0, {}]
            self.layers[id(weight)] = e
            for k in [k for k, v in self.layers.items() if v[0]() is None]:
                del self.layers[k]
        tick = self._tick(e)
        slot = e[1][tick % (self.LAG + 1)]
        slot[0].copy_(probe_dev, non_blocking=True)
        slot[1].record()
        slot[2] = int(nsamples)
        slot[3] = tick
        e[3].pop(tick, None)
        e[2] += 1

    def forward_halo(self, weight, Co):
        e = self.layers.get(id(weight))
        if e is None or e[0]() is not weight or e[2] == 0:
            return 0
        want = self._tick(e) - self.LAG
        # the record of `want`; while the ring fills (or after steps without a backward of this layer): the oldest one it holds
        live = [s for s in e[1] if s[3] >= 0]
        older = [s for s in live if s[3] <= want]
        slot = max(older, key=lambda s: s[3]) if older else min(live, key=lambda s: s[3])
        tick = slot[3]
        if tick not in e[3]:
            c, ev, n = slot[0], slot[1], slot[2]
            ev.synchronize()
            if n == 0:
                halo = 0
            elif int(c[1]) * 100 < 8 * n:
                halo = 3
            elif int(c[3]) * 100 < n or Co > 64:
                halo = 7
            else:
                halo = 11
            e[3] = {tick: halo}
        return e[3][tick]


dcn_offset_stats = DcnOffsetStats()


# ---- the general path of the deformable operator (include/realvsr_hip.h section 1c, csrc/dcn_generic.hip): any kernel size, anisotropic
# stride / padding / dilation, groups, any channels per deformable group, f32 / f64 / f16.  The fused kernels take the calls they cover.
_GENERIC_DTYPES = {torch.float32: 0, torch.float64: 1, torch.float16: 2}


def _pair2(v):
    from torch.nn.modules.utils import _pair
    a, b = _pair(v)
    return int(a), int(b)


def _dcn_fused_ok(input, weight, stride, padding, dilation, groups, dg):
    """Whether the fused f32 kernels cover this call: 3 x 3, isotropic geometry, one group, channels per deformable group a multiple or a
    divisor of 8 (dcn_kernels.hip fill_geom)."""
    (sh, sw), (ph, pw), (dh, dw) = _pair2(stride), _pair2(padding), _pair2(dilation)
    C = input.shape[1]
    if input.dtype != torch.float32 or tuple(weight.shape[2:]) != (3, 3) or sh != sw or ph != pw or dh != dw or groups != 1:
        return False
    if dg <= 0 or C % dg:
        return False
    cpg = C // dg
    return cpg % 8 == 0 or 8 % cpg == 0


def _generic_geom(input, weight, stride, padding, dilation):
    (sh, sw), (ph, pw), (dh, dw) = _pair2(stride), _pair2(padding), _pair2(dilation)
    B, C, H, W = input.shape
    Co, _, kh, kw = weight.shape
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    return (B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw), Ho, Wo


def _generic_check(input, offset, mask, weight, bias, groups, dg, Ho, Wo):
    dt = _GENERIC_DTYPES.get(input.dtype)
    if dt is None:
        raise TypeError('deformable convolution: element type %s (f32, f64 and f16 are implemented)' % input.dtype)
    B, C = input.shape[:2]
    Co, cg, kh, kw = weight.shape
    for name, t in (('offset', offset), ('mask', mask), ('weight', weight), ('bias', bias)):
        if t is not None and t.dtype != input.dtype:
            raise TypeError('deformable convolution: %s is %s, input is %s' % (name, t.dtype, input.dtype))
    if C % groups or Co % groups or cg * groups != C:
        raise RuntimeError('deformable convolution: channels (%d -> %d, weight %s) not divisible into %d groups' % (C, Co, tuple(weight.shape), groups))
    if dg <= 0 or C % dg:
        raise RuntimeError('deformable convolution: %d input channels not divisible into %d deformable groups' % (C, dg))
    if Ho <= 0 or Wo <= 0:
        raise ValueError('convolution input is too small (output would be %dx%d)' % (Ho, Wo))
    if tuple(offset.shape) != (B, 2 * dg * kh * kw, Ho, Wo):
        raise RuntimeError('deformable convolution: offset has shape %s, expected %s' % (tuple(offset.shape), (B, 2 * dg * kh * kw, Ho, Wo)))
    if mask is not None and tuple(mask.shape) != (B, dg * kh * kw, Ho, Wo):
        raise RuntimeError('deformable convolution: mask has shape %s, expected %s' % (tuple(mask.shape), (B, dg * kh * kw, Ho, Wo)))
    return dt


def _generic_dcn_forward(input, offset, mask, weight, bias, stride, padding, dilation, groups, dg):
    geo, Ho, Wo = _generic_geom(input, weight, stride, padding, dilation)
    dt = _generic_check(input, offset, mask, weight, bias, groups, dg, Ho, Wo)
    input, offset, weight = input.contiguous(), offset.contiguous(), weight.contiguous()
    mask = None if mask is None else mask.contiguous()
    bias = None if bias is None else bias.contiguous()
    out = input.new_empty(input.shape[0], weight.shape[0], Ho, Wo)
    L = _lib.lib()
    ws = _workspace(L.rvsr_deform_conv_generic_forward_workspace_bytes(dt, *geo[1:4], *geo[5:]), input.device)
    _lib.check(L.rvsr_deform_conv_generic_forward(dt, _p(input), _p(weight), _p(bias), _p(offset), _p(mask), _p(out), *geo, groups, dg,
                                                  _p(ws), ws.numel(), _stream()), 'deform_conv_generic_forward')
    return out


def _generic_dcn_backward(input, offset, mask, weight, grad_output, stride, padding, dilation, groups, dg, need_input, need_weight, with_bias):
    geo, Ho, Wo = _generic_geom(input, weight, stride, padding, dilation)
    dt = _GENERIC_DTYPES[input.dtype]
    input, offset, weight, grad_output = input.contiguous(), offset.contiguous(), weight.contiguous(), grad_output.contiguous()
    mask = None if mask is None else mask.contiguous()
    gx = torch.zeros_like(input) if need_input else None
    goff = torch.empty_like(offset) if need_input else None
    gmask = torch.empty_like(mask) if (need_input and mask is not None) else None
    gw = torch.zeros_like(weight) if need_weight else None
    gb = weight.new_zeros(weight.shape[0]) if with_bias else None
    L = _lib.lib()
    ws = _workspace(L.rvsr_deform_conv_generic_workspace_bytes(dt, *geo[1:4], *geo[5:]), input.device)
    _lib.check(L.rvsr_deform_conv_generic_backward(dt, _p(input), _p(weight), _p(offset), _p(mask), _p(grad_output), _p(gx), _p(goff), _p(gmask),
                                                   _p(gw), _p(gb), *geo, groups, dg, _p(ws), ws.numel(), _stream()), 'deform_conv_generic_backward')
    return gx, goff, gmask, gw, gb


class ModulatedDeformConvFunction(Function):
    """Same signature and semantics as the reference's autograd Function
    (codes/models/archs/dcn/deform_conv.py:97-153)."""

    @staticmethod
    def forward(ctx, input, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1,
                deformable_groups=1):
        if not input.is_cuda:
            raise NotImplementedError
        _need_cuda(input, offset, mask, weight, bias, any_float=True)
        if not input.is_contiguous():
            raise RuntimeError('input tensor has to be contiguous')      # deform_conv_cuda.cpp:497
        if not weight.is_contiguous():
            raise RuntimeError('weight tensor has to be contiguous')     # deform_conv_cuda.cpp:498
        offset, mask = offset.contiguous(), mask.contiguous()
        ctx.cfg = (stride, padding, dilation, groups, deformable_groups, bias is not None)
        ctx.generic = not _dcn_fused_ok(input, weight, stride, padding, dilation, groups, deformable_groups)
        if ctx.generic:   # any kernel size / anisotropic geometry / groups / f64, f16: the general path (realvsr_hip.h section 1c)
            ctx.save_for_backward(input, offset, mask, weight, bias)
            return _generic_dcn_forward(input, offset, mask, weight, bias, stride, padding, dilation, groups, deformable_groups)
        stride, padding, dilation = _pair2(stride)[0], _pair2(padding)[0], _pair2(dilation)[0]   # (isotropic here; pairs are accepted)
        ctx.cfg = (stride, padding, dilation, groups, deformable_groups, bias is not None)
        B, C, H, W = input.shape
        Co, _, kh, kw = weight.shape
        Ho = (H + 2 * padding - (dilation * (kh - 1) + 1)) // stride + 1
        Wo = (W + 2 * padding - (dilation * (kw - 1) + 1)) // stride + 1
        output = input.new_empty(B, Co, Ho, Wo)
        L = _lib.lib()
        ws = _workspace(L.rvsr_modulated_deform_conv_forward_workspace_bytes(C, Co), input.device)
        _lib.check(L.rvsr_modulated_deform_conv_forward(
            _p(input), _p(weight), _p(bias), _p(offset), _p(mask), _p(output), B, C, H, W, Co, kh, kw, stride, stride,
            padding, padding, dilation, dilation, groups, deformable_groups, int(bias is not None), _p(ws), ws.numel(),
            _stream()),
            'modulated_deform_conv_forward')
        ctx.save_for_backward(input, offset, mask, weight, bias)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        if not grad_output.is_cuda:
            raise NotImplementedError
        input, offset, mask, weight, bias = ctx.saved_tensors
        stride, padding, dilation, groups, dg, with_bias = ctx.cfg
        if ctx.generic:
            gx, goff, gmask, gw, gb = _generic_dcn_backward(input, offset, mask, weight, grad_output, stride, padding, dilation, groups, dg, True, True,
                                                            with_bias)
            return gx, goff, gmask, gw, gb, None, None, None, None, None
        grad_output = grad_output.contiguous()
        B, C, H, W = input.shape
        Co, _, kh, kw = weight.shape
        grad_input = torch.zeros_like(input)
        grad_offset = torch.empty_like(offset)
        grad_mask = torch.empty_like(mask)
        grad_weight = torch.zeros_like(weight)
        grad_bias = torch.zeros_like(bias) if with_bias else None
        L = _lib.lib()
        nbytes = L.rvsr_modulated_deform_conv_backward_workspace_bytes(B, C, H, W, Co, stride, padding, dilation)
        ws = _workspace(nbytes, input.device)
        _lib.check(L.rvsr_modulated_deform_conv_backward(
            _p(input), _p(weight), _p(bias), _p(offset), _p(mask), _p(grad_input), _p(grad_weight), _p(grad_bias),
            _p(grad_offset), _p(grad_mask), _p(grad_output), B, C, H, W, Co, kh, kw, stride, stride, padding, padding,
            dilation, dilation, groups, dg, int(with_bias), _p(ws), ws.numel(), _stream()),
            'modulated_deform_conv_backward')
        return grad_input, grad_offset, grad_mask, grad_weight, grad_bias, None, None, None, None, None


def modulated_deform_conv(input, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1):
    """The reference's functional entry point (deform_conv.py:99-100 binds it to ModulatedDeformConvFunction.apply).  groups > 1
    (deform_conv_cuda.cpp:539-561: one im2col over all channels, then one GEMM per group on that group's slice of the columns) is
    composed from `groups` calls of the groups == 1 operator on channel slices -- possible whenever a group's channels see whole
    deformable groups (deformable_groups % groups == 0) or lie inside one (groups % deformable_groups == 0)."""
    if groups == 1:
        return ModulatedDeformConvFunction.apply(input, offset, mask, weight, bias, stride, padding, dilation, 1, deformable_groups)
    if not input.is_cuda:
        raise NotImplementedError
    C, Co, dg = input.shape[1], weight.shape[0], deformable_groups
    slices_fused = (C % groups == 0 and (dg % groups == 0 or groups % dg == 0) and
                    _dcn_fused_ok(input[:, :C // groups], weight, stride, padding, dilation, 1, max(dg // groups, 1)))
    if not slices_fused:   # (the general path takes groups as the reference does: one GEMM per group on the shared columns)
        return ModulatedDeformConvFunction.apply(input, offset, mask, weight, bias, stride, padding, dilation, groups, deformable_groups)
    K = weight.shape[2] * weight.shape[3]
    if C % groups or Co % groups or weight.shape[1] * groups != C:
        raise RuntimeError('modulated_deform_conv: channels (%d -> %d) not divisible into %d groups' % (C, Co, groups))
    if dg % groups and groups % dg:
        raise RuntimeError('modulated_deform_conv: groups %d / deformable_groups %d: a group would see part of a deformable group; '
                           'not implemented on the HIP path' % (groups, dg))
    cg, og = C // groups, Co // groups
    outs = []
    for g in range(groups):
        if dg % groups == 0:
            d0, dn = g * (dg // groups), dg // groups
        else:
            d0, dn = g // (groups // dg), 1
        outs.append(ModulatedDeformConvFunction.apply(
            input[:, g * cg:(g + 1) * cg].contiguous(), offset[:, 2 * K * d0:2 * K * (d0 + dn)], mask[:, K * d0:K * (d0 + dn)],
            weight[g * og:(g + 1) * og].contiguous(), None if bias is None else bias[g * og:(g + 1) * og], stride, padding, dilation, 1, dn))
    return torch.cat(outs, 1)


class DeformConvFunction(Function):
    """DCNv1, same signature and semantics as the reference's autograd Function (codes/models/archs/dcn/deform_conv.py:15-95):
    (input, offset, weight, stride, padding, dilation, groups, deformable_groups, im2col_step); stride / padding / dilation are
    ints or pairs.  Runs the modulated kernels on a mask of ones (include/realvsr_hip.h section 1b)."""

    @staticmethod
    def forward(ctx, input, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1, im2col_step=64):
        from torch.nn.modules.utils import _pair
        if input is not None and input.dim() != 4:
            raise ValueError('Expected 4D tensor as input, got {}D tensor instead.'.format(input.dim()))
        ctx.stride, ctx.padding, ctx.dilation = _pair(stride), _pair(padding), _pair(dilation)
        ctx.groups, ctx.deformable_groups, ctx.im2col_step = groups, deformable_groups, im2col_step
        out_size = DeformConvFunction._output_size(input, weight, ctx.padding, ctx.dilation, ctx.stride)
        if not input.is_cuda:
            raise NotImplementedError
        _need_cuda(input, offset, weight, any_float=True)
        cur = min(im2col_step, input.shape[0])
        assert (input.shape[0] % cur) == 0, 'im2col step must divide batchsize'
        input, offset, weight = input.contiguous(), offset.contiguous(), weight.contiguous()   # deform_conv_cuda.cpp:170-172
        if offset.shape[0] != input.shape[0]:
            raise RuntimeError('invalid batch size of offset')                               # deform_conv_cuda.cpp:193
        ctx.save_for_backward(input, offset, weight)
        ctx.generic = not _dcn_fused_ok(input, weight, ctx.stride, ctx.padding, ctx.dilation, groups, deformable_groups)
        if ctx.generic:   # the general path (realvsr_hip.h section 1c): mask == NULL is DCNv1
            return _generic_dcn_forward(input, offset, None, weight, None, ctx.stride, ctx.padding, ctx.dilation, groups, deformable_groups)
        output = input.new_empty(out_size)
        DeformConvFunction._call('rvsr_deform_conv_forward', ctx, input, weight, cur, [input, weight, offset, output], [])
        return output

    @staticmethod
    def _call(name, ctx, input, weight, cur, tensors, extra):
        L = _lib.lib()
        B, C, H, W = input.shape
        Co, _, kh, kw = weight.shape
        (sh, sw), (ph, pw), (dh, dw) = ctx.stride, ctx.padding, ctx.dilation
        n = L.rvsr_deform_conv_workspace_bytes(B, C, H, W, Co, kw, kh, sw, pw, dw, ctx.deformable_groups)
        ws = _workspace(n, input.device)
        _lib.check(getattr(L, name)(*[_p(t) for t in tensors], B, C, H, W, Co, kw, kh, sw, sh, pw, ph, dw, dh, ctx.groups,
                                    ctx.deformable_groups, *extra, cur, _p(ws), ws.numel(), _stream()), name[5:])

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        input, offset, weight = ctx.saved_tensors
        grad_input = grad_offset = grad_weight = None
        if not grad_output.is_cuda:
            raise NotImplementedError
        cur = min(ctx.im2col_step, input.shape[0])
        grad_output = grad_output.contiguous()
        if ctx.generic:
            gx, goff, _, gw, _ = _generic_dcn_backward(input, offset, None, weight, grad_output, ctx.stride, ctx.padding, ctx.dilation, ctx.groups,
                                                       ctx.deformable_groups, ctx.needs_input_grad[0] or ctx.needs_input_grad[1],
                                                       ctx.needs_input_grad[2], False)
            return gx, goff, gw, None, None, None, None, None, None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            grad_input, grad_offset = torch.zeros_like(input), torch.zeros_like(offset)
            DeformConvFunction._call('rvsr_deform_conv_backward_input', ctx, input, weight, cur,
                                     [input, offset, grad_output, grad_input, grad_offset, weight], [])
        if ctx.needs_input_grad[2]:
            grad_weight = torch.zeros_like(weight)
            DeformConvFunction._call('rvsr_deform_conv_backward_parameters', ctx, input, weight, cur,
                                     [input, offset, grad_output, grad_weight], [1.0])
        return grad_input, grad_offset, grad_weight, None, None, None, None, None, None

    @staticmethod
    def _output_size(input, weight, padding, dilation, stride):
        size = (input.size(0), weight.size(0))
        for d in range(input.dim() - 2):
            kernel = dilation[d] * (weight.size(d + 2) - 1) + 1
            size += ((input.size(d + 2) + 2 * padding[d] - kernel) // stride[d] + 1,)
        if not all(s > 0 for s in size):
            raise ValueError('convolution input is too small (output would be {})'.format('x'.join(map(str, size))))
        return size


deform_conv = DeformConvFunction.apply


class _DcnPackFused(Function):
    """DCN fed directly by the raw conv_offset_mask output (chunk/cat/sigmoid fused)."""

    @staticmethod
    def forward(ctx, x, om, weight, bias, stride, padding, dilation, dg, act, slope, sink=None):
        _need_cuda(x, om, weight, bias)
        x, om, weight, bias = _c(x), _c(om), _c(weight), _c(bias)
        ctx.sink = sink   # GradSink of x: the DCN is a depositor (its backward accumulates atomically into the buffer)
        B, C, H, W = x.shape
        Co = weight.shape[0]
        if weight.shape[2:] != (3, 3):
            raise RuntimeError('fused DCN pack supports 3x3 kernels')
        Ho = (H + 2 * padding - (dilation * 2 + 1)) // stride + 1
        Wo = (W + 2 * padding - (dilation * 2 + 1)) // stride + 1
        if tuple(om.shape) != (B, 27 * dg, Ho, Wo):
            raise RuntimeError('conv_offset_mask output has shape %s, expected %s' % (tuple(om.shape), (B, 27 * dg, Ho, Wo)))
        out = x.new_empty(B, Co, Ho, Wo)
        L = _lib.lib()
        nbytes = L.rvsr_modulated_deform_conv_forward_workspace_bytes(C, Co)
        ws = packed_weights.get(weight, 'dcn', C, Co, 3, 0, nbytes)          # per-step weight image (act bit 8: already packed)
        flags = 0x100 if ws is not None else 0
        if ws is None:
            ws = _workspace(nbytes, x.device)
        # Halo of the forward's LDS tile.  Training: chosen HERE from the offset counters the previous backward of this layer left
        # (copied to the host asynchronously, see DcnOffsetStats) -- one kernel launch, no probe pass in the forward.  Inference (no
        # gradient wanted): a probe pass + selection on the device.
        training = any(ctx.needs_input_grad)
        probe = None
        if training:
            flags |= dcn_offset_stats.forward_halo(weight, Co) << 10
        elif stride == 1 and dilation == 1:
            probe = torch.zeros(8, dtype=torch.int32, device=x.device)
        _lib.check(L.rvsr_dcn_pack_forward(_p(x), _p(weight), _p(bias), _p(om), _p(out), B, C, H, W, Co, stride,
                                           padding, dilation, dg, act | flags, slope, _p(probe), _p(ws), ws.numel(), _stream()),
                   'dcn_pack_forward')
        ctx.cfg = (stride, padding, dilation, dg, act, slope, bias is not None)
        ctx.save_for_backward(x, om, weight, out if act != ACT_NONE else None)
        ctx.bias_p = bias   # only to find its gradient buffer (_pgrad)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        x, om, weight, act_out = ctx.saved_tensors
        stride, padding, dilation, dg, act, slope, has_bias = ctx.cfg
        gout = gout.contiguous()
        B, C, H, W = x.shape
        Co = weight.shape[0]
        dep = ctx.sink.get(x) if ctx.sink is not None else None
        gx = dep if dep is not None else torch.zeros_like(x)
        gom = torch.empty_like(om)
        gw = _pgrad(weight, zero=True)   # (the DCN backward accumulates into grad_weight / grad_bias)
        gb = _pgrad(ctx.bias_p, zero=True) if has_bias else None
        L = _lib.lib()
        nbytes = L.rvsr_modulated_deform_conv_backward_workspace_bytes(B, C, H, W, Co, stride, padding, dilation)
        ws = _workspace(nbytes, x.device)
        gslope = 0.0 if act == ACT_RELU else slope
        # offset counters of this layer: the backward selects its window halo from them on the device, and a copy travels to the host
        # (asynchronously; the next forward of this layer waits for it) for the forward of the next step
        probe = None
        if stride == 1 and dilation == 1 and C % (8 * dg) == 0:
            probe = torch.zeros(8, dtype=torch.int32, device=x.device)
            _lib.check(L.rvsr_dcn_offset_probe(_p(om), B, om.shape[2], om.shape[3], dg, _p(probe), _stream()), 'dcn_offset_probe')
            dcn_offset_stats.record(weight, probe, B * dg * 18 * ((om.shape[2] + 15) // 16) * om.shape[3])
        _lib.check(L.rvsr_dcn_pack_backward(_p(x), _p(weight), _p(om), _p(gout), _p(act_out), gslope, _p(gx), _p(gw),
                                            _p(gb), _p(gom), B, C, H, W, Co, stride, padding, dilation, dg, _p(probe), _p(ws),
                                            ws.numel(), _stream()), 'dcn_pack_backward')
        return (None if dep is not None else gx), gom, gw, gb, None, None, None, None, None, None, None


def dcn_pack(x, om, weight, bias, stride, padding, dilation, deformable_groups, act=ACT_NONE, slope=0.1, sink=None):
    return _DcnPackFused.apply(x, om, weight, bias, stride, padding, dilation, deformable_groups, act, slope, sink)


# ------------------------------------------------------------------------------------------ resampling / fusion
class _UpsampleBilinear(Function):
    @staticmethod
    def forward(ctx, x, factor, scale):
        _need_cuda(x)
        x = x.contiguous()
        B, C, H, W = x.shape
        out = x.new_empty(B, C, H * factor, W * factor)
        _lib.check(_lib.lib().rvsr_upsample_bilinear_forward(_p(x), _p(out), B * C, H, W, factor, scale, _stream()),
                   'upsample_bilinear_forward')
        ctx.cfg = (factor, scale, B, C, H, W)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        factor, scale, B, C, H, W = ctx.cfg
        gout = gout.contiguous()
        gin = gout.new_empty(B, C, H, W)
        _lib.check(_lib.lib().rvsr_upsample_bilinear_backward(_p(gout), _p(gin), B * C, H, W, factor, scale, _stream()),
                   'upsample_bilinear_backward')
        return gin, None, None


def upsample_bilinear(x, factor=2, scale=1.0):
    """scale * F.interpolate(x, scale_factor=factor, mode='bilinear', align_corners=False)"""
    return _UpsampleBilinear.apply(x, factor, float(scale))


class _MaxAvgPool(Function):
    @staticmethod
    def forward(ctx, x):
        _need_cuda(x)
        x = x.contiguous()
        B, C, H, W = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        out = x.new_empty(B, 2 * C, Ho, Wo)
        arg = torch.empty(B, C, Ho, Wo, dtype=torch.uint8, device=x.device)
        _lib.check(_lib.lib().rvsr_maxavgpool_forward(_p(x), _p(out), _p(arg), B, C, H, W, _stream()), 'maxavgpool_forward')
        ctx.cfg = (B, C, H, W)
        ctx.save_for_backward(arg)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        (arg,) = ctx.saved_tensors
        B, C, H, W = ctx.cfg
        gout = gout.contiguous()
        gin = gout.new_empty(B, C, H, W)
        _lib.check(_lib.lib().rvsr_maxavgpool_backward(_p(gout), _p(arg), _p(gin), B, C, H, W, _stream()),
                   'maxavgpool_backward')
        return gin


def maxavgpool(x):
    """cat([max_pool2d(x, 3, 2, 1), avg_pool2d(x, 3, 2, 1)], 1)"""
    return _MaxAvgPool.apply(x)


class _TSATemporal(Function):
    @staticmethod
    def forward(ctx, emb, emb_ref, aligned):
        _need_cuda(emb, emb_ref, aligned)
        emb, emb_ref, aligned = emb.contiguous(), emb_ref.contiguous(), aligned.contiguous()
        B, N, C, H, W = aligned.shape
        mod = aligned.new_empty(B, N * C, H, W)
        prob = aligned.new_empty(B, N, H, W)
        _lib.check(_lib.lib().rvsr_tsa_temporal_forward(_p(emb), _p(emb_ref), _p(aligned), _p(mod), _p(prob), B, N, C,
                                                        H, W, 0, _stream()), 'tsa_temporal_forward')
        ctx.save_for_backward(emb, emb_ref, aligned, prob)
        return mod

    @staticmethod
    @once_differentiable
    def backward(ctx, gmod):
        emb, emb_ref, aligned, prob = ctx.saved_tensors
        B, N, C, H, W = aligned.shape
        gmod = gmod.contiguous()
        galigned, gemb, gemb_ref = torch.empty_like(aligned), torch.empty_like(emb), torch.empty_like(emb_ref)
        _lib.check(_lib.lib().rvsr_tsa_temporal_backward(_p(gmod), _p(emb), _p(emb_ref), _p(aligned), _p(prob),
                                                         _p(galigned), _p(gemb), _p(gemb_ref), B, N, C, H, W, 0,
                                                         _stream()), 'tsa_temporal_backward')
        return gemb, gemb_ref, galigned


def tsa_temporal(emb, emb_ref, aligned):
    """aligned * sigmoid(sum_c emb * emb_ref), returned as (B, N*C, H, W)"""
    return _TSATemporal.apply(emb, emb_ref, aligned)


class _TSATemporalBlock(Function):
    """The temporal-attention front of TSA_Fusion (EDVR_arch.py:171-181) as ONE autograd node on the frame-major batch the
    alignment stage produces: emb = tAtt_1(aligned), emb_ref = tAtt_2(aligned[center]), mod[b, n*C + c] =
    aligned[n, b, c] * sigmoid(<emb[n, b], emb_ref[b]>).  `aligned` is [N, B, C, H, W]; no stack / transposing copy.
    Backward writes the gradient of `aligned` exactly once: the modulation gradient, then tAtt_1's data gradient with that
    buffer as its fused residual, then tAtt_2's into the centre block (three full-size autograd adds, a zero fill and two
    590 MB transposes per step at config 2 otherwise)."""

    @staticmethod
    def forward(ctx, aligned, center, w1, b1, w2, b2):
        _need_cuda(aligned, w1, b1, w2, b2)
        aligned, w1, b1, w2, b2 = _c(aligned), _c(w1), _c(b1), _c(w2), _c(b2)
        N, B, C, H, W = aligned.shape
        if tuple(w1.shape) != (C, C, 3, 3) or tuple(w2.shape) != (C, C, 3, 3):
            raise RuntimeError('tsa_temporal_block: expected two %dx%dx3x3 convs' % (C, C))
        L = _lib.lib()
        emb = aligned.new_empty(N, B, C, H, W)
        emb_ref = aligned.new_empty(B, C, H, W)
        cen = aligned[center]
        _conv_fwd(L, _p(aligned), C, None, 0, None, 0.0, 0, H, W, _p(w1), _p(b1), None, _p(emb), C, None, 0, N * B, 3, 1, 0,
                  ACT_NONE, 0.0, 0, H, W, 'tsa tAtt_1', wparam=w1)
        _conv_fwd(L, _p(cen), C, None, 0, None, 0.0, 0, H, W, _p(w2), _p(b2), None, _p(emb_ref), C, None, 0, B, 3, 1, 0,
                  ACT_NONE, 0.0, 0, H, W, 'tsa tAtt_2', wparam=w2)
        mod = aligned.new_empty(B, N * C, H, W)
        prob = aligned.new_empty(B, N, H, W)
        _lib.check(L.rvsr_tsa_temporal_forward(_p(emb), _p(emb_ref), _p(aligned), _p(mod), _p(prob), B, N, C, H, W, 1, _stream()),
                   'tsa_temporal_forward')
        ctx.center = center
        ctx.has_bias = (b1 is not None, b2 is not None)
        ctx.bias_p = (b1, b2)   # only to find their gradient buffers (_pgrad)
        ctx.save_for_backward(aligned, emb, emb_ref, prob, w1, w2)
        return mod

    @staticmethod
    @once_differentiable
    def backward(ctx, gmod):
        aligned, emb, emb_ref, prob, w1, w2 = ctx.saved_tensors
        N, B, C, H, W = aligned.shape
        center = ctx.center
        gmod = gmod.contiguous()
        L = _lib.lib()
        galigned, gemb, gemb_ref = torch.empty_like(aligned), torch.empty_like(emb), torch.empty_like(emb_ref)
        _lib.check(L.rvsr_tsa_temporal_backward(_p(gmod), _p(emb), _p(emb_ref), _p(aligned), _p(prob), _p(galigned), _p(gemb),
                                                _p(gemb_ref), B, N, C, H, W, 1, _stream()), 'tsa_temporal_backward')
        gw1 = gb1 = gw2 = gb2 = None
        cen = aligned[center]
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            gw1 = _pgrad(w1)
            gb1 = _pgrad(ctx.bias_p[0]) if ctx.has_bias[0] else None
            ws = _workspace(L.rvsr_conv2d_wgrad_workspace_bytes(C, 0, C, N * B, 3, 1, H, W), aligned.device)
            _lib.check(L.rvsr_conv2d_backward_weight(_p(aligned), C, None, 0, H, W, _p(gemb), None, 0.0, 0, H, W, _p(gw1), _p(gb1),
                                                     C, N * B, 3, 1, H, W, 0, _p(ws), ws.numel(), _stream()), 'tsa wgrad tAtt_1')
        if ctx.needs_input_grad[4] or ctx.needs_input_grad[5]:
            gw2 = _pgrad(w2)
            gb2 = _pgrad(ctx.bias_p[1]) if ctx.has_bias[1] else None
            ws = _workspace(L.rvsr_conv2d_wgrad_workspace_bytes(C, 0, C, B, 3, 1, H, W), aligned.device)
            _lib.check(L.rvsr_conv2d_backward_weight(_p(cen), C, None, 0, H, W, _p(gemb_ref), None, 0.0, 0, H, W, _p(gw2), _p(gb2),
                                                     C, B, 3, 1, H, W, 0, _p(ws), ws.numel(), _stream()), 'tsa wgrad tAtt_2')
        if ctx.needs_input_grad[0]:
            # galigned += dgrad(tAtt_1)(gemb), in place: the kernel's fused residual is its own output buffer (every lane reads
            # the residual values of exactly the addresses it then stores)
            _conv_fwd(L, _p(gemb), C, None, 0, None, 0.0, 0, H, W, _p(w1), None, _p(galigned), _p(galigned), C, None, 0,
                      N * B, 3, 1, 1, ACT_NONE, 0.0, 0, H, W, 'tsa dgrad tAtt_1', wparam=w1)
            gcen = galigned[center]
            _conv_fwd(L, _p(gemb_ref), C, None, 0, None, 0.0, 0, H, W, _p(w2), None, _p(gcen), _p(gcen), C, None, 0, B, 3, 1,
                      1, ACT_NONE, 0.0, 0, H, W, 'tsa dgrad tAtt_2', wparam=w2)
        else:
            galigned = None
        return galigned, None, gw1, gb1, gw2, gb2


def tsa_temporal_block(aligned_nb, center, tAtt_1, tAtt_2):
    """aligned_nb: [N, B, C, H, W] frame-major -> modulated features [B, N*C, H, W]"""
    return _TSATemporalBlock.apply(aligned_nb, int(center), tAtt_1.weight, tAtt_1.bias, tAtt_2.weight, tAtt_2.bias)


class _TSAOutput(Function):
    @staticmethod
    def forward(ctx, fea, att, att_add):
        _need_cuda(fea, att, att_add)
        fea, att, att_add = fea.contiguous(), att.contiguous(), att_add.contiguous()
        out = torch.empty_like(fea)
        _lib.check(_lib.lib().rvsr_tsa_output_forward(_p(fea), _p(att), _p(att_add), _p(out), fea.numel(), _stream()),
                   'tsa_output_forward')
        ctx.save_for_backward(fea, att)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        fea, att = ctx.saved_tensors
        g = g.contiguous()
        gfea, gatt = torch.empty_like(fea), torch.empty_like(att)
        _lib.check(_lib.lib().rvsr_tsa_output_backward(_p(g), _p(fea), _p(att), _p(gfea), _p(gatt), fea.numel(),
                                                       _stream()), 'tsa_output_backward')
        return gfea, gatt, g


def tsa_output(fea, att, att_add):
    """fea * sigmoid(att) * 2 + att_add"""
    return _TSAOutput.apply(fea, att, att_add)


# ------------------------------------------------------------------------------------------ pyramid / loss
class _PyrDown(Function):
    @staticmethod
    def forward(ctx, x):
        _need_cuda(x)
        x = x.contiguous()
        B, C, H, W = x.shape
        out = x.new_empty(B, C, (H + 1) // 2, (W + 1) // 2)
        _lib.check(_lib.lib().rvsr_pyr_down_forward(_p(x), _p(out), B * C, H, W, _stream()), 'pyr_down_forward')
        ctx.cfg = (B, C, H, W)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        B, C, H, W = ctx.cfg
        gout = gout.contiguous()
        gin = gout.new_empty(B, C, H, W)
        _lib.check(_lib.lib().rvsr_pyr_down_backward(_p(gout), _p(gin), B * C, H, W, _stream()), 'pyr_down_backward')
        return gin


def pyr_down(x):
    """downsample(conv_gauss(x, gauss_kernel)) (utils/util.py:503-510)"""
    return _PyrDown.apply(x)


class _PyrUpDiff(Function):
    @staticmethod
    def forward(ctx, cur, down):
        _need_cuda(cur, down)
        cur, down = cur.contiguous(), down.contiguous()
        B, C, H, W = cur.shape
        if tuple(down.shape) != (B, C, H // 2, W // 2) or H % 2 or W % 2:
            # same failure the reference hits as a shape mismatch in `current - up` (utils/util.py:550)
            raise RuntimeError('pyramid level %s cannot be rebuilt from %s' % (tuple(cur.shape), tuple(down.shape)))
        out = torch.empty_like(cur)
        _lib.check(_lib.lib().rvsr_pyr_updiff_forward(_p(cur), _p(down), _p(out), B * C, H, W, _stream()),
                   'pyr_updiff_forward')
        ctx.cfg = (B, C, H, W)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        B, C, H, W = ctx.cfg
        gout = gout.contiguous()
        gdown = gout.new_empty(B, C, H // 2, W // 2)
        _lib.check(_lib.lib().rvsr_pyr_updiff_backward(_p(gout), _p(gdown), B * C, H, W, _stream()),
                   'pyr_updiff_backward')
        return gout, gdown


def pyr_updiff(cur, down):
    """cur - upsample(down) (utils/util.py:513-516,548-551)"""
    return _PyrUpDiff.apply(cur, down)


class _Charbonnier(Function):
    @staticmethod
    def forward(ctx, x, y, eps, mean):
        _need_cuda(x, y)
        x, y = x.contiguous(), y.contiguous()
        n = x.numel()
        out = x.new_empty(())
        L = _lib.lib()
        ws = _workspace(L.rvsr_charbonnier_workspace_bytes(), x.device)
        scale = 1.0 / n if mean else 1.0
        _lib.check(L.rvsr_charbonnier_forward(_p(x), _p(y), n, eps, scale, _p(out), _p(ws), _stream()),
                   'charbonnier_forward')
        ctx.cfg = (eps, scale)
        ctx.save_for_backward(x, y)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        eps, scale = ctx.cfg
        g = g.contiguous()
        gx = torch.empty_like(x)
        _lib.check(_lib.lib().rvsr_charbonnier_backward(_p(x), _p(y), _p(g), scale, eps, _p(gx), x.numel(), _stream()),
                   'charbonnier_backward')
        gy = -gx if ctx.needs_input_grad[1] else None
        return gx, gy, None, None


def charbonnier(x, y, eps=1e-6, reduction='mean'):
    return _Charbonnier.apply(x, y, float(eps), reduction == 'mean')


class _GWLoss(Function):
    @staticmethod
    def forward(ctx, x1, x2, w, mean):
        _need_cuda(x1, x2)
        x1, x2 = x1.contiguous(), x2.contiguous()
        B, C, H, W = x1.shape
        n = x1.numel()
        out = x1.new_empty(())
        L = _lib.lib()
        ws = _workspace(L.rvsr_charbonnier_workspace_bytes(), x1.device)
        scale = 1.0 / n if mean else 1.0
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        fac = x1.new_empty(3, B, C, H, W) if need else None
        _lib.check(L.rvsr_gwloss_forward(_p(x1), _p(x2), B * C, H, W, float(w), scale, _p(out),
                                         _p(fac[0]) if need else None, _p(fac[1]) if need else None,
                                         _p(fac[2]) if need else None, _p(ws), _stream()), 'gwloss_forward')
        ctx.cfg = (scale, B, C, H, W)
        ctx.save_for_backward(fac)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (fac,) = ctx.saved_tensors
        scale, B, C, H, W = ctx.cfg
        g = g.contiguous()
        gx = fac.new_empty(B, C, H, W)
        _lib.check(_lib.lib().rvsr_gwloss_backward(_p(fac[0]), _p(fac[1]), _p(fac[2]), _p(g), scale, _p(gx), B * C, H, W,
                                                   _stream()), 'gwloss_backward')
        return gx, (-gx if ctx.needs_input_grad[1] else None), None, None


def gw_loss(x1, x2, w=4, reduction='mean'):
    """Gradient-weighted loss (codes/models/loss.py:54-80), one fused pass."""
    return _GWLoss.apply(x1, x2, float(w), reduction == 'mean')


class _PixelLoss(Function):
    """scale * sum f(x - y), f = |d| / d^2 / Huber / Charbonnier (rvsr_pixel_loss_*)."""

    @staticmethod
    def forward(ctx, x, y, mode, param, mean):
        _need_cuda(x, y)
        if x.shape != y.shape:
            raise RuntimeError('pixel loss: shapes %s and %s differ' % (tuple(x.shape), tuple(y.shape)))
        x, y = x.contiguous(), y.contiguous()
        n = x.numel()
        out = x.new_empty(())
        L = _lib.lib()
        ws = _workspace(L.rvsr_reduce_workspace_bytes(), x.device)
        scale = 1.0 / n if mean else 1.0
        _lib.check(L.rvsr_pixel_loss_forward(_p(x), _p(y), n, mode, param, scale, _p(out), _p(ws), _stream()),
                   'pixel_loss_forward')
        ctx.cfg = (mode, param, scale)
        ctx.save_for_backward(x, y)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        mode, param, scale = ctx.cfg
        g = g.contiguous()
        gx = torch.empty_like(x)
        _lib.check(_lib.lib().rvsr_pixel_loss_backward(_p(x), _p(y), _p(g), mode, param, scale, _p(gx), x.numel(),
                                                       _stream()), 'pixel_loss_backward')
        return gx, (-gx if ctx.needs_input_grad[1] else None), None, None, None


PIX_L1, PIX_L2, PIX_HUBER, PIX_CHARBONNIER = 0, 1, 2, 3


def pixel_loss(x, y, mode, param=0.0, reduction='mean'):
    """nn.L1Loss / nn.MSELoss / HuberLoss(delta=param) / CharbonnierLoss(eps=param) as one reduction kernel."""
    return _PixelLoss.apply(x, y, int(mode), float(param), reduction == 'mean')


class _SSIMLoss(Function):
    """1 - mean(ssim_map(x, y)) with the 11x11 sigma-1.5 window (IQA_pytorch.SSIM(...)(x, y, as_loss=True))."""

    @staticmethod
    def forward(ctx, x, y):
        _need_cuda(x, y)
        if x.shape != y.shape or x.dim() != 4:
            raise RuntimeError('ssim: expected two equal (B, C, H, W) tensors, got %s and %s' % (tuple(x.shape), tuple(y.shape)))
        x, y = x.contiguous(), y.contiguous()
        B, C, H, W = x.shape
        out = x.new_empty(())
        L = _lib.lib()
        ws = _workspace(L.rvsr_reduce_workspace_bytes(), x.device)
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        maps = x.new_empty(3, B * C, H - 10, W - 10) if need and H > 10 and W > 10 else None
        scale = 1.0 / max(B * C * (H - 10) * (W - 10), 1)
        _lib.check(L.rvsr_ssim_forward(_p(x), _p(y), B * C, H, W, scale, _p(out),
                                       _p(maps[0]) if maps is not None else None,
                                       _p(maps[1]) if maps is not None else None,
                                       _p(maps[2]) if maps is not None else None, _p(ws), _stream()), 'ssim_forward')
        ctx.cfg = (scale, B * C, H, W)
        ctx.save_for_backward(x, y, maps)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, y, maps = ctx.saved_tensors
        scale, planes, H, W = ctx.cfg
        g = g.contiguous()
        L = _lib.lib()
        gx = gy = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            _lib.check(L.rvsr_ssim_backward(_p(x), _p(y), _p(maps[0]), _p(maps[1]), _p(maps[2]), _p(g), scale, _p(gx),
                                            planes, H, W, _stream()), 'ssim_backward')
        if ctx.needs_input_grad[1]:
            # SSIM is symmetric: the derivative maps w.r.t. the second image are those of ssim(y, x)
            tmp = x.new_empty(())
            m2 = torch.empty_like(maps)
            ws = _workspace(L.rvsr_reduce_workspace_bytes(), x.device)
            _lib.check(L.rvsr_ssim_forward(_p(y), _p(x), planes, H, W, scale, _p(tmp), _p(m2[0]), _p(m2[1]), _p(m2[2]),
                                           _p(ws), _stream()), 'ssim_forward (swapped)')
            gy = torch.empty_like(y)
            _lib.check(L.rvsr_ssim_backward(_p(y), _p(x), _p(m2[0]), _p(m2[1]), _p(m2[2]), _p(g), scale, _p(gy), planes,
                                            H, W, _stream()), 'ssim_backward (swapped)')
        return gx, gy


def ssim_loss(x, y):
    return _SSIMLoss.apply(x, y)


class _ConvGauss(Function):
    @staticmethod
    def forward(ctx, x, gain):
        _need_cuda(x)
        x = x.contiguous()
        B, C, H, W = x.shape
        out = torch.empty_like(x)
        _lib.check(_lib.lib().rvsr_conv_gauss_forward(_p(x), _p(out), B * C, H, W, gain, _stream()), 'conv_gauss_forward')
        ctx.cfg = (B * C, H, W, gain)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        planes, H, W, gain = ctx.cfg
        gout = gout.contiguous()
        gin = torch.empty_like(gout)
        _lib.check(_lib.lib().rvsr_conv_gauss_backward(_p(gout), _p(gin), planes, H, W, gain, _stream()),
                   'conv_gauss_backward')
        return gin, None


def conv_gauss(x, gain=1.0):
    """conv_gauss(x, gain * gauss_kernel()) (utils/util.py:503-506)"""
    return _ConvGauss.apply(x, float(gain))


class _PyrUpsample(Function):
    @staticmethod
    def forward(ctx, x):
        _need_cuda(x)
        x = x.contiguous()
        B, C, H, W = x.shape
        out = x.new_empty(B, C, 2 * H, 2 * W)
        _lib.check(_lib.lib().rvsr_pyr_upsample_forward(_p(x), _p(out), B * C, H, W, _stream()), 'pyr_upsample_forward')
        ctx.cfg = (B, C, H, W)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        B, C, H, W = ctx.cfg
        gout = gout.contiguous()
        gin = gout.new_empty(B, C, H, W)
        _lib.check(_lib.lib().rvsr_pyr_upsample_backward(_p(gout), _p(gin), B * C, H, W, _stream()),
                   'pyr_upsample_backward')
        return gin


def pyr_upsample(x):
    """upsample(x) of the pyramid helpers (utils/util.py:513-516)"""
    return _PyrUpsample.apply(x)


# ------------------------------------------------------------------------------------------ optimizer / augmentation
def adam_step_(param, grad, exp_avg, exp_avg_sq, step_size, beta1, beta2, eps, weight_decay, bias_correction2_sqrt):
    """In-place Adam update of one flat f32 buffer (rvsr_adam_step); no autograd."""
    _need_cuda(param, grad, exp_avg, exp_avg_sq)
    n = param.numel()
    for t in (grad, exp_avg, exp_avg_sq):
        if t.numel() != n or not t.is_contiguous():
            raise RuntimeError('adam_step_: buffers must be contiguous and of equal length')
    with torch.cuda.device(param.device):
        _lib.check(_lib.lib().rvsr_adam_step(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), n, step_size, beta1, beta2,
                                             eps, weight_decay, bias_correction2_sqrt, _stream()), 'adam_step')


def augment_clips(im1, im2, perm=(0, 1, 2), box_mode=0, box=(0, 0, 0, 0), v=1.0, colour=None):
    """One pass over a clip pair [..., 3, H, W] (rvsr_augment_clips): channel permutation, CutBlur box paste
    (box = (y0, y1, x0, x1)), blend with a per-frame colour.  Returns two new tensors; no autograd (data path)."""
    _need_cuda(im1, im2, colour)
    if im1.shape != im2.shape or im1.dim() < 3 or im1.shape[-3] != 3:
        raise ValueError('augment_clips: expected two equal [..., 3, H, W] clips, got %s and %s'
                         % (tuple(im1.shape), tuple(im2.shape)))
    im1, im2 = im1.contiguous(), im2.contiguous()
    H, W = im1.shape[-2:]
    frames = im1.numel() // (3 * H * W)
    if colour is not None:
        colour = colour.contiguous()
        if colour.numel() != frames * 3:
            raise ValueError('augment_clips: colour must hold one value per (frame, channel)')
    out1, out2 = torch.empty_like(im1), torch.empty_like(im2)
    with torch.cuda.device(im1.device):
        _lib.check(_lib.lib().rvsr_augment_clips(_p(im1), _p(im2), _p(out1), _p(out2), _p(colour), frames, H, W,
                                                 int(perm[0]), int(perm[1]), int(perm[2]), int(box_mode), int(box[0]),
                                                 int(box[1]), int(box[2]), int(box[3]), float(v), _stream()),
                   'augment_clips')
    return out1, out2


# ------------------------------------------------------------------------------------------ device guard
def _guarded(fn):
    """Run an operator on the device of its tensors: HIP launches go to the CURRENT device, and `_stream()` /
    `_workspace()` are the current device's, so a module living on cuda:1 while cuda:0 is current (DataParallel
    replicas, VideoSR_AllPair_model_YCbCr_Split.py:36; the reference's extension does the same with
    at::DeviceGuard, deform_conv_cuda.cpp:499,581) must switch first."""
    @functools.wraps(fn)
    def wrapper(ctx, *args):
        for a in args:
            if torch.is_tensor(a) and a.is_cuda:
                if a.device.index != torch.cuda.current_device():
                    with torch.cuda.device(a.device):
                        return fn(ctx, *args)
                break
        return fn(ctx, *args)
    return wrapper


for _cls in list(globals().values()):
    if isinstance(_cls, type) and issubclass(_cls, Function) and _cls is not Function:
        _cls.forward = staticmethod(_guarded(_cls.forward))
        _cls.backward = staticmethod(_guarded(_cls.backward))
