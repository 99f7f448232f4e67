"""Test-time driver around the hot path.

Mirrors, with the same names and argument meaning, the helpers the reference's test scripts use
(codes/test_RealVSR_wi_GT.py:106-130):

  index_generation   codes/data/util.py:169-214     temporal padding of the N-frame window
  single_forward     codes/utils/util.py:222-237
  flipx4_forward     codes/utils/util.py:240-261    x4 flip self-ensemble
  ycbcr_to_bgr_u8    codes/utils/util.py:151-181 + codes/data/util.py:397-416 (+ clip/round/uint8 of the script)

and adds what the reference does not have: `SlidingWindowRunner`, which runs a whole clip and computes the
per-frame part of EDVR (conv_first, front residual blocks, L2/L3 pyramid convs) ONCE per frame instead of once
per window the frame appears in (N times).  The outputs are identical, bit for bit, to calling the network on
every window: the kernels are deterministic per sample and the per-frame stage does not look at its neighbours.

Unlike the reference helpers the outputs stay on the GPU (the reference moves every f32 output to the host,
`.cpu()`, utils/util.py:236): use `ycbcr_to_bgr_u8` to bring back 3 bytes per pixel.
"""
import ctypes

import torch

from . import _lib
from .functional import _need_cuda, _p, _stream


def index_generation(crt_i, max_n, N, padding='reflection'):
    """Indices of the N frames centred on `crt_i` in a sequence of `max_n` frames (counted from 1).

    padding: replicate | reflection | new_info | circle; e.g. crt_i = 0, N = 5:
    [0, 0, 0, 1, 2] | [2, 1, 0, 1, 2] | [4, 3, 0, 1, 2] | [3, 4, 0, 1, 2]   (codes/data/util.py:169-214)."""
    last = max_n - 1
    half = N // 2
    if padding not in ('replicate', 'reflection', 'new_info', 'circle'):
        raise ValueError('Wrong padding mode')
    idx = []
    for i in range(crt_i - half, crt_i + half + 1):
        if i < 0:
            j = {'replicate': 0, 'reflection': -i, 'new_info': crt_i + half - i, 'circle': N + i}[padding]
        elif i > last:
            j = {'replicate': last, 'reflection': 2 * last - i, 'new_info': crt_i - half - (i - last),
                 'circle': i - N}[padding]
        else:
            j = i
        idx.append(j)
    return idx


def single_forward(model, inp):
    """model(inp) without autograd; the first element if the model returns a list/tuple
    (codes/utils/util.py:222-237).  The result stays on the GPU."""
    with torch.no_grad():
        out = model(inp)
        if isinstance(out, (list, tuple)):
            out = out[0]
    return out.detach().float()


def flipx4_forward(model, inp):
    """Mean of the outputs for the input, its W flip, its H flip and both (codes/utils/util.py:240-261)."""
    acc = single_forward(model, inp)
    for dims in ((-1,), (-2,), (-2, -1)):
        acc = acc + torch.flip(single_forward(model, torch.flip(inp, dims)), dims)
    return acc / 4


def ycbcr_to_bgr_u8(ycc):
    """[3, H, W] (or [1, 3, H, W]) f32 YCbCr network output -> [H, W, 3] uint8 BGR, on the GPU, bit-exact with
    the reference's host chain tensor2img(float32) -> ycbcr2bgr -> clip*255 round uint8
    (codes/test_RealVSR_wi_GT.py:122-123)."""
    if ycc.dim() == 4 and ycc.shape[0] == 1:
        ycc = ycc[0]
    if ycc.dim() != 3 or ycc.shape[0] != 3:
        raise RuntimeError('ycbcr_to_bgr_u8: expected [3, H, W], got %s' % (tuple(ycc.shape),))
    _need_cuda(ycc)
    ycc = ycc.float().contiguous()
    _, H, W = ycc.shape
    out = torch.empty(H, W, 3, dtype=torch.uint8, device=ycc.device)
    _lib.check(_lib.lib().rvsr_ycbcr_to_bgr_u8(_p(ycc), ctypes.c_void_p(out.data_ptr()), H, W, _stream()),
               'ycbcr_to_bgr_u8')
    return out


class SlidingWindowRunner(object):
    """Super-resolve a clip frame by frame with per-frame feature reuse.

    net: a realvsr_amd EDVR / EDVR_NoUp (anything with `extract_features`, `align_fuse_reconstruct`, `center`).
    N: frames per window; padding: temporal padding mode of `index_generation`.
    chunk: how many frames go through the per-frame stage at once.
    """

    def __init__(self, net, N, padding='replicate', chunk=8, flip_ensemble=False, use_graph=False):
        """use_graph: run the window stage (alignment + fusion + reconstruction of one output frame) as hipGraph replays
        (torch.cuda.CUDAGraph drives hipStreamBeginCapture; every kernel of the path is a plain launch on the capturing stream) --
        BASELINE config 5 asks for a graph-captured sliding window; it matters for small frames, which are launch-bound.  The
        features of the N frames of a window live in a RING of N static slots: consecutive windows share N - 1 frames, so a step
        overwrites one slot (the entering frame; at a clip's ends whatever the padding mode changes) instead of gathering the whole
        window, and the window's frame order is a rotation of the ring -- one captured graph per rotation (N graphs, captured on
        first use, sharing one memory pool), replayed round-robin."""
        if N // 2 != net.center:
            raise RuntimeError('window of %d frames does not match the network centre %d' % (N, net.center))
        self.net, self.N, self.padding, self.chunk, self.flip = net, N, padding, chunk, flip_ensemble
        self.use_graph = use_graph
        self._ring = None       # graph mode: {key, slots s1 / s2 / s3, centre-frame buffer sx, graphs: rotation -> (graph, output), pool}

    def _ring_state(self, L1, L2, L3, frame):
        key = (tuple(L1.shape[1:]), tuple(frame.shape), L1.device.index)
        if self._ring is None or self._ring['key'] != key:
            N = self.N
            slots = [f.new_zeros((N,) + tuple(f.shape[1:])) for f in (L1, L2, L3)]
            self._ring = {'key': key, 'slots': slots, 'sx': frame.new_zeros(frame.shape), 'graphs': {}, 'pool': torch.cuda.graph_pool_handle()}
        return self._ring

    def _ring_graph(self, ring, r):
        """The window stage on the ring rotated by r: frame j of the window is slot (r + j) % N."""
        if r not in ring['graphs']:
            N, (s1, s2, s3), sx = self.N, ring['slots'], ring['sx']
            order = [(r + j) % N for j in range(N)]

            def stage():
                return self.net.align_fuse_reconstruct([s1[k:k + 1] for k in order], [s2[k:k + 1] for k in order],
                                                       [s3[k:k + 1] for k in order], sx)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):   # warm-up on the capture stream: sizes the scratch workspace
                stage()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side, pool=ring['pool']):
                out = stage()
            ring['graphs'][r] = (graph, out)
        return ring['graphs'][r]

    def _features(self, clip):
        feats = [[], [], []]
        for s in range(0, clip.shape[0], self.chunk):
            for lvl, f in enumerate(self.net.extract_features(clip[s:s + self.chunk].contiguous())):
                feats[lvl].append(f)
        return [torch.cat(f, 0) for f in feats]

    def _run(self, clip):
        T = clip.shape[0]
        L1, L2, L3 = self._features(clip)
        outs = []
        if self.use_graph:
            ring = self._ring_state(L1, L2, L3, clip[0:1])
            held = [None] * self.N      # which frame each ring slot holds
        for t in range(T):
            idx = index_generation(t, T, self.N, self.padding)
            if self.use_graph:
                r = t % self.N
                for j, f in enumerate(idx):
                    k = (r + j) % self.N
                    if held[k] != f:
                        for slot, feat in zip(ring['slots'], (L1, L2, L3)):
                            slot[k].copy_(feat[f])
                        held[k] = f
                ring['sx'].copy_(clip[t:t + 1])
                graph, gout = self._ring_graph(ring, r)
                graph.replay()
                outs.append(gout.clone())
            else:
                outs.append(self.net.align_fuse_reconstruct([L1[j:j + 1] for j in idx], [L2[j:j + 1] for j in idx],
                                                            [L3[j:j + 1] for j in idx], clip[t:t + 1].contiguous()))
        return torch.cat(outs, 0)

    def __call__(self, clip):
        """clip: [T, C, H, W] LR frames on the GPU -> [T, C, sH, sW] outputs (s = 4 for EDVR, 1 for EDVR_NoUp)."""
        if clip.dim() != 4:
            raise RuntimeError('SlidingWindowRunner: expected [T, C, H, W]')
        if clip.shape[2] % 4 or clip.shape[3] % 4:
            raise RuntimeError('EDVR needs H and W divisible by 4 (got %dx%d)' % (clip.shape[2], clip.shape[3]))
        _need_cuda(clip)
        with torch.no_grad():
            out = self._run(clip)
            if self.flip:
                for dims in ((-1,), (-2,), (-2, -1)):
                    out = out + torch.flip(self._run(torch.flip(clip, dims)), dims)
                out = out / 4
        return out

    def reference_order(self, clip):
        """The reference's loop (one full network call per window, test_RealVSR_wi_GT.py:114-119); for parity
        tests and for measuring what the reuse buys."""
        T = clip.shape[0]
        fwd = flipx4_forward if self.flip else single_forward
        outs = []
        for t in range(T):
            idx = torch.tensor(index_generation(t, T, self.N, self.padding), device=clip.device)
            outs.append(fwd(self.net, clip.index_select(0, idx).unsqueeze(0)))
        return torch.cat(outs, 0)
