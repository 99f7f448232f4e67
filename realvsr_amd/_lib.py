"""ctypes binding of librealvsr_hip.so (C ABI: include/realvsr_hip.h).

The library is built in-tree (``realvsr_amd/csrc/librealvsr_hip.so``) by ``build()`` /
``__graft_entry__.build()``.  If it is missing the product fails loudly: there is no fallback.
"""
import ctypes
import os
import subprocess

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
SO_PATH = os.environ.get('RVSR_SO', os.path.join(_CSRC, 'librealvsr_hip.so'))  # RVSR_SO: developer override for A/B builds
_lib = None

c_fp = ctypes.c_void_p  # device pointers travel as void*
c_int, c_float, c_double, c_size = ctypes.c_int, ctypes.c_float, ctypes.c_double, ctypes.c_size_t

# name -> (restype, argtypes); must list every symbol declared in include/realvsr_hip.h
SIGNATURES = {
    'rvsr_last_error': (ctypes.c_char_p, []),
    'rvsr_modulated_deform_conv_forward': (c_int, [c_fp] * 6 + [c_int] * 16 + [c_fp, c_size, c_fp]),
    'rvsr_modulated_deform_conv_forward_workspace_bytes': (c_size, [c_int] * 2),
    'rvsr_modulated_deform_conv_backward_workspace_bytes': (c_size, [c_int] * 8),
    'rvsr_modulated_deform_conv_backward': (c_int, [c_fp] * 11 + [c_int] * 16 + [c_fp, c_size, c_fp]),
    'rvsr_deform_conv_workspace_bytes': (c_size, [c_int] * 11),
    'rvsr_deform_conv_forward': (c_int, [c_fp] * 4 + [c_int] * 16 + [c_fp, c_size, c_fp]),
    'rvsr_deform_conv_backward_input': (c_int, [c_fp] * 6 + [c_int] * 16 + [c_fp, c_size, c_fp]),
    'rvsr_deform_conv_backward_parameters': (c_int, [c_fp] * 4 + [c_int] * 15 + [c_float, c_int, c_fp, c_size, c_fp]),
    'rvsr_deform_conv_generic_workspace_bytes': (c_size, [c_int] * 12),
    'rvsr_deform_conv_generic_forward_workspace_bytes': (c_size, [c_int] * 12),
    'rvsr_deform_conv_generic_forward': (c_int, [c_int] + [c_fp] * 6 + [c_int] * 15 + [c_fp, c_size, c_fp]),
    'rvsr_deform_conv_generic_backward': (c_int, [c_int] + [c_fp] * 10 + [c_int] * 15 + [c_fp, c_size, c_fp]),
    'rvsr_dcn_pack_forward': (c_int, [c_fp] * 5 + [c_int] * 10 + [c_float, c_fp, c_fp, c_size, c_fp]),
    'rvsr_dcn_offset_probe': (c_int, [c_fp, c_int, c_int, c_int, c_int, c_fp, c_fp]),
    'rvsr_dcn_pack_backward': (c_int, [c_fp] * 5 + [c_float] + [c_fp] * 4 + [c_int] * 9 + [c_fp, c_fp, c_size, c_fp]),
    'rvsr_conv2d_forward': (c_int, [c_fp, c_int, c_fp, c_int, c_fp, c_float, c_int, c_int, c_int, c_fp, c_fp, c_fp,
                                    c_fp, c_int, c_fp, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int,
                                    c_int, c_int, c_fp, c_size, c_fp]),
    'rvsr_conv2d_forward_workspace_bytes': (c_size, [c_int] * 4),
    'rvsr_conv2d_pack_weights': (c_size, [c_fp, c_int, c_int, c_int, c_int, c_fp, c_size, ctypes.POINTER(ctypes.c_longlong), c_fp]),
    'rvsr_dcn_pack_weights': (c_size, [c_fp, c_int, c_int, c_fp, c_size, ctypes.POINTER(ctypes.c_longlong), c_fp]),
    'rvsr_pack_weights_batched': (c_int, [c_fp, c_int, c_fp]),
    'rvsr_set_gemm_mode': (None, [c_int]),
    'rvsr_get_gemm_mode': (c_int, []),
    'rvsr_set_gemm_mode_thread': (None, [c_int]),
    'rvsr_conv2d_wgrad_workspace_bytes': (c_size, [c_int] * 8),
    'rvsr_conv2d_backward_weight': (c_int, [c_fp, c_int, c_fp, c_int, c_int, c_int, c_fp, c_fp, c_float, c_int, c_int,
                                            c_int, c_fp, c_fp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_fp,
                                            c_size, c_fp]),
    'rvsr_upsample_bilinear_forward': (c_int, [c_fp, c_fp, c_size, c_int, c_int, c_int, c_float, c_fp]),
    'rvsr_upsample_bilinear_backward': (c_int, [c_fp, c_fp, c_size, c_int, c_int, c_int, c_float, c_fp]),
    'rvsr_maxavgpool_forward': (c_int, [c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp]),
    'rvsr_maxavgpool_backward': (c_int, [c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp]),
    'rvsr_tsa_temporal_forward': (c_int, [c_fp] * 5 + [c_int] * 6 + [c_fp]),
    'rvsr_tsa_temporal_backward': (c_int, [c_fp] * 8 + [c_int] * 6 + [c_fp]),
    'rvsr_tsa_output_forward': (c_int, [c_fp] * 4 + [c_size, c_fp]),
    'rvsr_tsa_output_backward': (c_int, [c_fp] * 5 + [c_size, c_fp]),
    'rvsr_pyr_down_forward': (c_int, [c_fp, c_fp, c_size, c_int, c_int, c_fp]),
    'rvsr_pyr_down_backward': (c_int, [c_fp, c_fp, c_size, c_int, c_int, c_fp]),
    'rvsr_pyr_updiff_forward': (c_int, [c_fp, c_fp, c_fp, c_size, c_int, c_int, c_fp]),
    'rvsr_pyr_updiff_backward': (c_int, [c_fp, c_fp, c_size, c_int, c_int, c_fp]),
    'rvsr_charbonnier_workspace_bytes': (c_size, []),
    'rvsr_charbonnier_forward': (c_int, [c_fp, c_fp, c_size, c_float, c_double, c_fp, c_fp, c_fp]),
    'rvsr_charbonnier_backward': (c_int, [c_fp, c_fp, c_fp, c_float, c_float, c_fp, c_size, c_fp]),
    'rvsr_gwloss_forward': (c_int, [c_fp, c_fp, c_size, c_int, c_int, c_float, c_double, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]),
    'rvsr_gwloss_backward': (c_int, [c_fp, c_fp, c_fp, c_fp, c_float, c_fp, c_size, c_int, c_int, c_fp]),
    'rvsr_ycbcr_to_bgr_u8': (c_int, [c_fp, c_fp, c_int, c_int, c_fp]),
    'rvsr_reduce_workspace_bytes': (c_size, []),
    'rvsr_pixel_loss_forward': (c_int, [c_fp, c_fp, c_size, c_int, c_float, c_double, c_fp, c_fp, c_fp]),
    'rvsr_pixel_loss_backward': (c_int, [c_fp, c_fp, c_fp, c_int, c_float, c_float, c_fp, c_size, c_fp]),
    'rvsr_ssim_forward': (c_int, [c_fp, c_fp, c_size, c_int, c_int, c_double, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]),
    'rvsr_ssim_backward': (c_int, [c_fp] * 6 + [c_float, c_fp, c_size, c_int, c_int, c_fp]),
    'rvsr_conv_gauss_forward': (c_int, [c_fp, c_fp, c_size, c_int, c_int, c_float, c_fp]),
    'rvsr_conv_gauss_backward': (c_int, [c_fp, c_fp, c_size, c_int, c_int, c_float, c_fp]),
    'rvsr_pyr_upsample_forward': (c_int, [c_fp, c_fp, c_size, c_int, c_int, c_fp]),
    'rvsr_pyr_upsample_backward': (c_int, [c_fp, c_fp, c_size, c_int, c_int, c_fp]),
    'rvsr_adam_step': (c_int, [c_fp] * 4 + [c_size] + [c_float] * 6 + [c_fp]),
    'rvsr_bcast_add_act': (c_int, [c_fp, c_fp, c_size, c_int, c_int, c_float, c_fp]),
    'rvsr_bcast_reduce_act': (c_int, [c_fp, c_fp, c_fp, c_size, c_int, c_float, c_fp]),
    'rvsr_augment_clips': (c_int, [c_fp] * 5 + [c_size] + [c_int] * 10 + [c_float, c_fp]),
    'rvsr_debug_mfma_rate': (c_int, [c_fp, c_fp, c_int, c_int, c_fp]),
}


def build(verbose=False):
    """Compile every HIP source for gfx950 into csrc/librealvsr_hip.so (hipcc cross-compiles
    without a GPU)."""
    cmd = ['make', '-C', _CSRC, '-j4'] + ([] if verbose else ['-s'])
    subprocess.check_call(cmd)
    if not os.path.exists(SO_PATH):
        raise RuntimeError('build did not produce %s' % SO_PATH)
    return SO_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                'realvsr_amd: %s is missing -- run `python -c "import __graft_entry__ as g; g.build()"` '
                '(or `make -C realvsr_amd/csrc`).  There is no CPU / eager fallback.' % SO_PATH)
        import torch  # noqa: F401  (loads libamdhip64 the .so links against)
        handle = ctypes.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the header and the library disagree
            fn.restype = res
            fn.argtypes = args
        mode = os.environ.get('RVSR_GEMM', 'bf16x3')
        if mode not in GEMM_MODES:
            raise RuntimeError("RVSR_GEMM must be one of %s, got %r" % (sorted(GEMM_MODES), mode))
        handle.rvsr_set_gemm_mode(GEMM_MODES[mode])
        global _fmt_f16fp8
        _fmt_f16fp8 = mode == 'f16fp8'
        _lib = handle
    return _lib


GEMM_MODES = {'bf16x3': 0, 'f32': 1, 'bf16x2': 2, 'bf16': 3, 'f16fp8': 0}   # ('f16fp8': library mode 0 + the per-call format flag, below)
_fmt_f16fp8 = False


def set_gemm_mode(mode):
    """How the matrix cores form a product (tensors, accumulation and all other arithmetic are f32 in every mode):
    'bf16x3' (default) three-term bf16 split, ~2^-17 relative error per product -- f32-grade results;
    'f32'    exact-f32 MFMA, bit-for-bit an fmaf chain, ~5x slower GEMMs;
    'bf16x2' two terms: the weights (in a weight gradient: the output gradient) are rounded to bf16, the other operand stays a hi + lo
             pair -- ~2^-9 per product, i.e. the network with bf16-rounded weights evaluated on f32 activations;
    'bf16'   one term, both operands rounded to bf16 (the usual mixed-precision product).
    The reduced-term modes are opt-in speed modes of conv_fwd5 / conv_wgrad2 / the DCN kernels (every other kernel keeps three terms);
    tests/test_gpu_modes.py holds them to the 1e-3 dB PSNR bound of the north star.
    'f16fp8' (round 5, opt-in, FORWARD convolutions only): everything as in 'bf16x3' except that the 3x3 / stride-1 forward convs with more
             than 32 output channels form a product as a1*b1 in f16 + (a1*b2 + a2*b1) in fp8 e4m3 (a1 = f16(a), a2 = a - a1): 56 instead of
             108 matrix instructions per 16-channel stage, ~1.2e-5 instead of 4.6e-6 per convolution.  It is a per-call
             flag of rvsr_conv2d_forward (w_mode | 4) that realvsr_amd.functional sets while this mode is selected; data and weight
             gradients, the DCN kernels and every other conv keep the three-term bf16 split.
             VALID RANGE (the fp8 pieces are stored unscaled; csrc/bf16x3.h): the cross terms carry their ~4 bits only while |w| and |x| stay
             within e4m3's normal range after the 2^12 residual scale -- |x| <~ 200 (beyond it the residual piece saturates at 448) and
             |w| >~ 2^-6 for the a1 piece (smaller weights -- EDVR's 0.1-scaled kaiming residual blocks, std ~0.006 -- quantise towards 0
             and the format degrades to its f16 main term, ~5e-4 per product, on those layers); |v| > 65504 overflows f16.  Measured on the
             bench network (default + rescaled init): output 2.5e-7 of the oracle's; it is an opt-in speed mode, not the reference arithmetic."""
    global _fmt_f16fp8
    lib().rvsr_set_gemm_mode(GEMM_MODES[mode])
    _fmt_f16fp8 = mode == 'f16fp8'


def fmt_f16fp8():
    return _fmt_f16fp8


def set_gemm_mode_thread(mode):
    """The same choice for the calling host thread only (None: back to the process-wide setting).  Race-free per-call selection when
    several host threads drive the library (nn.DataParallel replicas): every entry point reads the mode on the calling thread."""
    if mode == 'f16fp8':
        # the f16 + fp8 format is a process-wide Python flag on top of library mode 0 (set_gemm_mode): selecting it per thread would silently
        # run the three-term split instead
        raise ValueError("set_gemm_mode_thread: 'f16fp8' is a process-wide format (set_gemm_mode('f16fp8')), not a per-thread mode")
    lib().rvsr_set_gemm_mode_thread(-1 if mode is None else GEMM_MODES[mode])


def get_gemm_mode():
    m = lib().rvsr_get_gemm_mode()
    if m == 0 and _fmt_f16fp8:
        return 'f16fp8'
    return [k for k, v in GEMM_MODES.items() if v == m][0]


def check(rc, what):
    if rc != 0:
        msg = lib().rvsr_last_error()
        raise RuntimeError('%s failed (code %d): %s' % (what, rc, msg.decode() if msg else ''))
