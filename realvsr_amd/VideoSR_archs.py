"""define_G for the EDVR family (codes/models/VideoSR_archs.py:35-77, EDVR / EDVR_NoUp branches)."""
from .archs import EDVR_arch


def define_G(opt):
    opt_net = opt['network_G']
    which_model = opt_net['which_model_G']
    if which_model in ('EDVR', 'EDVR_NoUp'):
        cls = EDVR_arch.EDVR if which_model == 'EDVR' else EDVR_arch.EDVR_NoUp
        get = opt_net.get if hasattr(opt_net, 'get') else (lambda k: opt_net[k])
        return cls(nf=opt_net['nf'], nc=opt_net['nc'], nframes=opt_net['nframes'], groups=opt_net['groups'],
                   front_RBs=opt_net['front_RBs'], back_RBs=opt_net['back_RBs'], center=get('center'),
                   predeblur=get('predeblur'), HR_in=get('HR_in'), w_TSA=get('w_TSA'))
    raise NotImplementedError('Generator model [{:s}] not recognized'.format(which_model))
