"""define_G for the DCN-based generators (codes/models/VideoSR_archs.py:18-45: EDVR / EDVR_NoUp / TDAN branches)."""
from .archs import EDVR_arch, TDAN_arch


def define_G(opt):
    opt_net = opt['network_G']
    which_model = opt_net['which_model_G']
    if which_model in ('EDVR', 'EDVR_NoUp'):
        cls = EDVR_arch.EDVR if which_model == 'EDVR' else EDVR_arch.EDVR_NoUp
        get = opt_net.get if hasattr(opt_net, 'get') else (lambda k: opt_net[k])
        return cls(nf=opt_net['nf'], nc=opt_net['nc'], nframes=opt_net['nframes'], groups=opt_net['groups'],
                   front_RBs=opt_net['front_RBs'], back_RBs=opt_net['back_RBs'], center=get('center'),
                   predeblur=get('predeblur'), HR_in=get('HR_in'), w_TSA=get('w_TSA'))
    if which_model == 'TDAN':
        return TDAN_arch.TDAN(nf=opt_net['nf'], channel=opt_net['nc'], nframes=opt_net['nframes'], nb_f=opt_net['nb_f'],
                              nb_b=opt_net['nb_b'], groups=opt_net['groups'], scale=opt['scale'])
    raise NotImplementedError('Generator model [{:s}] not recognized'.format(which_model))
