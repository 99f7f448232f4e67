"""Operator boundary of the MI355X build: the package exports exactly the public names of the reference's
`models.archs.dcn` package (its __init__ lists six: two functional entry points and four nn.Modules), so
`from ...dcn import ModulatedDeformConvPack as DCN` keeps working after an import swap."""
from . import deform_conv as _impl

_MODULES = ('DeformConv', 'DeformConvPack', 'ModulatedDeformConv', 'ModulatedDeformConvPack')
_FUNCTIONS = ('deform_conv', 'modulated_deform_conv')
__all__ = list(_MODULES + _FUNCTIONS)
globals().update({name: getattr(_impl, name) for name in __all__})
