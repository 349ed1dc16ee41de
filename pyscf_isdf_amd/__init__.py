"""MI355X-native ISDF (interpolative separable density fitting) for PySCF's periodic DF layer.

``ISDF`` mirrors ``pyscf.pbc.df.FFTDF``; the compute path is libmi355_isdf.so (HIP, gfx950) reached
through the C ABI in include/mi355_isdf.h.  Importing this package does not touch the GPU.
"""
__version__ = '0.1.0'


def __getattr__(name):
    if name == 'ISDF':
        from .isdf import ISDF
        return ISDF
    if name == 'MultiGridFFTDF':
        from .multigrid import MultiGridFFTDF
        return MultiGridFFTDF
    raise AttributeError(name)
