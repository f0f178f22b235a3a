"""oodgan — MI355X-native hot path of OOD-GAN-inversion (StyleGAN2 generator, SAMM/SAIM ops, W+ loop).

Python host code over hand-written HIP kernels (liboodgan_hip.so, C ABI in include/oodgan.h).
Class / function names mirror the reference (src/ops/StyleGAN/model.py, src/ops/op, src/archs)."""
from . import synth  # noqa: F401
from ._lib import LIB_PATH, lib  # noqa: F401

__version__ = '0.1.0'


def __getattr__(name):
    # heavy modules are imported lazily so that `import oodgan` works on a box without the .so
    import importlib
    if name in ('modules', 'ops', 'engine', 'samm', 'arch', 'io'):
        return importlib.import_module(f'.{name}', __name__)
    for mod in ('modules', 'ops', 'engine'):
        m = importlib.import_module(f'.{mod}', __name__)
        if hasattr(m, name):
            return getattr(m, name)
    raise AttributeError(name)
