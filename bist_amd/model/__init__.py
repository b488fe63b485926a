"""Drop-in counterpart of the reference's ``model`` package (same module and class names)."""
from . import decode, decoder, encoder, generator, label_smoothing, modules, mtn, optimize  # noqa: F401
from .mtn import MTN, make_model  # noqa: F401
