"""Host layer of the MI355X NNUE hot path: ctypes binding (lib), autograd bridges (ops) and the
data-parallel trainer (trainer).  The drop-in modules that mirror the reference's ``nnue.py`` and
``serialize.py`` live one directory up and import from here."""
from . import lib  # noqa: F401

__all__ = ["lib"]
