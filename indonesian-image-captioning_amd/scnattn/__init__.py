"""Host-side runtime of the MI355X SCN+Attention path: ctypes binding of libscnattn.so, autograd
wrappers, the ResNet-152 trunk definition and the data-parallel gradient reducer."""
from . import _lib  # noqa: F401
from ._lib import LIB_PATH, lib  # noqa: F401
