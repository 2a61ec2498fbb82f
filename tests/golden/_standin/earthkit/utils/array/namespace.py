from . import _NS

_NUMPY_NAMESPACE = _NS
