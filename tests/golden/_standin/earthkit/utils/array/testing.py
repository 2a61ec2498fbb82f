from . import _NS

NAMESPACE_DEVICES = [(_NS, "cpu")]
