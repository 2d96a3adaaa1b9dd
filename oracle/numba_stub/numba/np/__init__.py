from . import numpy_support  # noqa: F401
