"""Import the pydata/sparse reference *in place* from /root/reference — TEST INFRASTRUCTURE.

Only usable in the build container (``/root/reference`` does not exist on the GPU box).
Used by ``oracle/gen_golden.py`` to produce the committed fixtures under ``tests/golden/``
and by ``tests/test_oracle_pin.py`` (skipped when the reference is absent) to pin the
C/NumPy restatement in ``oracle/`` against the real reference source.

Nothing in ``sparse_amd`` imports this module.
"""
import os
import sys
import types

REF_ROOT = "/root/reference"
_STUB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "numba_stub")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "sparse", "numba_backend"))


def load():
    """Return the reference ``sparse`` module (numba backend, interpreter mode)."""
    if "sparse" in sys.modules and getattr(sys.modules["sparse"], "_IS_REFERENCE", False):
        return sys.modules["sparse"]
    if not available():
        raise RuntimeError("reference tree not present (expected on the GPU box)")
    sys.dont_write_bytecode = True  # never write __pycache__ into the read-only tree
    os.environ.pop("SPARSE_BACKEND", None)
    if _STUB not in sys.path:
        sys.path.insert(0, _STUB)
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    ver = types.ModuleType("sparse._version")
    ver.__version__ = "0.0.0+reference"
    ver.__version_tuple__ = (0, 0, 0)
    sys.modules["sparse._version"] = ver
    import sparse  # noqa: E402

    sparse._IS_REFERENCE = True
    return sparse
