"""`COO` container (device-resident).  Filled in by the conversion/elementwise milestone."""
from ._sparse_array import NDArrayOperatorsMixin, SparseArray


class COO(SparseArray, NDArrayOperatorsMixin):
    pass


def as_coo(x, shape=None, fill_value=None, idx_dtype=None):
    raise NotImplementedError
