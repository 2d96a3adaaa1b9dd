"""save_npz / load_npz in the reference's on-disk layout (reference
sparse/numba_backend/_io.py:7-132): COO -> `data, coords, shape, fill_value`; GCXS ->
`data, indices, indptr, compressed_axes, shape, fill_value`.  Files are interchangeable with the
reference in both directions.  Host-side I/O only (arrays are copied D2H / H2D)."""
import numpy as np

from . import _device as dev


def save_npz(filename, matrix, compressed=True):
    from ._coo import COO
    from ._gcxs import GCXS

    nodes = {"data": dev.to_numpy(matrix.data), "shape": np.asarray(matrix.shape), "fill_value": matrix.fill_value}
    if isinstance(matrix, COO):
        nodes["coords"] = dev.to_numpy(matrix.coords)
    elif isinstance(matrix, GCXS):
        nodes["indices"] = dev.to_numpy(matrix.indices)
        nodes["indptr"] = dev.to_numpy(matrix.indptr)
        nodes["compressed_axes"] = matrix.compressed_axes
    else:
        raise NotImplementedError(f"cannot save {type(matrix)}")
    (np.savez_compressed if compressed else np.savez)(filename, **nodes)


def load_npz(filename, device=None):
    from ._coo import COO
    from ._gcxs import GCXS

    with np.load(filename) as fp:
        try:
            coords, data, shape = fp["coords"], fp["data"], tuple(fp["shape"])
            fill_value = fp["fill_value"][()] if "fill_value" in fp else None
            return COO(coords, data, shape=shape, sorted=True, has_duplicates=False, fill_value=fill_value, device=device)
        except KeyError:
            pass
        try:
            data, indices, indptr = fp["data"], fp["indices"], fp["indptr"]
            ca = fp["compressed_axes"]
            ca = None if ca.ndim == 0 and ca[()] is None else tuple(int(c) for c in np.atleast_1d(ca))
            return GCXS((data, indices, indptr), shape=tuple(fp["shape"]), compressed_axes=ca,
                        fill_value=fp["fill_value"][()], device=device)
        except KeyError as e:
            raise RuntimeError(f"The file {filename!s} does not contain a valid sparse matrix") from e
