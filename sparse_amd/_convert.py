"""Device-side format conversion (A6).  Filled in by the conversion milestone."""


def coo_to_gcxs_arrays(x, compressed_axes=None, idx_dtype=None):
    raise NotImplementedError


def gcxs_relayout(x, shape, axes, compressed_axes, transpose=False, reshape=False):
    raise NotImplementedError


def gcxs_to_coo(x):
    raise NotImplementedError


def gcxs_todense(x):
    raise NotImplementedError


def gcxs_prune(x):
    raise NotImplementedError
