def from_dtype(dt):
    return dt


def as_dtype(dt):
    return dt
