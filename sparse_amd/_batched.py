"""N-D matmul broadcasting (reference `_matmul_recurser`, _common.py:278-293): SURVEY.md §8f row N2
("next").  Not built in round 1."""


def matmul_batched(a, b):
    raise NotImplementedError("N-D (x) N-D matmul broadcasting is a 'next' row (SURVEY.md §8f N2)")


def take_leading(x, i):
    raise NotImplementedError("integer indexing is a 'next' row (SURVEY.md §8f N2)")
