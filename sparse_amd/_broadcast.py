"""Sparse (x) sparse broadcasting (reference _umath.py:95-389): SURVEY.md §8f row N3 ("next")."""


def broadcast_pair(a, b):
    raise NotImplementedError("broadcasting between sparse operands is a 'next' row (SURVEY.md §8f N3)")
