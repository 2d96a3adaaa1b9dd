"""``numba.typed.List`` stand-in: a plain Python list."""


class List(list):
    @classmethod
    def empty_list(cls, _item_type=None):
        return cls()
