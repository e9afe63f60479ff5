"""Shape-only stand-ins for gym.spaces (fixture generation only)."""


class Space(object):
    def __init__(self, shape=None, dtype=None):
        self.shape = shape
        self.dtype = dtype


class Box(Space):
    def __init__(self, low=None, high=None, shape=None, dtype=None):
        super().__init__(shape, dtype)
        self.low, self.high = low, high


class Discrete(Space):
    def __init__(self, n):
        super().__init__((), int)
        self.n = n


class Tuple(Space):
    def __init__(self, spaces):
        super().__init__(None, None)
        self.spaces = list(spaces)


class Dict(Space):
    def __init__(self, spaces=None, **kw):
        super().__init__(None, None)
        self.spaces = dict(spaces or {}, **kw)
