"""Attribute sink standing in for pygame: the reference imports its graphics modules
unconditionally, but fixture generation never renders."""


class Surface(object):
    def __init__(self, *a, **k):
        pass


class SurfaceType(Surface):
    pass


class Rect(object):
    def __init__(self, *a, **k):
        pass


class _Sink(object):
    def __getattr__(self, name):
        return _Sink()

    def __call__(self, *a, **k):
        return _Sink()


def __getattr__(name):
    return _Sink()
