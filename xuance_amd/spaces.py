"""Minimal observation/action space objects with the attributes the reference reads from gymnasium spaces
(``.shape``, ``.n``, ``.low``, ``.high``); real gymnasium spaces are accepted everywhere by duck typing."""
import numpy as np


class Space:
    def __init__(self, shape=(), dtype=np.float32):
        self.shape, self.dtype = tuple(shape), dtype


class Box(Space):
    def __init__(self, low=-np.inf, high=np.inf, shape=(), dtype=np.float32):
        super().__init__(shape, dtype)
        self.low, self.high = low, high


class Discrete(Space):
    def __init__(self, n):
        super().__init__((), np.int64)
        self.n = int(n)


def space2shape(space):
    """xuance/environment/utils/shapes.py:5-46 for the Box / Discrete / tuple cases used on this path."""
    if space is None:
        return None
    if isinstance(space, (tuple, list)):
        return tuple(space)
    if hasattr(space, "n") and not getattr(space, "shape", ()):
        return ()
    return tuple(space.shape)


def is_discrete(space):
    return hasattr(space, "n")
