"""Mirror of the reference utils/helpers.py:58-68 (`as_mat`): no UMat exists here, arrays pass through."""
import numpy as np


def as_mat(mat):
    get = getattr(mat, "get", None)
    return get() if callable(get) and not isinstance(mat, np.ndarray) else mat


def to_odd(n: int) -> int:
    n = int(n)
    return n if n % 2 == 1 else n + 1
