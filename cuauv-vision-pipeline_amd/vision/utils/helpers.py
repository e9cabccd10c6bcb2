"""Mirror of the reference utils/helpers.py:58-68 (`as_mat`): no UMat exists here, arrays pass through."""
import numpy as np

from vision.devmat import DeviceMat


def as_mat(mat):
    if isinstance(mat, (np.ndarray, DeviceMat)):
        return mat
    get = getattr(mat, "get", None)
    return get() if callable(get) else mat


def device_image(ctx, mat, channels, pending=None):
    """DeviceMat holding `mat` on `ctx`: the image itself when it already lives there, otherwise an upload of the (validated, packed)
    host array (with `pending`, a list: only enqueued - see DeviceMat.from_host).  channels: 1 -> (h, w), 3 -> (h, w, 3), 0 -> either; a trailing axis of length 1 is dropped."""
    if isinstance(mat, DeviceMat):
        if mat.dtype != np.uint8:
            raise TypeError("expected a uint8 image")
        shp = mat.shape
        if len(shp) == 3 and shp[2] == 1 and channels in (0, 1):
            mat = mat.reshaped(shp[:2])
            shp = mat.shape
        if (channels == 3 and not (len(shp) == 3 and shp[2] == 3)) or (channels == 1 and len(shp) != 2) or len(shp) not in (2, 3):
            raise ValueError("expected an (h, w, 3) image" if channels == 3 else "expected an (h, w) image")
        if shp[0] == 0 or shp[1] == 0:
            raise ValueError("empty image")
        mat.refresh_device(ctx)
        return mat
    if not isinstance(mat, np.ndarray) or mat.dtype != np.uint8:
        raise TypeError("expected a uint8 numpy image")
    if mat.ndim == 3 and mat.shape[2] == 1 and channels in (0, 1):
        mat = mat[:, :, 0]
    if (channels == 3 and not (mat.ndim == 3 and mat.shape[2] == 3)) or (channels == 1 and mat.ndim != 2) or mat.ndim not in (2, 3):
        raise ValueError("expected an (h, w, 3) image" if channels == 3 else "expected an (h, w) image")
    if mat.shape[0] == 0 or mat.shape[1] == 0:
        raise ValueError("empty image")
    return DeviceMat.from_host(ctx, mat, pending=pending)


def to_odd(n: int) -> int:
    n = int(n)
    return n if n % 2 == 1 else n + 1
