"""Test double of the CUAUV `shm` module (outside the reference tree): `shm.<group>.<var>.set(v)` / `.get()`."""


class _Var:
    def __init__(self):
        self._v = 0

    def set(self, v):
        self._v = v

    def get(self):
        return self._v


class _Snapshot:
    """What `shm.<group>.get()` hands out: plain attributes, written back as a whole with `shm.<group>.set(snapshot)`."""


class _Group:
    def __init__(self):
        self.__dict__["_vars"] = {}

    def __getattr__(self, name):
        return self._vars.setdefault(name, _Var())

    def get(self):
        snap = _Snapshot()
        for k, v in self._vars.items():
            setattr(snap, k, v.get())
        return snap

    def set(self, snap):
        for k, v in vars(snap).items():
            self._vars.setdefault(k, _Var()).set(v)


class _Shm:
    def __init__(self):
        self._groups = {}

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return self._groups.setdefault(name, _Group())


import sys as _sys
_sys.modules[__name__] = _Shm()
