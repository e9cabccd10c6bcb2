"""Handlers: post-processing hooks that borrow their parent module's services (mirror of the reference
core/handlers.py:18-99).  A handler registered through HandlerMixin gets the parent's normalize / normalize_axis /
post / tuners / get_latency bound onto itself, so handler code reads like module code."""
from abc import ABC, abstractmethod
from typing import Dict, List

import numpy as np

_BORROWED = ("normalize_axis", "normalize", "post", "tuners", "get_latency", "_loop")


class HandlerBase(ABC):
    def __init__(self, name: str, parent=None):
        self._name = name
        self._parent = parent
        if parent is not None:
            self._initialize_methods()

    def register(self, parent):
        self._parent = parent
        self._initialize_methods()

    def _initialize_methods(self):
        for attr in _BORROWED:
            setattr(self, attr, getattr(self._parent, attr))

    @abstractmethod
    def process(self, direction: str, image: np.ndarray, *args, **kwargs):
        raise NotImplementedError("HandlerBase.process")

    @property
    def name(self):
        return self._name


class HandlerMixin:
    """Gives a module a name -> handler table (reference core/handlers.py:77-99: `handlers`, `handler_names`; a name used twice is a
    KeyError) and lends every handler the module's services."""

    def __init__(self, handlers: List[HandlerBase] = []):
        table: Dict[str, HandlerBase] = {}
        for h in handlers:
            if h.name in table:
                raise KeyError("Duplicate handler names found!")
            table[h.name] = h
        self._handlers = table
        self._handler_names = set(table)
        for h in table.values():
            h.register(self)

    @property
    def handlers(self):
        return self._handlers

    @property
    def handler_names(self):
        return self._handler_names
