"""Runtime-tunable parameters of a vision module (mirror of the reference core/tuners.py:10-135).

A tuner travels between the module and the GUI as a one-frame CMF block whose payload is
struct.pack(f"{len(name)}s" + fields) with native alignment: 'iii' (value, min, max) for IntTuner,
'ddd' for DoubleTuner, '?' for BoolTuner — the wire format must stay byte-identical."""
import struct
from abc import ABC, abstractmethod
from typing import Callable, Generic, TypeVar

MAX_OPTION_SIZE_BYTE = 256
T = TypeVar("T")


class TunerBase(ABC, Generic[T]):
    _fields = ""

    def __init__(self, name: str, default_value: T):
        assert name.count(" ") == 0, f"Tuner name '{name}' cannot have spaces"
        assert name.count("/") == 0, f"Tuner name '{name}' cannot have slashes"
        self._name = name
        self._current_value = default_value
        self._packing_format = f"{len(name)}s{self._fields}"

    def __str__(self) -> str:
        return f"{self.__class__.__name__}_{self._name}"

    def __hash__(self) -> int:
        return hash(str(self))

    def __eq__(self, other: object) -> bool:
        return isinstance(other, self.__class__) and self._name == other._name

    @property
    def name(self):
        return self._name

    @property
    def value(self):
        return self._current_value

    def byte_size(self) -> int:
        return struct.calcsize(self._packing_format)

    @abstractmethod
    def serialize(self) -> bytes:
        ...

    @abstractmethod
    def deserialize(self, buffer: bytes):
        ...


class _RangedTuner(TunerBase[T]):
    """value + [min, max] + optional validator; an update outside the range is ignored."""

    def __init__(self, name, default_value, min_value, max_value, validator):
        assert min_value <= max_value, f"min value = {min_value} is not leq to max value = {max_value}"
        super().__init__(name, default_value)
        self._min_value, self._max_value = min_value, max_value
        self._validator = lambda x: validator(x) and min_value <= x <= max_value

    def serialize(self) -> bytes:
        return struct.pack(self._packing_format, self._name.encode(), self._current_value, self._min_value, self._max_value)

    def deserialize(self, buffer: bytes):
        name, value, self._min_value, self._max_value = struct.unpack(self._packing_format, buffer)
        self._name = name.decode()
        if self._validator(value):
            self._current_value = value


class IntTuner(_RangedTuner[int]):
    _fields = "iii"

    def __init__(self, name: str, default_value: int, min_value: int = 0, max_value: int = 255,
                 validator: Callable[[int], bool] = lambda x: True):
        super().__init__(name, default_value, min_value, max_value, validator)


class DoubleTuner(_RangedTuner[float]):
    _fields = "ddd"

    def __init__(self, name: str, default_value: float, min_value: float = -10_000, max_value: float = 10_000,
                 validator: Callable[[float], bool] = lambda x: True):
        super().__init__(name, default_value, min_value, max_value, validator)


class BoolTuner(TunerBase[bool]):
    _fields = "?"

    def __init__(self, name: str, default_value: bool):
        super().__init__(name, default_value)

    def serialize(self) -> bytes:
        return struct.pack(self._packing_format, self._name.encode(), self._current_value)

    def deserialize(self, buffer: bytes):
        name, self._current_value = struct.unpack(self._packing_format, buffer)
        self._name = name.decode()
