import tempfile


class _DummyMeta(type):
    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return Dummy()

    def __or__(cls, other):
        return Dummy()

    def __ror__(cls, other):
        return Dummy()


class Dummy(metaclass=_DummyMeta):
    """Constructible with anything, callable, attribute access returns more dummies."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return Dummy()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return Dummy()

    def __or__(self, other):
        return Dummy()

    __ror__ = __and__ = __rand__ = __or__

    def __int__(self):
        return 0

    def __index__(self):
        return 0

    def __iter__(self):
        return iter(())

    def __bool__(self):
        return False


class _BoundSignal:
    def __init__(self):
        self._slots = []

    def connect(self, slot):
        self._slots.append(slot)

    def disconnect(self, slot=None):
        if slot is None:
            self._slots.clear()
        elif slot in self._slots:
            self._slots.remove(slot)

    def emit(self, *args):
        for s in list(self._slots):
            s(*args)


class pyqtSignal:
    def __init__(self, *types, **kw):
        self._name = None

    def __set_name__(self, owner, name):
        self._name = "_sig_" + name

    def __get__(self, obj, objtype=None):
        if obj is None:
            return self
        d = obj.__dict__
        if self._name not in d:
            d[self._name] = _BoundSignal()
        return d[self._name]


def pyqtSlot(*a, **k):
    def deco(f):
        return f
    return deco


class QObject:
    def __init__(self, *a, **k):
        pass


class QDir(Dummy):
    @staticmethod
    def tempPath():
        return tempfile.gettempdir()


class QSettings(Dummy):
    def value(self, key, default=None, type=None):
        return default

    def setValue(self, *a):
        pass

    def sync(self):
        pass

    def allKeys(self):
        return []

    def fileName(self):
        return ""


class QByteArray(bytearray):
    """Enough of QByteArray for path_creator.array_to_QPath (path_creator.pyx:101-129): a resizable byte buffer that
    numpy can map (np.frombuffer) and QDataStream can hand to a QPainterPath."""

    def resize(self, n):
        if n < len(self):
            del self[n:]
        else:
            self.extend(b"\0" * (n - len(self)))

    def replace(self, pos, length, data):
        self[pos:pos + length] = bytes(data)


class QPainterPath:
    """Keeps the serialised form QDataStream delivered (numVerts, then (type, x, y) per vertex, big endian)."""

    def __init__(self, *a, **k):
        self.serialised = b""

    def vertices(self):
        import numpy as np
        if not self.serialised:
            return np.zeros(0), np.zeros(0)
        n = int.from_bytes(self.serialised[:4], "big", signed=True)
        arr = np.frombuffer(self.serialised, dtype=[("c", ">i4"), ("x", ">f8"), ("y", ">f8")], count=n, offset=4)
        return arr["x"].astype(np.float64), arr["y"].astype(np.float64)


class QDataStream:
    def __init__(self, buffer=None, *a, **k):
        self.buffer = buffer

    def __rshift__(self, path):
        path.serialised = bytes(self.buffer)
        return self


def module_getattr(known):
    def __getattr__(name):
        if name.startswith("__"):
            raise AttributeError(name)
        if name in known:
            return known[name]
        return type(name, (Dummy,), {})
    return __getattr__
