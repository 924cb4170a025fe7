"""Ambient per-iteration metric store.

The models only need three things from it (reference: vidgen/utils/events.py:16-25 and the EventStorage class):
the current iteration (`get_event_storage().iter`), somewhere to drop scalars / images during a forward, and a
`with EventStorage(i):` scope that makes a store the ambient one.  This is an independent, smaller implementation
with the same public names: a thread-local stack of stores, scalars kept as per-name series of (value, iteration).
"""
import contextlib
import threading


class _Ambient(threading.local):
    def __init__(self):
        self.stack = []


_ambient = _Ambient()


def get_event_storage():
    """The innermost active store; an error outside any `with EventStorage(...)` block."""
    if not _ambient.stack:
        raise AssertionError("get_event_storage() has to be called inside a 'with EventStorage(...)' context!")
    return _ambient.stack[-1]


class EventStorage:
    """Scalars, images and the iteration counter of one training / evaluation loop."""

    def __init__(self, start_iter=0):
        self._n = int(start_iter)          # current iteration
        self._series = {}                  # metric name -> [(value, iteration), ...]
        self._fresh = {}                   # metrics written since the last step()
        self._images = []                  # (name, tensor, iteration)
        self._scope = ""                   # prefix applied by name_scope()

    # ---- iteration ---------------------------------------------------------------------------
    @property
    def iter(self):
        return self._n

    iteration = iter

    def step(self):
        """Advance to the next iteration; `latest()` starts empty again."""
        self._n += 1
        self._fresh = {}

    # ---- scalars -----------------------------------------------------------------------------
    def put_scalar(self, name, value, smoothing_hint=True):
        key = self._scope + name
        number = float(value)
        self._series.setdefault(key, []).append((number, self._n))
        self._fresh[key] = number

    def put_scalars(self, *, smoothing_hint=True, **named_values):
        for name in named_values:
            self.put_scalar(name, named_values[name], smoothing_hint=smoothing_hint)

    def history(self, name):
        try:
            return self._series[name]
        except KeyError:
            raise KeyError("No history metric available for {}!".format(name)) from None

    def histories(self):
        return self._series

    def latest(self):
        return self._fresh

    @contextlib.contextmanager
    def name_scope(self, name):
        """Metrics written inside the block are stored as `<name>/<metric>`."""
        outer, self._scope = self._scope, name.rstrip("/") + "/"
        try:
            yield
        finally:
            self._scope = outer

    # ---- images ------------------------------------------------------------------------------
    def put_image(self, img_name, img_tensor):
        self._images.append((img_name, img_tensor, self._n))

    def clear_images(self):
        self._images = []

    @property
    def vis_data(self):
        return self._images

    # ---- ambient scope -----------------------------------------------------------------------
    def __enter__(self):
        _ambient.stack.append(self)
        return self

    def __exit__(self, exc_type, exc, tb):
        top = _ambient.stack.pop()
        if top is not self:
            raise AssertionError("EventStorage scopes must be exited in the order they were entered")
