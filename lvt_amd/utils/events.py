"""Ambient per-iteration metric storage (subset of vidgen/utils/events.py:16-25, 210-375 that the
model code touches: `get_event_storage().iter`, `put_scalar(s)`, `put_image`, context manager)."""
from collections import defaultdict
from contextlib import contextmanager

_CURRENT_STORAGE_STACK = []


def get_event_storage():
    assert len(_CURRENT_STORAGE_STACK), \
        "get_event_storage() has to be called inside a 'with EventStorage(...)' context!"
    return _CURRENT_STORAGE_STACK[-1]


class EventStorage:
    def __init__(self, start_iter=0):
        self._history = defaultdict(list)
        self._latest_scalars = {}
        self._iter = start_iter
        self._vis_data = []
        self._current_prefix = ""

    def put_image(self, img_name, img_tensor):
        self._vis_data.append((img_name, img_tensor, self._iter))

    def clear_images(self):
        self._vis_data = []

    def put_scalar(self, name, value, smoothing_hint=True):
        name = self._current_prefix + name
        value = float(value)
        self._history[name].append((value, self._iter))
        self._latest_scalars[name] = value

    def put_scalars(self, *, smoothing_hint=True, **kwargs):
        for k, v in kwargs.items():
            self.put_scalar(k, v, smoothing_hint=smoothing_hint)

    def history(self, name):
        if name not in self._history:
            raise KeyError("No history metric available for {}!".format(name))
        return self._history[name]

    def histories(self):
        return self._history

    def latest(self):
        return self._latest_scalars

    def step(self):
        self._iter += 1
        self._latest_scalars = {}

    @property
    def vis_data(self):
        return self._vis_data

    @property
    def iter(self):
        return self._iter

    @property
    def iteration(self):
        return self._iter

    def __enter__(self):
        _CURRENT_STORAGE_STACK.append(self)
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        assert _CURRENT_STORAGE_STACK[-1] == self
        _CURRENT_STORAGE_STACK.pop()

    @contextmanager
    def name_scope(self, name):
        old = self._current_prefix
        self._current_prefix = name.rstrip("/") + "/"
        yield
        self._current_prefix = old
