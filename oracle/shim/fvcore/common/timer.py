from time import perf_counter


class Timer:
    def __init__(self):
        self.reset()

    def reset(self):
        self._start = perf_counter()
        self._paused = None
        self._total_paused = 0

    def pause(self):
        self._paused = perf_counter()

    def is_paused(self):
        return self._paused is not None

    def resume(self):
        self._total_paused += perf_counter() - self._paused
        self._paused = None

    def seconds(self):
        end = self._paused if self._paused is not None else perf_counter()
        return end - self._start - self._total_paused
