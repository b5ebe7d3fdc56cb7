"""Wall-clock bucket timer with the reference's tic / toc surface (lib/utils/timer.py; buckets of test_engine.py:167-182).

Built on time.perf_counter (monotonic); besides the reference's attributes (`diff`, `total_time`, `calls`, `average_time`)
it can be used as a context manager: `with timers['im_detect_bbox']: ...`.
"""
import time


class Timer(object):
    __slots__ = ('total_time', 'calls', 'diff', '_t0')

    def __init__(self):
        self.reset()

    def reset(self):
        self.total_time, self.calls, self.diff, self._t0 = 0.0, 0, 0.0, None

    @property
    def average_time(self):
        return self.total_time / self.calls if self.calls else 0.0

    def tic(self):
        self._t0 = time.perf_counter()

    def toc(self, average=True):
        assert self._t0 is not None, 'toc() without tic()'
        self.diff = time.perf_counter() - self._t0
        self._t0 = None
        self.calls += 1
        self.total_time += self.diff
        return self.average_time if average else self.diff

    def __enter__(self):
        self.tic()
        return self

    def __exit__(self, *exc):
        self.toc()
        return False
