"""Host-side mirror of the reference's ``nerf/timer.py``: the running-average iteration timer the entry scripts print ETA lines with
(train.py:12,142,203-206).  Pure host bookkeeping -- nothing here touches the device; note that ``toc()`` measures host time between two
calls, so a caller who wants device time must synchronise first, exactly as with the reference's."""
import collections
import datetime
import time


class Timer:
    def __init__(self, max_len) -> None:
        self.deque = collections.deque(maxlen=max_len)        # the last `max_len` intervals, seconds
        self.last_time = 0.

    def get_mean_time(self):
        return sum(self.deque) / len(self.deque)

    def tic(self):
        self.last_time = time.time()

    def toc(self):
        self.deque.append(time.time() - self.last_time)
        return self.get_mean_time()

    def remaining_time(self, exec_needed: int):
        return str(datetime.timedelta(seconds=self.get_mean_time() * exec_needed))
