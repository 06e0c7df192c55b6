import time


class Stopwatch(object):
    """txt2vid/util/stopwatch.py:3-22."""

    def __init__(self):
        self.start_time = None
        self.elapsed_time = 0.0

    def start(self):
        self.start_time = time.time()

    def stop(self):
        if self.start_time is not None:
            self.elapsed_time = time.time() - self.start_time
        return self.elapsed_time
