class RollingAvg(object):
    """Mean of the last `window_size` values — txt2vid/util/metrics.py:3-23."""

    def __init__(self, window_size=10):
        self.window_size = max(1, int(window_size))
        self.values = []

    def update(self, v):
        self.values.append(float(v))
        if len(self.values) > self.window_size:
            self.values.pop(0)

    def get(self):
        return sum(self.values) / len(self.values) if self.values else 0.0
