import os


def ensure_exists(path):
    """txt2vid/util/dir.py:3-8."""
    if path and not os.path.exists(path):
        os.makedirs(path, exist_ok=True)
