"""Output directories (txt2vid/util/dir.py)."""
from pathlib import Path


def ensure_exists(path):
    """mkdir -p; returns the path as given."""
    Path(path).mkdir(parents=True, exist_ok=True)
    return path
