import pickle


def load(path):
    """txt2vid/util/pick.py:3-5."""
    with open(path, 'rb') as f:
        return pickle.load(f)
