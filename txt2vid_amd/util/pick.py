import pickle


class _Unpickler(pickle.Unpickler):
    """Vocab pickles written by the reference name `txt2vid.data.Vocab` (or `__main__.Vocab`, when
    data/__init__.py:385-397 ran as a script): resolve both to this package's class."""

    def find_class(self, module, name):
        if name == 'Vocab' and module in ('txt2vid.data', '__main__'):
            from ..data import Vocab
            return Vocab
        return super().find_class(module, name)


def load(path):
    """txt2vid/util/pick.py:3-5."""
    with open(path, 'rb') as f:
        return _Unpickler(f).load()
