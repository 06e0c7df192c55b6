"""Table-driven argparse set-up: the entry points keep the reference's command-line surface (flag names, types, defaults) as a
compact table instead of one `add_argument` call per flag.

Row format: `name  kind  [default]  [!]` — kind is flag | int | float | str | ints | floats | strs (the plural forms take one or
more values); default `-` means None; a trailing `!` marks the flag required."""
import argparse

_SCALAR = {'int': int, 'float': float, 'str': str}
_LIST = {'ints': int, 'floats': float, 'strs': str}


def add_flags(parser, table):
    for row in table.strip().splitlines():
        row = row.split('#', 1)[0].split()
        if not row:
            continue
        required = row[-1] == '!'
        if required:
            row = row[:-1]
        name, kind = row[0], row[1]
        default = row[2:] if len(row) > 2 else None
        opt = '--' + name
        if kind == 'flag':
            parser.add_argument(opt, action='store_true', default=False)
            continue
        conv = _SCALAR.get(kind) or _LIST[kind]
        if default is None or default == ['-']:
            value = None
        elif kind in _LIST:
            value = [conv(v) for v in default]
        else:
            value = conv(default[0])
        kw = dict(type=conv, default=value)
        if kind in _LIST:
            kw['nargs'] = '+'
        if required:
            kw['required'] = True
        parser.add_argument(opt, **kw)
    return parser


def parser_from(*tables):
    p = argparse.ArgumentParser()
    for t in tables:
        add_flags(p, t)
    return p
