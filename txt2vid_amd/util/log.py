import sys
import time


def _p(kind, msg):
    print('[%s] %s: %s' % (time.strftime('%Y-%m-%d %H:%M:%S'), kind, msg))
    sys.stdout.flush()


def status(msg):
    _p('STATUS', msg)


def warn(msg):
    _p('WARN', msg)


def error(msg):
    _p('ERROR', msg)
