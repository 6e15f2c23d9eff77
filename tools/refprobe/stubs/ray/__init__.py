"""In-process stand-in for `ray` so the reference's learner code can be imported
and run synchronously in THIS container (ray is not installed, no network).
Test infrastructure only: used by tools/refprobe/gen_*.py to produce golden vectors."""


class _Remote:
    def __init__(self, f):
        self._f = f

    def remote(self, *a, **k):
        return self._f(*a, **k)

    def __call__(self, *a, **k):
        return self._f(*a, **k)


def remote(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return _Remote(args[0])
    return lambda f: _Remote(f)


def get(x):
    return x


def wait(ids, num_returns=1):
    return ids[:num_returns], ids[num_returns:]


def init(*a, **k):
    pass


def is_initialized():
    return True
