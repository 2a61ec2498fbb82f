"""Minimal stand-in for the `deprecation` package (an un-vendored dependency of the reference's
`vertical` module; it only decorates deprecated aliases and touches no arithmetic).
Used ONLY by tests/golden/gen_golden_vertical.py in the build container."""


def deprecated(*args, **kwargs):
    def wrap(func):
        return func

    return wrap


def fail_if_not_removed(func):
    return func
