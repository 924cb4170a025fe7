"""Stand-in for `termcolor` (test infrastructure only; see fvcore/__init__.py)."""


def colored(text, *args, **kwargs):
    return text
