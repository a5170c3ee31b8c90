"""Minimal stand-in for the `unidecode` package (text/cleaners.py:16): ASCII input passes through unchanged."""


def unidecode(s):
    return s.encode("ascii", "ignore").decode("ascii")
