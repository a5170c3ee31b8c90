"""Minimal stand-in for the `inflect` package (text/numbers.py:3): enough for digit-free synthetic sentences."""


class engine:
    _ONES = "zero one two three four five six seven eight nine".split()

    def number_to_words(self, n, andword="", zero="zero", group=0):
        return " ".join(self._ONES[int(c)] for c in str(n) if c.isdigit())

    def ordinal(self, w):
        return str(w) + "th"
