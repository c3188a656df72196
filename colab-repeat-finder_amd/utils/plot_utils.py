"""Only the string helper the reference's tests import (reference utils/plot_utils.py:6-9);
the plotting functions are visualisation and out of scope."""


def shift_string_by(string, shift):
    """Rotate `string` to the right by `shift` characters ("AGTTT", 2 -> "TTAGT")."""
    n = len(string)
    if n == 0:
        return string
    cut = (n - shift % n) % n
    return string[cut:] + string[:cut]
