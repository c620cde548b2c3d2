"""Exact-division guard, same contract as gance/divisor.py:10-24."""

from typing import Union


def divide_no_remainder(numerator: Union[int, float], denominator: Union[int, float]) -> int:
    """
    numerator / denominator as an int.
    :raises ValueError: if the quotient is not a whole number.
    """
    quotient = numerator / denominator
    if quotient != int(quotient):
        raise ValueError(f"Cannot evenly divide {numerator} into {denominator}")
    return int(quotient)
