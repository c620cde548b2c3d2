"""Image types shared across the path (names follow gance/gance_types.py:31-34)."""

from typing import Iterator, NewType, Optional

import numpy as np  # noqa: F401  pylint: disable=unused-import

# dimensions are (Height, Width, Colors), uint8 RGB
RGBInt8ImageType = NewType("RGBInt8ImageType", "np.ndarray[np.uint8]")  # type: ignore
ImageSourceType = Iterator[RGBInt8ImageType]
OptionalImageSourceType = Iterator[Optional[RGBInt8ImageType]]
