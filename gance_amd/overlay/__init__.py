"""Eye-tracking overlay gate: host mirror of gance/overlay, pixel work on the GPU."""
from . import overlay_common, overlay_eye_tracking  # noqa: F401
