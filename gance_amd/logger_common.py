"""Logger shared by the package; same logger name convention as gance/logger_common.py."""

import logging

LOGGER = logging.getLogger("gance_amd")
