"""utils.timer.Timer -- lib/utils/timer.py:10-32."""
from frcnn_hip.runtime import Timer  # noqa: F401
