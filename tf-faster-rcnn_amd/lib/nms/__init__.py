"""Host-side mirror of the reference package of the same name (lib/nms), backed by libfrcnn_hip.so."""
