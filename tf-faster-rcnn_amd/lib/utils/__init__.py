"""Host-side mirror of the reference package of the same name (lib/utils), backed by libfrcnn_hip.so."""
