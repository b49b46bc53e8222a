"""Host-side mirror of the reference package of the same name (lib/layer_utils), backed by libfrcnn_hip.so."""
