"""Host-side mirror of the reference package of the same name (lib/model), backed by libfrcnn_hip.so."""
