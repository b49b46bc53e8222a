"""Host-side mirror of the reference package of the same name (lib/nets), backed by libfrcnn_hip.so."""
