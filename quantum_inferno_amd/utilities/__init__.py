"""Host-side scalar helpers on the TFR path (mirror of quantum_inferno/utilities)."""
