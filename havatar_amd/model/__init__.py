"""Host-side mirror of the reference's `model` package for the hot path (same names, arguments, error behaviour)."""
