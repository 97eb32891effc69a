"""Host-side mirrors of mjhmc.misc for the hot path (distributions)."""
