"""Host-side mirrors of the reference's net_utils modules on the hot path."""
