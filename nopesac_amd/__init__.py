"""nopesac_amd — MI355X-native (gfx950) implementation of NopeSAC's inference hot path."""
__version__ = "0.1.0"
