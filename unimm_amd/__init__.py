"""unimm_amd -- MI355X-native (gfx950) implementation of the UniMM-UL forward/backward hot path."""
__version__ = "0.1.0"
