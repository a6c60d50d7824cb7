"""phaser_amd: MI355X-native read-backed phasing hot path (phASER drop-in for the mapper + phasing core)."""
__version__ = "0.1.0"
