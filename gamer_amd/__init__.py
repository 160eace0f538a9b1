"""gamer_amd: MI355X-native (gfx950) train step for GAMER's Qwen3Multi SMB decoder."""
__version__ = "0.1.0"
