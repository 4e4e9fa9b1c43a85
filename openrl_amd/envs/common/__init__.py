from .registration import make

__all__ = ["make"]
