from .common import make

__all__ = ["make"]
