from .losses import InfoNCELoss  # noqa: F401

__all__ = ["InfoNCELoss"]
