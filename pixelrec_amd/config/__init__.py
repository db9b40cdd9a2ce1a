"""YAML + command-line configuration of a run (`Config`, see configurator.py); import path kept like the reference's
`REC.config` so `from <package>.config import Config` reads the same."""
from .configurator import Config  # noqa: F401

__all__ = ["Config"]
