from .configurator import Config  # noqa: F401
