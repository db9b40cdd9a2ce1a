from .enum_type import InputType  # noqa: F401
