from . import operators_set  # noqa: F401
